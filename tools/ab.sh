#!/bin/bash
# GPU box: alternate bench runs between the in-tree libscg.so and the builds under tools/ab/*.so
# (made by tools/ab_build.sh <git-rev> <name>), so that kernel timings are compared on one box.
# usage: tools/ab.sh [rounds] [bench args...]
cd $GRAFT_REPO_ROOT
ROUNDS=${1:-3}; shift
for r in $(seq $ROUNDS); do
  for L in screencounter_amd/libscg.so tools/ab/*.so; do
    SCG_LIB=$GRAFT_REPO_ROOT/$L timeout -k 10 200 python3 bench.py --cpu-sample 0 --e2e-sample 0 --e2e-file-sample 0 --settle 0 --steps 5 --warmup 1 "$@" > gpurun_out/ab.log 2>&1 || { echo "$L failed"; tail -3 gpurun_out/ab.log; exit 1; }
    tail -1 gpurun_out/ab.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', 'kernel_ms', d['roofline']['avg_kernel_ms'], 'Mreads/s', d['value'])"
  done
done
