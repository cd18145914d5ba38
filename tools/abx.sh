#!/bin/bash
# GPU box: A/B kernel timing of the in-tree libscg.so against every build under tools/ab/*.so, on the same box,
# alternating libraries, for the listed bench configurations.
# usage: tools/abx.sh "<configs>" [rounds] [extra bench args...]       e.g. tools/abx.sh "2 3 4 5" 2
cd $GRAFT_REPO_ROOT
CONFIGS=${1:-2}; ROUNDS=${2:-2}; shift; shift
for C in $CONFIGS; do
  for r in $(seq $ROUNDS); do
    for L in screencounter_amd/libscg.so tools/ab/*.so; do
      SCG_LIB=$GRAFT_REPO_ROOT/$L timeout -k 10 200 python3 bench.py --config $C --cpu-sample 0 --e2e-sample 0 --e2e-file-sample 0 --settle 0.7 --steps 60 --warmup 5 "$@" > gpurun_out/abx.log 2>&1 || { echo "$L failed"; tail -3 gpurun_out/abx.log; exit 1; }
      tail -1 gpurun_out/abx.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('config $C', '%-34s' % '$L', 'kernel_ms', d['roofline']['avg_kernel_ms'], 'step_ms', d['ms_per_step'], 'mapped', d['mapped_fraction'])"
    done
  done
done
