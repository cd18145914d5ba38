#!/bin/bash
# GPU box: kernel time of every bench configuration for the in-tree library and the builds under tools/ab/.
cd $GRAFT_REPO_ROOT
for C in 2 3 4 5; do
  for L in screencounter_amd/libscg.so tools/ab/*.so; do
    SCG_LIB=$GRAFT_REPO_ROOT/$L timeout -k 10 200 python3 bench.py --config $C --cpu-sample 0 --e2e-sample 0 --e2e-file-sample 0 --settle 0 --steps 5 --warmup 1 > gpurun_out/abc.log 2>&1 || { echo "$L failed"; tail -3 gpurun_out/abc.log; exit 1; }
    tail -1 gpurun_out/abc.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('config $C', '$L', 'kernel_ms', d['roofline']['avg_kernel_ms'], d['unit'], d['value'])"
  done
done
