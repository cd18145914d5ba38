#!/bin/bash
# GPU box: un-profiled kernel time of the headline kernel with phases switched off (SCG_ABLATE:
# 1 = no phase C, 2 = phase A only, 3 = everything but the counter atomics).
cd $GRAFT_REPO_ROOT
export SCG_LIB=$GRAFT_REPO_ROOT/tools/ablate/libscg_ablate.so   # tools/ablate_build.sh
for A in 0 3 1 2 0; do   # with SCG_TALLY=0 for the atomics figure
  SCG_ABLATE=$A timeout -k 10 200 python3 bench.py --cpu-sample 0 --e2e-sample 0 --e2e-file-sample 0 --settle 0 --steps 5 --warmup 1 "$@" > gpurun_out/ablate_time.log 2>&1
  tail -1 gpurun_out/ablate_time.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate $A kernel_ms', d['roofline']['avg_kernel_ms'])"
done
