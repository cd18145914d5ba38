#!/bin/bash
# GPU box: headline bench with the tally path off / on (SCG_TALLY), step time and dominant-kernel time.
cd $GRAFT_REPO_ROOT
for M in 0 1 0 1; do
  for C in "$@"; do
    SCG_TALLY=$M timeout -k 10 200 python3 bench.py --config $C --cpu-sample 0 --e2e-sample 0 --e2e-file-sample 0 --settle 0 --steps 5 --warmup 1 > gpurun_out/tally.log 2>&1 || { echo failed; tail -3 gpurun_out/tally.log; exit 1; }
    tail -1 gpurun_out/tally.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('config $C tally $M step_ms', d['ms_per_step'], 'kernel_ms', d['roofline']['avg_kernel_ms'], d['unit'], d['value'])"
  done
done
