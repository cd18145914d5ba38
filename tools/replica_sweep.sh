#!/bin/bash
# GPU box: headline kernel time against the number of privatised counter addresses.
cd $GRAFT_REPO_ROOT
for L in 20 16 18 22 24 20; do
  SCG_REPLICA_ADDR_LOG2=$L timeout -k 10 200 python3 bench.py --cpu-sample 0 --e2e-sample 0 --e2e-file-sample 0 --settle 0 --steps 5 --warmup 1 "$@" > gpurun_out/replica.log 2>&1
  tail -1 gpurun_out/replica.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('addr_log2 $L kernel_ms', d['roofline']['avg_kernel_ms'], 'step_ms', d['ms_per_step'])"
done
