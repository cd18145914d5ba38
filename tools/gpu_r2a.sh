#!/bin/bash
# GPU box, round 2 first pass: box probe, GPU tests, default bench line, SQ counter passes for configs 3/4/5.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2a; mkdir -p $O
{ nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python3 -c "import os;print('affinity',len(os.sched_getaffinity(0)))"; free -g | head -2; df -h /dev/shm | tail -1; } > $O/box.txt 2>&1
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 600 python bench.py > $O/bench_default.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_default.log | cut -c1-1500
for C in 3 4 5; do
  timeout -k 10 300 python bench.py --config $C --steps 100 --warmup 5 --cpu-sample 0 --e2e-sample 0 --e2e-file-sample 0 > $O/bench_cfg$C.log 2>&1; echo "bench cfg$C rc=$?"; tail -1 $O/bench_cfg$C.log | cut -c1-700
done
