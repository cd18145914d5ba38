#!/bin/bash
# GPU box: sample the shader clock / power while the headline kernel runs back to back.
cd $GRAFT_REPO_ROOT
(timeout -k 10 120 python3 bench.py --cpu-sample 0 --e2e-sample 0 --e2e-file-sample 0 --settle 0 --steps 1500 --warmup 5 > gpurun_out/clock_bench.log 2>&1) &
BP=$!
sleep 14
for i in $(seq 12); do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (edge|junction|hot)" | tr '\n' ' ' | sed 's/GPU\[0\]//g; s/\s\+/ /g'
  echo
  sleep 0.5
done
wait $BP
tail -1 gpurun_out/clock_bench.log | cut -c1-300
