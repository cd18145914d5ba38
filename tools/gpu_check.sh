#!/bin/bash
# Runs on the GPU box: GPU tests, one bench line, and the PMC summary of the headline kernel.
# usage: tools/gpu_check.sh <tag> [kernel-substring]
TAG=$1; KSUB=${2:-single_staged}
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_$TAG.log
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --cpu-sample 0 --e2e-sample 0 --e2e-file-sample 0 --settle 0 > gpurun_out/bench_$TAG.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/bench_$TAG.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('Mreads/s', d['value'], 'frac', d['roofline']['frac'], 'kernel_ms', d['roofline']['avg_kernel_ms'], 'mapped', d['mapped_fraction'])"
tools/prof_pmc.sh pmc_$TAG --reads 20000000 --steps 2 --warmup 1 > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_$TAG $KSUB
