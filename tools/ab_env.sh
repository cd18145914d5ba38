#!/bin/bash
# GPU box: kernel timing of bench configurations under different environments / libraries, alternating on one box.
# usage: tools/ab_env.sh "<configs>" "<VAR=val ...|->" ...     each further argument is one variant ("-" = plain)
cd $GRAFT_REPO_ROOT
CONFIGS=$1; shift
for C in $CONFIGS; do
  for r in 1 2; do
    for V in "$@"; do
      [ "$V" = "-" ] && E="" || E="$V"
      env $E timeout -k 10 200 python3 bench.py --config $C --cpu-sample 0 --e2e-sample 0 --e2e-file-sample 0 --settle 0.7 --steps 60 --warmup 5 > gpurun_out/abe.log 2>&1 || { echo "$V failed"; tail -3 gpurun_out/abe.log; exit 1; }
      tail -1 gpurun_out/abe.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('config $C', '%-60s' % '$V', 'kernel_ms', d['roofline']['avg_kernel_ms'], 'step_ms', d['ms_per_step'], 'mapped', d['mapped_fraction'])"
    done
  done
done
