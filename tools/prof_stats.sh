#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats of the bench command for several configurations, with enough timed steps that the
# first launches (cold clocks) do not carry the average.  Writes gpurun_out/stats_<tag>_config<N>.csv (copy to profiles/).
# usage: tools/prof_stats.sh <tag> "<configs>" [steps]
TAG=$1; CONFIGS=${2:-2}; STEPS=${3:-200}
cd /tmp && export TMPDIR=/tmp
for C in $CONFIGS; do
  D=$GRAFT_REPO_ROOT/gpurun_out/stats_${TAG}_config$C
  rm -rf $D
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/bench.py --config $C --cpu-sample 0 --e2e-sample 0 --e2e-file-sample 0 --steps $STEPS --warmup 5 > $D.log 2>&1 || { echo "config $C failed"; tail -3 $D.log; exit 1; }
  f=$(find $D -name "*kernel_stats.csv" | head -1)
  cp $f $GRAFT_REPO_ROOT/gpurun_out/stats_${TAG}_config$C.csv
  rm -rf $D
  python3 - $GRAFT_REPO_ROOT/gpurun_out/stats_${TAG}_config$C.csv $C <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "staged_kernel" in r["Name"] or "passes_kernel" in r["Name"]:
        avg = float(r["AverageNs"]) / 1e6
        print("config %s %-60s calls %4s avg %.4f ms min %.4f max %.4f  frac %.3f" % (sys.argv[2], r["Name"][:60], r["Calls"], avg, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6, 15e9 / (avg / 1e3) / 8e12))
PY
done
