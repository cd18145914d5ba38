#!/bin/bash
# GPU box: kernel durations of the BGZF path for several window sizes (SCG_WINDOW_KB), one rocprofv3 kernel trace each.
# usage: tools/prof_bgzf_window.sh "<kb> <kb> ..."
cd /tmp && export TMPDIR=/tmp
for KB in $1; do
  export SCG_WINDOW_KB=$KB
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_bgzf_w$KB
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bgzf_w$KB -- python3 $GRAFT_REPO_ROOT/tools/prof_bgzf.py 16000000 > $GRAFT_REPO_ROOT/gpurun_out/prof_bgzf_w$KB.log 2>&1 || { echo "failed"; tail -5 $GRAFT_REPO_ROOT/gpurun_out/prof_bgzf_w$KB.log; exit 1; }
  echo "== window $KB KB"; grep "^rep" $GRAFT_REPO_ROOT/gpurun_out/prof_bgzf_w$KB.log
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_bgzf_w$KB -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 1.0:
        print("   %-60s calls %5s avg %9.3f ms total %8.1f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
PY
  t=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_bgzf_w$KB -name "*kernel_trace.csv" | head -1)
  python3 - "$t" <<'PY'
import csv, sys
# overlap: wall time covered by inflate kernels vs the sum of their durations
iv = []
for r in csv.DictReader(open(sys.argv[1])):
    if "inflate_members" in r["Kernel_Name"]:
        iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
iv.sort()
tot = sum(b - a for a, b in iv)
cover, end = 0, 0
for a, b in iv:
    if b <= end: continue
    cover += b - max(a, end); end = b
print("   inflate: %d launches, sum %.1f ms, wall covered %.1f ms" % (len(iv), tot / 1e6, cover / 1e6))
PY
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_bgzf_w$KB
done
