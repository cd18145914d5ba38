#!/bin/bash
# GPU box: VALU/SALU instruction counts of the headline kernel with phases switched off (SCG_ABLATE; needs the
# measurement build of tools/ablate_build.sh, made in the container before gpurun).
export SCG_LIB=$GRAFT_REPO_ROOT/tools/ablate/libscg_ablate.so
cd /tmp && export TMPDIR=/tmp
for A in 0 1 2; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/ablate_$A
  SCG_ABLATE=$A timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --e2e-sample 0 --e2e-file-sample 0 --settle 0 --reads 20000000 --steps 2 --warmup 1 > $OUT.log 2>&1
  echo "ablate=$A rc=$?"; tail -1 $OUT.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(' kernel_ms', d['roofline']['avg_kernel_ms'])"
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/.. single_staged 2>/dev/null | head -0
  python3 - <<PY
import csv,glob
from collections import defaultdict
acc=defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "single_staged" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
w=sum(acc["SQ_WAVES"])
for k in sorted(acc): print("  %-20s per wave %.1f" % (k, sum(acc[k])/w))
PY
done
