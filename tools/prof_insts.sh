#!/bin/bash
# GPU box: dynamic instruction counts per wavefront of a bench configuration's counting kernel, for the in-tree
# library and every build under tools/ab/*.so (one rocprofv3 --pmc pass each).
# usage: tools/prof_insts.sh "<configs>" [kernel substring]
CONFIGS=${1:-2}; KERNEL=${2:-staged_kernel}
cd /tmp && export TMPDIR=/tmp
for C in $CONFIGS; do
  for L in screencounter_amd/libscg.so tools/ab/*.so; do
    [ -e $GRAFT_REPO_ROOT/$L ] || continue
    N=$(basename $L .so)
    OUT=$GRAFT_REPO_ROOT/gpurun_out/insts_c${C}_$N
    rm -rf $OUT
    SCG_LIB=$GRAFT_REPO_ROOT/$L timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --config $C --cpu-sample 0 --e2e-sample 0 --e2e-file-sample 0 --settle 0 --reads 20000000 --steps 2 --warmup 1 > $OUT.log 2>&1
    echo "config $C $L rc=$?"
    python3 - <<PY
import csv,glob
from collections import defaultdict
acc=defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "$KERNEL" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
w=sum(acc["SQ_WAVES"]) or 1
print("   " + "  ".join("%s %.1f" % (k.replace("SQ_INSTS_","").replace("SQ_",""), sum(acc[k])/w) for k in sorted(acc) if k != "SQ_WAVES"))
PY
  done
done
