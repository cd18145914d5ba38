#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE of the headline kernel, with the streaming-only calibration run
# (SCG_ABLATE=2 = phase A alone = exactly the algorithmic bytes) next to the full kernel.
cd /tmp && export TMPDIR=/tmp
for A in 2 0; do
  for C in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    TAG=$(echo $C | tr ' ' '_')
    OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic_a${A}_$TAG
    SCG_LIB=$GRAFT_REPO_ROOT/tools/ablate/libscg_ablate.so SCG_ABLATE=$A timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --e2e-sample 0 --e2e-file-sample 0 --settle 0 --reads 20000000 --steps 2 --warmup 1 > $OUT.log 2>&1
    python3 - <<PY
import csv,glob
from collections import defaultdict
acc=defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "single_staged" in row["Kernel_Name"]:
            acc[(row["Counter_Name"], row["Dispatch_Id"])].append(float(row["Counter_Value"]))
per=defaultdict(list)
for (c,d),v in acc.items(): per[c].append(sum(v))
for c in sorted(per): print("ablate=$A  %-24s per dispatch (20M reads, 3.0e9 algorithmic bytes): %.4e" % (c, sum(per[c])/len(per[c])))
PY
  done
done
