#!/bin/bash
# GPU box: the device inflate kernel on the two BGZF measurement streams (the benchmark stream; random sequences with
# varied qualities): end-to-end rate, DEFLATE symbols of the file (tools/deflate_symbols.cpp), kernel-trace stats and
# instruction counters of the inflate kernel per symbol.  Writes gpurun_out/bgzf_pmc_<tag>.txt.
# usage: tools/prof_bgzf_pmc.sh <tag> [SCG_INFLATE_LANES=0]
TAG=$1; shift
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/bgzf_pmc_$TAG.txt
g++ -O2 -std=c++17 -o /tmp/deflate_symbols tools/deflate_symbols.cpp || exit 1
: > $OUT
for S in "bench 16000000" "random 8000000"; do
  set -- $S; KIND=$1; N=$2
  F=/dev/shm/scg_pmc_$KIND.fastq.gz
  timeout -k 10 600 python3 tools/bgzf_stream.py make $KIND $N $F 2>/dev/null | tee -a $OUT
  /tmp/deflate_symbols $F | tee -a $OUT
  echo "-- end to end ($KIND, $N reads) $ENVSET" | tee -a $OUT
  env $ENVSET SCG_DEVICE_INFLATE=2 timeout -k 10 300 python3 tools/bgzf_stream.py count $F 3 2>/dev/null | tee -a $OUT
  D=$GRAFT_REPO_ROOT/gpurun_out/bgzf_pmc_${TAG}_$KIND
  rm -rf $D
  (cd /tmp && export TMPDIR=/tmp && env $ENVSET SCG_DEVICE_INFLATE=2 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 $GRAFT_REPO_ROOT/tools/bgzf_stream.py count $F 1 > $D.stats.log 2>&1)
  (cd /tmp && export TMPDIR=/tmp && env $ENVSET SCG_DEVICE_INFLATE=2 timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $D/pmc -- python3 $GRAFT_REPO_ROOT/tools/bgzf_stream.py count $F 1 > $D.pmc.log 2>&1)
  python3 - $D $OUT <<'PY'
import csv, glob, sys
from collections import defaultdict
d, out = sys.argv[1], sys.argv[2]
lines = open(out).read().splitlines()
symbols = [int(l.split()[5]) for l in lines if l.startswith("members")][-1]
text = [int(l.split()[3]) for l in lines if l.startswith("members")][-1]
acc = defaultdict(float); calls = 0
for f in glob.glob(d + "/pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "inflate_members" in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"])
with open(out, "a") as o:
    # (two calls per run: the warm-up on a 16-entry library and the measured one; both inflate the whole file)
    o.write("-- inflate kernel, counters summed over the run's dispatches (2 passes over the file), per DEFLATE symbol / per byte of text\n")
    for k in sorted(acc):
        o.write(f"   {k:20s} {acc[k]:.4g}   {acc[k] / (2 * symbols):8.2f} per symbol   {acc[k] / (2 * text):7.3f} per byte\n")
    for f in glob.glob(d + "/stats/**/*kernel_stats.csv", recursive=True):
        o.write("-- kernel-trace stats (same command)\n")
        for row in list(csv.DictReader(open(f)))[:8]:
            o.write(f"   {row['Name'][:90]:90s} calls {row['Calls']:>4s} avg {float(row['AverageNs']) / 1e6:8.3f} ms  total {float(row['TotalDurationNs']) / 1e6:8.1f} ms\n")
PY
  rm -f $F
done
cat $OUT
