#!/bin/bash
# GPU box: the evidence bundle for profiles/ -- kernel-trace stats of the default bench command,
# FETCH/WRITE PMC passes (full kernel + streaming-only calibration), per-phase instruction counts.
# usage: tools/make_profiles.sh <tag> [config]
TAG=$1; CFG=${2:-2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 5 --warmup 1 --cpu-sample 0 --e2e-sample 0 --e2e-file-sample 0 --settle 0 > $OUT/stats_bench.log 2>&1
echo "stats rc=$?"
for A in 2 0; do
  for C in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    T=$(echo $C | tr ' ' '_')
    SCG_LIB=$GRAFT_REPO_ROOT/tools/ablate/libscg_ablate.so SCG_ABLATE=$A timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_a${A}_$T -- python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 3 --warmup 1 --cpu-sample 0 --e2e-sample 0 --e2e-file-sample 0 --settle 0 > $OUT/pmc_a${A}_$T.log 2>&1
    echo "pmc ablate=$A $C rc=$?"
  done
done
cd $GRAFT_REPO_ROOT && timeout -k 10 600 python bench.py --config $CFG > $OUT/bench_default.log 2>&1; echo "bench rc=$?"; tail -1 $OUT/bench_default.log
