#!/bin/bash
# GPU box: the evidence bundle for profiles/ of one bench configuration -- kernel-trace stats of the default bench
# command, HBM-side counters in separate --pmc passes (full kernel, and for the single-barcode kernel the streaming-only
# calibration run of the -DSCG_ABLATE measurement build), SQ instruction / wait counters, and the default bench line
# (with the CPU baseline and its parity check).
# usage: tools/make_profiles.sh <tag> [config]
TAG=$1; CFG=${2:-2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_${TAG}_config$CFG
mkdir -p $OUT
B="$GRAFT_REPO_ROOT/bench.py --config $CFG --cpu-sample 0 --e2e-sample 0 --e2e-file-sample 0 --settle 0"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $B --steps 20 --warmup 3 > $OUT/stats_bench.log 2>&1
echo "stats rc=$?"
ABL="0"; [ "$CFG" = "2" ] && ABL="2 0"
for A in $ABL; do
  for C in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum"; do
    T=$(echo $C | tr ' ' '_')
    if [ "$A" = "0" ]; then
      timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_a${A}_$T -- python3 $B --steps 3 --warmup 1 > $OUT/pmc_a${A}_$T.log 2>&1
    else
      SCG_LIB=$GRAFT_REPO_ROOT/tools/ablate/libscg_ablate.so SCG_ABLATE=$A timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_a${A}_$T -- python3 $B --steps 3 --warmup 1 > $OUT/pmc_a${A}_$T.log 2>&1
    fi
    echo "pmc ablate=$A $C rc=$?"
  done
done
i=0
for P in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
         "TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/sq/pass$i -- python3 $B --steps 3 --warmup 1 > $OUT/sq_pass$i.log 2>&1
  echo "sq pass$i rc=$?"
  i=$((i+1))
done
KSUB=staged; [ "$CFG" = "4" ] && KSUB=dual_passes      # (the pair search in two passes: dual_passes_kernel)
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/sq $KSUB > $OUT/pmc_summary.txt 2>&1
cd $GRAFT_REPO_ROOT && timeout -k 10 900 python bench.py --config $CFG > $OUT/bench_default.log 2>&1; echo "bench rc=$?"; tail -1 $OUT/bench_default.log | cut -c1-400
