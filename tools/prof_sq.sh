#!/bin/bash
# GPU box: SQ instruction-mix and wait-cycle counters of one bench configuration (3 rocprofv3 --pmc passes).
# usage: tools/prof_sq.sh <config> <outdir under gpurun_out>
set -u
CFG=$1; OUT=$GRAFT_REPO_ROOT/gpurun_out/$2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"
)
i=0
for P in "${PASSES[@]}"; do
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --cpu-sample 0 --e2e-sample 0 --e2e-file-sample 0 --settle 0 --reads 20000000 --steps 2 --warmup 1 > $OUT/pass$i.log 2>&1
  echo "cfg $CFG pass$i rc=$?"
  i=$((i+1))
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT staged > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
