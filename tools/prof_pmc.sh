#!/bin/bash
# Runs on the GPU box: separate rocprofv3 --pmc passes over a short bench run, one directory per pass.
# usage: tools/prof_pmc.sh <outdir under gpurun_out> [bench args...]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
 "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_WAVES"
 "FETCH_SIZE"
 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
 "TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum TCC_REQ_sum GRBM_GUI_ACTIVE"
)
i=0
for P in "${PASSES[@]}"; do
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --e2e-sample 0 --e2e-file-sample 0 --settle 0 "$@" > $OUT/pass$i.log 2>&1
  echo "pass$i rc=$? : $P"
  i=$((i+1))
done
find $OUT -name "*counter_collection.csv" | head -20
