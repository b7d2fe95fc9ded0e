#!/usr/bin/env bash
# Instruction-mix / stall PMC passes (separate runs, kernel-trace only) for the training kernels.  Usage: tools/gpu_pmc_mix.sh [tag]
set -u
TAG="${1:-pmcmix}"
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SMEM" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
         "SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT" \
         "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_IOPS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_BRANCH SQ_WAIT_ANY"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C -d "$OUT/p$i" -o pmc -- python "$REPO/bench.py" --steps 30 --warmup 10 --no-cpu-baseline > "$OUT/p$i.log" 2>&1)
  echo "pmc pass $i exit $?"; tail -2 "$OUT/p$i.log" | cut -c1-300
done
python "$REPO/tools/rocpd_pmc.py" "$OUT" | tee "$OUT/pmc_summary.md"
