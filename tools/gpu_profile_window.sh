#!/usr/bin/env bash
# rocprofv3 evidence for the window the driver times (bench.py --gpus 1 --steps 20 --warmup 5: iterations 5..25 from init, every
# sample still carries a gradient) and, with EXTRA=800, for the late-training regime.  One counter set per pass, kernel-trace only
# (gpurun refuses --pmc together with other trace domains).  Summaries -> gpurun_out/<tag>/*.md, raw databases removed.
#   tools/gpu_profile_window.sh <tag> [extra-steps-before-the-window]
set -u
TAG="${1:-win}"; EXTRA="${2:-0}"
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python $REPO/tools/profile_window.py --warmup 5 --steps 20 --extra $EXTRA ${WINDOW_ARGS:-}"      # WINDOW_ARGS: e.g. "--log2-hashmap-size 22 --objects 8"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- $CMD > "$OUT/trace.log" 2>&1); echo "trace exit $?"; tail -1 "$OUT/trace.log"
python "$REPO/tools/rocpd_window.py" "$OUT/trace" --skip $(( (5 + EXTRA) * ${WINDOW_OBJECTS:-1} )) --take $(( 20 * ${WINDOW_OBJECTS:-1} )) > "$OUT/kernel_window.md"; cat "$OUT/kernel_window.md"; rm -rf "$OUT/trace"
i=0
while IFS= read -r C; do
  [ -z "$C" ] && continue
  i=$((i + 1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc$i" -o pmc -- $CMD > "$OUT/pmc$i.log" 2>&1); echo "pmc [$C] exit $?"
done <<'LIST'
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum
TCP_TCC_READ_REQ_sum TCC_REQ_sum
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
GRBM_GUI_ACTIVE
LIST
python "$REPO/tools/rocpd_window.py" "$OUT" --skip $(( (5 + EXTRA) * ${WINDOW_OBJECTS:-1} )) --take $(( 20 * ${WINDOW_OBJECTS:-1} )) > "$OUT/pmc_window.md"; cat "$OUT/pmc_window.md"
for d in "$OUT"/pmc*/; do rm -rf "$d"; done
grep -il "error\|invalid\|not found" "$OUT"/pmc*.log 2>/dev/null | head
