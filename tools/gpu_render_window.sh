#!/usr/bin/env bash
# rocprofv3 evidence for the render half of the metric (tools/render_window.py: 313x229 crop, the 60-view orbit, 64^3 mesh):
# one kernel trace, then one counter set per pass (kernel-trace only: gpurun refuses --pmc together with other trace domains).
# Summaries -> gpurun_out/<tag>/*.md, raw databases removed.
#   tools/gpu_render_window.sh <tag> [render_window.py arguments]
set -u
TAG="${1:-rwin}"; shift || true
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp
ONLY="k_render,k_fused_render,k_tile,k_encode_feat,k_build,k_occ,k_grid_points,k_encode,k_mlp,k_extract,k_mc,k_mesh,k_density,k_copy"
python "$REPO/tools/render_window.py" "$@" > "$OUT/plain.log" 2>&1; cat "$OUT/plain.log"
CMD="python $REPO/tools/render_window.py --train 300 --crops 10 --orbit 60 --meshes 3 $*"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- $CMD > "$OUT/trace.log" 2>&1); echo "trace exit $?"; tail -5 "$OUT/trace.log"
python "$REPO/tools/rocpd_window.py" "$OUT/trace" --skip 0 --take 1000000 --only "$ONLY" > "$OUT/kernel_window.md"; cat "$OUT/kernel_window.md"; rm -rf "$OUT/trace"
i=0
while IFS= read -r C; do
  [ -z "$C" ] && continue
  i=$((i + 1))
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc$i" -o pmc -- $CMD > "$OUT/pmc$i.log" 2>&1); echo "pmc [$C] exit $?"
done <<'LIST'
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum
TCP_TCC_READ_REQ_sum TCC_REQ_sum
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
LIST
python "$REPO/tools/rocpd_window.py" "$OUT" --skip 0 --take 1000000 --only "$ONLY" > "$OUT/pmc_window.md"; cat "$OUT/pmc_window.md"
for d in "$OUT"/pmc*/; do rm -rf "$d"; done
grep -il "error\|invalid\|not found" "$OUT"/pmc*.log 2>/dev/null | head
