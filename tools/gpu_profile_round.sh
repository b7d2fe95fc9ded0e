#!/usr/bin/env bash
# Round profile on the MI355X box: rocprofv3 kernel trace of bench.py + the HBM-traffic PMC passes (one counter set per pass,
# kernel-trace only), summaries under gpurun_out/<tag>/ (raw databases removed: they exceed the copy-back limit).
set -u
TAG="${1:-prof}"
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o trace -- python "$REPO/bench.py" --steps 100 --warmup 10 --no-cpu-baseline --objects-per-gpu 0 > "$OUT/rocprof.log" 2>&1); echo "rocprof exit $?"
DB=$(find "$OUT/prof" -name "*_results.db" | head -1); python "$REPO/tools/rocpd_stats.py" "$DB" "$OUT/kernel_stats.md" | head -14; rm -rf "$OUT/prof"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum"; do
  N=$(echo $C | tr ' ' '_')
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $C -d "$OUT/$N" -o pmc -- python "$REPO/bench.py" --steps 30 --warmup 10 --no-cpu-baseline --objects-per-gpu 0 > "$OUT/$N.log" 2>&1); echo "pmc $C exit $?"
done
python "$REPO/tools/rocpd_pmc.py" "$OUT" > "$OUT/pmc_summary.md"; grep -E "k_fused_train|k_grid_scatter|k_optimizer" "$OUT/pmc_summary.md"
for C in FETCH_SIZE WRITE_SIZE TCC_HIT_sum_TCC_MISS_sum TCP_TCC_READ_REQ_sum_TCC_REQ_sum; do rm -rf "$OUT/$C"; done
timeout 600 python "$REPO/bench.py" --steps 200 --warmup 20 > "$OUT/bench.log" 2>&1; tail -1 "$OUT/bench.log" > "$OUT/bench.json"; cut -c1-300 "$OUT/bench.json"
