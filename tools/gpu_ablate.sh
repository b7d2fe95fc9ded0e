#!/usr/bin/env bash
# Timing ablations of k_fused_train (results are wrong by construction; kernel durations only).
set -u
TAG="${1:-ablate}"
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp
for A in 0 8 2 6; do
  (cd /tmp && MON_OPTIONS=fused_ablate=$A timeout 300 rocprofv3 --kernel-trace -d "$OUT/a$A" -o t -- python "$REPO/bench.py" --steps 60 --warmup 10 --no-cpu-baseline > "$OUT/a$A.log" 2>&1)
  echo "ablate=$A"; python "$REPO/tools/rocpd_stats.py" "$OUT/a$A/t_results.db" | grep -E "fused_train|scatter|optimizer" | cut -c1-160
done
