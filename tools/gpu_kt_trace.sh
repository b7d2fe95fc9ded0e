#!/usr/bin/env bash
# per-kernel rocprofv3 stats of tools/kernel_times.py for a configuration:   tools/gpu_kt_trace.sh '<json cfg>' [tag]
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; TAG="${2:-kt_trace}"; OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"; export TMPDIR=/tmp
(cd /tmp && MON_KT_CFG="$1" MON_KT_DENSE_ONLY=1 timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/p" -o t -- python "$REPO/tools/kernel_times.py" > "$OUT/run.log" 2>&1); tail -1 "$OUT/run.log"
DB=$(find "$OUT/p" -name "*_results.db" | head -1); python "$REPO/tools/rocpd_stats.py" "$DB" "$OUT/kernel_stats.md" | head -16; rm -rf "$OUT/p"
