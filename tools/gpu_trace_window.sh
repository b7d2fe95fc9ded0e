#!/usr/bin/env bash
# rocprofv3 kernel trace of one window of tools/profile_window.py (no PMC passes): per-kernel mean durations -> gpurun_out/<tag>/kernel_window.md
#   [MON_CORE_LIB=...] [WINDOW_ARGS="--log2-hashmap-size 22"] tools/gpu_trace_window.sh <tag> [extra-steps-before-the-window]
set -u
TAG="${1:-win}"; EXTRA="${2:-0}"
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"; export TMPDIR=/tmp
CMD="python $REPO/tools/profile_window.py --warmup 5 --steps 20 --extra $EXTRA ${WINDOW_ARGS:-}"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d "$OUT/trace" -o trace -- $CMD > "$OUT/trace.log" 2>&1); echo "trace exit $?"; grep "^window" "$OUT/trace.log"
python "$REPO/tools/rocpd_window.py" "$OUT/trace" --skip $(( 5 + EXTRA )) --take 20 > "$OUT/kernel_window.md"; grep -E "^\| k_|^\| kernel|sum of" "$OUT/kernel_window.md" | cut -c1-150; rm -rf "$OUT/trace"
