#!/usr/bin/env bash
# One PMC pass with the counters given on the command line; prints the per-kernel means and removes the raw databases
# (they exceed gpurun's 64 MiB copy-back limit).  Usage: [PMC_CMD='python tools/param_crc.py 30'] tools/gpu_pmc_one.sh <tag> <counter> [<counter> ...]
set -u
TAG="$1"; shift
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp
(cd /tmp && timeout ${PMC_TIMEOUT:-200} rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/p" -o pmc -- ${PMC_CMD:-python "$REPO/bench.py" --steps ${PMC_STEPS:-10} --warmup 5 --no-cpu-baseline --objects-per-gpu 0} > "$OUT/run.log" 2>&1)
echo "pmc exit $?"; tail -1 "$OUT/run.log" | cut -c1-200
python "$REPO/tools/rocpd_pmc.py" "$OUT" > "$OUT/pmc_summary.md"; rm -rf "$OUT/p"
grep -E "k_fused_train|k_grid_scatter|k_optimizer|k_big" "$OUT/pmc_summary.md"
