#!/usr/bin/env bash
# PMC passes (separate runs, kernel-trace only) for HBM traffic of the training kernels.  Usage: tools/gpu_pmc.sh [tag]
set -u
TAG="${1:-pmc}"
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  N=$(echo $C | tr ' ' '_')
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C -d "$OUT/$N" -o pmc -- python "$REPO/bench.py" --steps 40 --warmup 10 --no-cpu-baseline > "$OUT/$N.log" 2>&1)
  echo "pmc $C exit $?"
done
python "$REPO/tools/rocpd_pmc.py" "$OUT" | tee "$OUT/pmc_summary.md"
