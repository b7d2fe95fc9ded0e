#!/usr/bin/env bash
# The large-table optimizer's A/B switches of rounds 4-5 (chunk records | arrays, 16- | 32-bit step counters, chunk flags | table scan) are variant BUILDS
# since round 6 (model.h MON_VARIANT_*).  This runs the ORACLE tests of that path -- three whole steps, the 30-step run on the device's own gradients, the
# T = 2^22 step -- against each of them; build the variants first (cross-compiles without a GPU):
#   for v in "arrays -DMON_VARIANT_ARRAYS" "noflags -DMON_VARIANT_NO_FLAGS" "allold -DMON_VARIANT_STEPS32 -DMON_VARIANT_ARRAYS -DMON_VARIANT_NO_FLAGS"; do tools/variant_build.sh $v; done
# then on the GPU box:  bash tools/gpu_variants_large.sh   (summary -> gpurun_out/variants_large.log)
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p "$REPO/gpurun_out"; : > "$REPO/gpurun_out/variants_large.log"
for tag in arrays noflags allold; do
  lib="$REPO/ro-map_amd/build_$tag/libmon_core.so"
  [ -f "$lib" ] || { echo "$tag: not built" | tee -a "$REPO/gpurun_out/variants_large.log"; continue; }
  echo "== variant $tag" | tee -a "$REPO/gpurun_out/variants_large.log"
  MON_CORE_LIB="$lib" python -m pytest "$REPO/tests/test_gpu_parity.py" -m gpu -q -k "large_table or t22_matches or lazy_ema or binned_large" 2>&1 | tail -3 | tee -a "$REPO/gpurun_out/variants_large.log"
done
