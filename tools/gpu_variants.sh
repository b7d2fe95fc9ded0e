#!/usr/bin/env bash
# A/B of kernel variants on the GPU box: tools/gpu_variants.sh <tag> "<name>:<-D flags>" ...   (variant libraries must have been built here, in the
# container, with tools/variant_build.sh <name> <flags> -- they travel with the snapshot).  Each variant runs tools/kernel_times.py twice.
set -u
TAG="$1"; shift
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"
for rep in 1 2; do
  for V in default "$@"; do
    if [ "$V" = default ]; then LIBV=""; else LIBV="$REPO/ro-map_amd/build_$V/libmon_core.so"; fi
    MON_CORE_LIB="$LIBV" timeout 300 python "$REPO/tools/kernel_times.py" 2>&1 | tail -1 | sed "s/^/$V /" | tee -a "$OUT/times.txt"
  done
done
