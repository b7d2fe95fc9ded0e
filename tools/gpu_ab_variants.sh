#!/usr/bin/env bash
# Kernel-time A/B of variant libraries (tools/variant_build.sh) on the MI355X box: tools/gpu_ab_variants.sh tag lib_tag ...   ("default" = the shipped library)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/gpurun_out/${1:-ab_variants}"; mkdir -p "$OUT"; cd "$REPO"; shift || true
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = default ]; then MON_KT_DENSE_ONLY="${MON_KT_DENSE_ONLY:-1}" timeout 300 python tools/kernel_times.py 2>&1 | tail -1 | tee -a "$OUT/times.json"
    else MON_KT_DENSE_ONLY="${MON_KT_DENSE_ONLY:-1}" MON_CORE_LIB="ro-map_amd/build_$v/libmon_core.so" timeout 300 python tools/kernel_times.py 2>&1 | tail -1 | tee -a "$OUT/times.json"; fi
  done
done
