#!/usr/bin/env bash
# A/B on the MI355X box: level-tile encode (default) against gathers inside k_fused_train (lds_encode=0); same CRC expected.
# Usage: tools/gpu_ab_encode.sh tag [MON_OPTIONS variants ...]   (each variant is run dense-only after the two reference runs)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/gpurun_out/${1:-ab_encode}"; mkdir -p "$OUT"; cd "$REPO"; shift || true
MON_OPTIONS=lds_encode=0 timeout 300 python tools/kernel_times.py 2>&1 | tail -1 | tee -a "$OUT/old.json"
timeout 300 python tools/kernel_times.py 2>&1 | tail -1 | tee -a "$OUT/new.json"
for v in "$@"; do echo "== $v"; MON_KT_DENSE_ONLY=1 MON_OPTIONS="$v" timeout 300 python tools/kernel_times.py 2>&1 | tail -1 | tee -a "$OUT/variants.json"; done
