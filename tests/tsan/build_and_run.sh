#!/usr/bin/env bash
# ThreadSanitizer build of the host side (every source of libmon_core.so compiled --offload-host-only: the kernels become launch stubs) linked against a stand-in
# HIP runtime (hip_stub.cpp), driven by tsan_driver.cpp.  No GPU needed.  Usage: tests/tsan/build_and_run.sh [build-dir]; exit status 0 = clean.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"; REPO="$HERE/../.."
OUT="${1:-/tmp/mon_tsan}"; mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-host-only -fsanitize=thread -g -O1 -std=c++17 -fPIC -x hip -ffp-contract=off -fno-math-errno -w)
SRCS=(config.cpp model.cpp c_api.cpp manager.cpp png_io.cpp mesh.cpp kernels_batch.hip kernels_net.hip kernels_net_wide.hip kernels_net_deep.hip kernels_layers.hip kernels_composite.hip kernels_optim.hip kernels_fused.hip kernels_scatter.hip kernels_render.hip kernels_tilerender.hip kernels_encode.hip kernels_step.hip kernels_bigscatter.hip kernels_mesh.hip)
pids=()
for s in "${SRCS[@]}"; do "$HIPCC" "${FLAGS[@]}" -c "$REPO/ro-map_amd/csrc/$s" -o "$OUT/${s%.*}.o" & pids+=($!); done
"$HIPCC" "${FLAGS[@]}" -c "$HERE/hip_stub.cpp" -o "$OUT/hip_stub.o" & pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
# the host objects reference their (absent) device images: define those symbols
{ for o in "$OUT"/*.o; do nm -u "$o"; done; } | grep -o "__hip_fatbin_[0-9a-f]*" | sort -u | awk '{ printf "char %s[8];\n", $1 }' > "$OUT/fatbin_syms.c"
gcc -c "$OUT/fatbin_syms.c" -o "$OUT/fatbin_syms.o"
objs=(); for s in "${SRCS[@]}"; do objs+=("$OUT/${s%.*}.o"); done
clang=/opt/rocm/lib/llvm/bin/clang++
"$clang" -fsanitize=thread -g -O1 -std=c++17 "$HERE/tsan_driver.cpp" "${objs[@]}" "$OUT/hip_stub.o" "$OUT/fatbin_syms.o" -o "$OUT/tsan_driver" -lz -lpthread -ldl
cd "$REPO"
set +e
TSAN_OPTIONS="halt_on_error=0 exitcode=66 report_signal_unsafe=0" "$OUT/tsan_driver" ro-map_amd/configs/base.json 2> "$OUT/tsan.log" | tail -3
rc=${PIPESTATUS[0]}
set -e
n=$(grep -c "WARNING: ThreadSanitizer" "$OUT/tsan.log" || true); echo "tsan reports: $n"
exit "$rc"
