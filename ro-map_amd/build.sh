#!/usr/bin/env bash
# Builds ro-map_amd/libmon_core.so for gfx950 (cross-compiles without a GPU).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
OBJ="$HERE/build"; mkdir -p "$OBJ"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -ffp-contract=off -fno-math-errno -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result)
SRCS=(config.cpp model.cpp c_api.cpp manager.cpp png_io.cpp mesh.cpp kernels_batch.hip kernels_net.hip kernels_net_wide.hip kernels_net_deep.hip kernels_layers.hip kernels_composite.hip kernels_optim.hip kernels_fused.hip kernels_scatter.hip kernels_render.hip kernels_tilerender.hip kernels_encode.hip kernels_step.hip kernels_bigscatter.hip kernels_mesh.hip)
DIAG_SRCS=(diag.cpp diag_kernels.hip microbench.hip)      # libmon_core_diag.so: test scaffolding and micro-benchmarks, not the product
pids=()
# every object depends on every header (frag_layout.h is the MFMA weight image shared by k_optimizer and k_fused_train: a partial rebuild must not mix layouts)
newest_hdr="$HERE/../include/mon_core.h"; for h in "$HERE"/csrc/*.h; do [[ "$h" -nt "$newest_hdr" ]] && newest_hdr="$h"; done
for s in "${SRCS[@]}" "${DIAG_SRCS[@]}"; do
  o="$OBJ/${s%.*}.o"
  if [[ ! -f "$o" || "$HERE/csrc/$s" -nt "$o" || "$newest_hdr" -nt "$o" || "$HERE/build.sh" -nt "$o" ]]; then
    "$HIPCC" "${FLAGS[@]}" -c "$HERE/csrc/$s" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
objs=(); for s in "${SRCS[@]}"; do objs+=("$OBJ/${s%.*}.o"); done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$HERE/libmon_core.so" "${objs[@]}" -lz -lpthread -ldl
dobjs=(); for s in "${DIAG_SRCS[@]}"; do dobjs+=("$OBJ/${s%.*}.o"); done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$HERE/libmon_core_diag.so" "${dobjs[@]}" -L"$HERE" -lmon_core -Wl,-rpath,'$ORIGIN'
# libmon_core_rccl.so: the in-process gather-to-root over RCCL (include/mon_core_rccl.h), written against the public boundary; the core does not depend on RCCL
"$HIPCC" --offload-arch=gfx950 -O2 -std=c++17 -fPIC -Wall -x hip -c "$HERE/csrc/rccl_gather.cpp" -o "$OBJ/rccl_gather.o"
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$HERE/libmon_core_rccl.so" "$OBJ/rccl_gather.o" -L"$HERE" -lmon_core -L/opt/rocm/lib -lrccl -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib
g++ -O2 -std=c++17 "$HERE/../tools/offline_nerf.cpp" -o "$HERE/offline_nerf" -L"$HERE" -lmon_core -ldl -lpthread -Wl,-rpath,'$ORIGIN' -Wl,-rpath-link,/opt/rocm/lib
echo "built $HERE/libmon_core.so, $HERE/libmon_core_diag.so, $HERE/libmon_core_rccl.so and $HERE/offline_nerf"
