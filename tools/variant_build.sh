#!/usr/bin/env bash
# Builds a variant of libmon_core.so with extra compiler flags into ro-map_amd/build_<tag>/ (experiments / instrumentation).
# Usage: tools/variant_build.sh <tag> [-DFLAG ...]     then run anything with MON_CORE_LIB=ro-map_amd/build_<tag>/libmon_core.so
set -euo pipefail
TAG="$1"; shift
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/../ro-map_amd" && pwd)"
OBJ="$HERE/build_$TAG"; mkdir -p "$OBJ"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -ffp-contract=off -fno-math-errno -w "$@")
SRCS=(config.cpp model.cpp c_api.cpp manager.cpp png_io.cpp mesh.cpp kernels_batch.hip kernels_net.hip kernels_net_wide.hip kernels_net_deep.hip kernels_layers.hip kernels_composite.hip kernels_optim.hip kernels_fused.hip kernels_scatter.hip kernels_render.hip kernels_tilerender.hip kernels_encode.hip kernels_step.hip kernels_bigscatter.hip kernels_mesh.hip)
DIAG_SRCS=(diag.cpp diag_kernels.hip microbench.hip)
pids=()
for s in "${SRCS[@]}" "${DIAG_SRCS[@]}"; do /opt/rocm/bin/hipcc "${FLAGS[@]}" -I"$HERE/../include" -c "$HERE/csrc/$s" -o "$OBJ/${s%.*}.o" & pids+=($!); done
for p in "${pids[@]}"; do wait "$p"; done
objs=(); for s in "${SRCS[@]}"; do objs+=("$OBJ/${s%.*}.o"); done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OBJ/libmon_core.so" "${objs[@]}" -lz -lpthread -ldl
dobjs=(); for s in "${DIAG_SRCS[@]}"; do dobjs+=("$OBJ/${s%.*}.o"); done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OBJ/libmon_core_diag.so" "${dobjs[@]}" -L"$OBJ" -lmon_core -Wl,-rpath,'$ORIGIN'
echo "built $OBJ/libmon_core.so"
