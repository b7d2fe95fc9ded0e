#!/usr/bin/env bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_step_variant.py tests/test_gpu_mesh.py tests/test_tile_render.py -m gpu -q -x -k "other_fully_fused or trains_the_grid or mesh or render" 2>&1 | tail -5 | tee gpurun_out/layers_test.log
for sh in "16 1" "64 3" "128 2" "32 4" "16 4"; do python tools/shape_times.py $sh; done 2>&1 | grep "^{" | tee gpurun_out/shape_times.log
MON_OPTIONS=lds_encode=0 python tools/shape_times.py 64 3 | grep "^{"
