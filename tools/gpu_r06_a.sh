#!/usr/bin/env bash
# round 6, call A: the new parity tests' log + late-window A/B of the single-partition scatter
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_step_variant.py -m gpu -q -k "30_steps or trains_the_grid" > gpurun_out/t1.log 2>&1
tail -3 gpurun_out/t1.log
for v in 0 12288 0 12288 6144 24576 1073741824; do
  echo "single_below=$v"; MON_OPTIONS="scatter_single_below=$v" python tools/kernel_times.py
done 2>&1 | tee gpurun_out/kt_single.log
