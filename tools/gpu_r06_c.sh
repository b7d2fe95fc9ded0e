#!/usr/bin/env bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_step_variant.py -m gpu -q -k "30_steps or trains_the_grid" > gpurun_out/t2.log 2>&1; tail -5 gpurun_out/t2.log
