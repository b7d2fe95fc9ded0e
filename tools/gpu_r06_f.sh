#!/usr/bin/env bash
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -q -x ) > gpurun_out/gpu_suite.log 2>&1; grep -E "passed|failed|error|^real" gpurun_out/gpu_suite.log | tail -5
bash tools/gpu_variants_large.sh > /dev/null 2>&1; cat gpurun_out/variants_large.log
python tools/kernel_times.py | tee gpurun_out/kt_after_retire.log
