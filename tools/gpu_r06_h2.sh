#!/usr/bin/env bash
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -q -x ) > gpurun_out/gpu_suite.log 2>&1; grep -E "passed|failed|error|^real" gpurun_out/gpu_suite.log | tail -5
python tools/kernel_times.py
