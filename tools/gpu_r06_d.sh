#!/usr/bin/env bash
mkdir -p gpurun_out
python -m pytest tests/test_occupancy.py -m gpu -q -x -s 2>&1 | tail -5 | tee gpurun_out/occ_test.log
for r in 1 2; do MON_KT_CFG='{"occupancy_skip":1}' python tools/kernel_times.py; done 2>&1 | tee gpurun_out/kt_occ.log
python tools/occ_timing.py 2>&1 | tee gpurun_out/occ_timing.log
