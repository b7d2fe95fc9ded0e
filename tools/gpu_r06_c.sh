#!/usr/bin/env bash
mkdir -p gpurun_out
for r in 1 2 3; do
  MON_CORE_LIB=ro-map_amd/build_lateepre/libmon_core.so python tools/kernel_times.py
  python tools/kernel_times.py
done 2>&1 | tee gpurun_out/kt_epre.log | cut -c1-420
