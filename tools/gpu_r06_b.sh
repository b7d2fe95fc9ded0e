#!/usr/bin/env bash
mkdir -p gpurun_out
python tools/_probe_stepvar.py 2>&1 | tee gpurun_out/stepvar.log
bash tools/variant_build.sh sctime -DMON_SCATTER_TIMING > /dev/null 2>&1
for v in 0 12288; do echo "== single_below=$v"; MON_OPTIONS="scatter_single_below=$v" python tools/scatter_timing.py; done 2>&1 | tee gpurun_out/sctime.log | grep -v "^\[\|^ \[" 
