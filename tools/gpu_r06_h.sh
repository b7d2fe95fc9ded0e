#!/usr/bin/env bash
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -q -x ) > gpurun_out/gpu_suite.log 2>&1; grep -E "passed|failed|error|^real" gpurun_out/gpu_suite.log | tail -5
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r06.json 2> gpurun_out/bench_r06.err; python - <<'PY'
import json
j=json.load(open('gpurun_out/bench_r06.json'))
print(j['value'], j['ms_per_step'], j['ms_per_step_repeats'], j['roofline']['frac'], j['roofline']['traffic_stale'], j['late_training']['ms_per_step'], j['late_training_with_occupancy_skipping']['ms_per_step'], j['offline_job']['ms_per_step'])
PY
