#!/usr/bin/env bash
# round 6: the round-end checks in one gpurun call -- the whole GPU suite, the driver's bench command, smoke(), the other network shapes' step times
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time python -m pytest tests -m gpu -q ) > gpurun_out/gpu_suite.log 2>&1; grep -E "passed|failed|error|^real" gpurun_out/gpu_suite.log | tail -4
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r06.json 2> gpurun_out/bench_r06.err
python - <<'PY'
import json
j = json.load(open("gpurun_out/bench_r06.json")); r = j["roofline"]
print("bench", j["value"], j["ms_per_step"], j["ms_per_step_repeats"], "frac", r["frac"], "stale", r["traffic_stale"], "late", j["late_training"]["ms_per_step"],
      "occ", j["late_training_with_occupancy_skipping"]["ms_per_step"], "offline", j["offline_job"]["ms_per_step"], "T22", j["stress_T22"]["from_init"]["ms_per_step"],
      j["stress_T22"]["late"]["ms_per_step"], "multi", j["multi_object"]["value"])
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for sh in "16 1" "64 3" "128 2" "32 4" "16 4"; do python tools/shape_times.py $sh; done 2>&1 | grep "^{" | tee gpurun_out/shape_times.log
