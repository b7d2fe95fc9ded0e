#!/usr/bin/env bash
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r06.json 2> gpurun_out/bench_r06.err; tail -c 300 gpurun_out/bench_r06.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
