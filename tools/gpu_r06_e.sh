#!/usr/bin/env bash
mkdir -p gpurun_out
python -m pytest tests/test_bench_multirank.py tests/test_rccl_gather.py -m gpu -q -x 2>&1 | tail -8 | tee gpurun_out/mr_test.log
python bench.py --steps 20 --warmup 5 --no-sustained > gpurun_out/bench_e.json 2> gpurun_out/bench_e.err; tail -c 600 gpurun_out/bench_e.json; tail -3 gpurun_out/bench_e.err
