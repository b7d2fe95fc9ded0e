#!/usr/bin/env bash
# rocprofv3 --kernel-trace --stats of the driver's bench command itself (all legs mixed; the per-regime windows are the ones to price)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o b -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-sustained > $GRAFT_REPO_ROOT/gpurun_out/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/bench_under_rocprof.err)
db=$(find /tmp/prof_bench -name "*.db" | head -1); python tools/rocpd_stats.py "$db" gpurun_out/r06_bench_kernel_stats.md > /dev/null 2>&1; head -12 gpurun_out/r06_bench_kernel_stats.md | cut -c1-150
