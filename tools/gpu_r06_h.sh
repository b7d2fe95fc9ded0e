#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for sh in "16 1" "64 3"; do python tools/shape_times.py $sh; MON_OPTIONS=use_graph=1 python tools/shape_times.py $sh; done 2>&1 | grep "^{"
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_a -o t -- python $GRAFT_REPO_ROOT/tools/shape_times.py 16 1 > /dev/null 2>&1)
db=$(find /tmp/prof_a -name "*.db" | head -1); python tools/rocpd_stats.py "$db" gpurun_out/shape_16x1_stats.md > /dev/null 2>&1; head -18 gpurun_out/shape_16x1_stats.md | cut -c1-110
