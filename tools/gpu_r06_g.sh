#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for sh in "16 1" "64 3"; do
  tag=$(echo $sh | tr ' ' 'x')
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o t -- python $GRAFT_REPO_ROOT/tools/shape_times.py $sh > $GRAFT_REPO_ROOT/gpurun_out/shape_$tag.log 2>&1)
  grep "^{" gpurun_out/shape_$tag.log
  db=$(find /tmp/prof_$tag -name "*.db" | head -1); python tools/rocpd_stats.py "$db" gpurun_out/shape_${tag}_stats.md > /dev/null 2>&1; head -19 gpurun_out/shape_${tag}_stats.md | cut -c1-120
done
