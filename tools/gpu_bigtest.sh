#!/usr/bin/env bash
# Stress configuration (T = 2^22): rocprofv3 kernel split of the first STEPS steps and the bench line, per large-level scatter mode
# (option big_switch: 16384 = default binned-then-atomic, 0 = global atomics only, 1 = always binned).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for sw in ${SWITCHES:-16384 0}; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/big_$sw; mkdir -p $OUT
  (cd /tmp && MON_OPTIONS=big_switch=$sw MON_CRC_CFG='{"log2_hashmap_size": 22}' timeout 300 rocprofv3 --kernel-trace -d $OUT/prof -o st -- python $GRAFT_REPO_ROOT/tools/param_crc.py ${STEPS:-60} > $OUT/run.log 2>&1)
  echo "== big_switch=$sw"
  DB=$(find "$OUT/prof" -name "*_results.db" | head -1); python tools/rocpd_stats.py "$DB" $OUT/kernel_stats.md | head -16; rm -rf $OUT/prof
  MON_OPTIONS=big_switch=$sw timeout 300 python bench.py --log2-hashmap-size 22 --steps 200 --warmup 20 --no-cpu-baseline --objects-per-gpu 0 2>&1 | tail -1 > $OUT/bench.json
  python -c "
import sys,json; d=json.load(open('$OUT/bench.json')); print('bench', d['value'], d['ms_per_step'], d.get('late_training'))"
done
