#!/usr/bin/env bash
# Late in training (default: the stress configuration; CFG='{}' = base.json): kernel split of iterations 800..860 (the kernel trace covers all 860; the table shows min/avg/max).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/big_late; mkdir -p $OUT
(cd /tmp && MON_CRC_CFG="${CFG:-{\"log2_hashmap_size\": 22\}}" timeout 300 rocprofv3 --kernel-trace -d $OUT/prof -o st -- python $GRAFT_REPO_ROOT/tools/param_crc.py ${WARM:-800} 60 > $OUT/run.log 2>&1)
DB=$(find "$OUT/prof" -name "*_results.db" | head -1)
python - "$DB" <<'PY'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
per = {}
for n, s, e in rows:
    k = re.sub(r"\(.*", "", n); k = re.sub(r"^void ", "", k); k = k[:40]
    per.setdefault(k, []).append((e - s) / 1e3)
for k, v in per.items():
    tail = v[-60:]
    print("%-42s calls %5d   last-60 avg %8.2f us   min %8.2f   max %8.2f" % (k, len(v), sum(tail) / len(tail), min(tail), max(tail)))
PY
rm -rf $OUT/prof; tail -2 $OUT/run.log
