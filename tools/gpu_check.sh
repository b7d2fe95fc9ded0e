#!/usr/bin/env bash
# Runs on the MI355X box via gpurun: GPU tests, smoke, short bench, rocprofv3 kernel stats.
# Usage: tools/gpu_check.sh [tag] [pytest-args...]
set -u
TAG="${1:-run}"; shift || true
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"
cd "$REPO"
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > "$OUT/device.txt"; nproc >> "$OUT/device.txt"
echo "== pytest gpu" ; timeout 1800 python -m pytest tests -m gpu -q --durations=8 --timeout 600 "$@" > "$OUT/pytest.log" 2>&1; echo "pytest exit $?" | tee -a "$OUT/pytest.log"; tail -25 "$OUT/pytest.log"
echo "== debug smoke"; for v in a; do timeout 120 python -u tools/debug_smoke.py $v > "$OUT/debug_smoke_$v.log" 2>&1; echo "variant $v exit $?"; tail -4 "$OUT/debug_smoke_$v.log"; done
echo "== smoke"; timeout 300 python -X faulthandler -u -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke exit $?" | tee -a "$OUT/smoke.log"; tail -5 "$OUT/smoke.log"
echo "== microbench"; timeout 300 python -u tools/run_microbench.py > "$OUT/microbench.log" 2>&1; tail -8 "$OUT/microbench.log"
for P in 4 16; do echo "== bench scatter P=$P"; MON_SCATTER_P=$P timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > "$OUT/bench_P$P.log" 2>&1; tail -1 "$OUT/bench_P$P.log" | cut -c1-400; done
echo "== bench graph"; MON_USE_GRAPH=1 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > "$OUT/bench_graph.log" 2>&1; tail -1 "$OUT/bench_graph.log" | cut -c1-400
echo "== bench atomics"; MON_LDS_SCATTER=0 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > "$OUT/bench_atomics.log" 2>&1; tail -1 "$OUT/bench_atomics.log"
echo "== bench0"; timeout 600 python bench.py --steps 200 --warmup 20 --backend 0 --no-cpu-baseline > "$OUT/bench_backend0.log" 2>&1; tail -1 "$OUT/bench_backend0.log"
echo "== bench"; timeout 900 python bench.py --steps 200 --warmup 20 > "$OUT/bench.log" 2>&1; echo "bench exit $?" | tee -a "$OUT/bench.log"; tail -3 "$OUT/bench.log"
echo "== rocprof"; export TMPDIR=/tmp; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o trace -- python "$REPO/bench.py" --steps 100 --warmup 10 --no-cpu-baseline > "$OUT/rocprof.log" 2>&1); echo "rocprof exit $?"
find "$OUT/prof" -name "*kernel_stats*" | head -3 | while read f; do echo "--- $f"; head -25 "$f"; done
