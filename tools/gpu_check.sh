#!/usr/bin/env bash
# Runs on the MI355X box via gpurun: what the driver runs at round end (GPU tests, smoke, bench), logs under gpurun_out/<tag>/.
# Usage: tools/gpu_check.sh [tag] [pytest-args...]
set -u
TAG="${1:-run}"; shift || true
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"
cd "$REPO"
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > "$OUT/device.txt"; nproc >> "$OUT/device.txt"
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q --durations=8 --timeout 600 "$@" > "$OUT/pytest.log" 2>&1; echo "pytest exit $?" | tee -a "$OUT/pytest.log"; grep -E "passed|failed|error" "$OUT/pytest.log" | tail -3
echo "== smoke"; timeout 300 python -X faulthandler -u -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke exit $?" | tee -a "$OUT/smoke.log"; tail -3 "$OUT/smoke.log"
echo "== loaded libraries"; timeout 120 python -c "
import __graft_entry__ as g, os
pkg = g.load_package(); pkg.lib()
print([l.split()[-1] for l in open('/proc/self/maps') if 'libmon' in l][:2])"
echo "== bench (default flags)"; timeout 900 python bench.py > "$OUT/bench.log" 2>&1; echo "bench exit $?"; grep "^{" "$OUT/bench.log" | tail -1 > "$OUT/bench.json"; cut -c1-600 "$OUT/bench.json"
