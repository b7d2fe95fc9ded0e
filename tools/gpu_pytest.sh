#!/usr/bin/env bash
# gpurun helper: pytest on the MI355X box with the log kept under gpurun_out/<tag>/ and pytest's own exit status.   tools/gpu_pytest.sh <tag> <pytest args...>
set -u
TAG="$1"; shift
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"; cd "$REPO"
timeout ${PYTEST_TIMEOUT:-1800} python -m pytest "$@" -q --durations=6 > "$OUT/pytest.log" 2>&1; rc=$?
grep -E "passed|failed|error|Error|assert|^E " "$OUT/pytest.log" | tail -${PYTEST_TAIL:-25}; echo "pytest exit $rc"; exit $rc
