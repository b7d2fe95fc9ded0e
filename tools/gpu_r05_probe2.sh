#!/usr/bin/env bash
# Round 5, second probe call: XCD-hierarchical grid barrier (microbench 72) + fetch granularity of random 4-byte reads (modes 80-82) with their EA request counters.
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/gpurun_out/r05_probe2"; mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 300 python tools/gridbarrierbench.py 2>&1 | tee "$OUT/gridbarrier.txt"
timeout 600 python tools/fetchbench.py 2>&1 | tee "$OUT/fetchbench.txt"
for C in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum"; do
  tag=$(echo "$C" | tr ' ' '_')
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C -d "$OUT/$tag/p" -o pmc -- python $REPO/tools/fetchbench.py --quick > "$OUT/pmc_$tag.log" 2>&1); echo "pmc [$C] exit $?"
done
python tools/rocpd_pmc.py "$OUT" | grep -E "k_ub_fetch|counter" | tee "$OUT/fetch_pmc.md"
rm -rf "$OUT"/*/p
