#!/usr/bin/env bash
# Round 5 probe: k_optimizer(i) next to a throw-away k_encode_tiles (option overlap: side stream / no barrier bit), step times + a kernel trace.
# The options exist in a VARIANT build only (the product library carries none of it): build it first, in the container,
#   tools/variant_build.sh ovl -DMON_OVERLAP_PROBE
# then on the GPU box:  gpurun -- 'bash tools/gpu_overlap_probe.sh'      (results: profiles/r05_probes.md section 1, HISTORY 7.9)
#   PROBE_SET / TRACE_SET: space-separated option strings ("-" = no options)
set -u
export MON_CORE_LIB="${MON_CORE_LIB:-${GRAFT_REPO_ROOT:-/root/repo}/ro-map_amd/build_ovl/libmon_core.so}"
[ -e "$MON_CORE_LIB" ] || { echo "build the probe variant first: tools/variant_build.sh ovl -DMON_OVERLAP_PROBE"; exit 1; }
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/gpurun_out/${PROBE_TAG:-r05_overlap_probe}"; mkdir -p "$OUT"; cd "$REPO"
export TMPDIR=/tmp
for o in ${PROBE_SET:-- overlap=1 overlap=4,enc_lds_kb=144 overlap=5,enc_lds_kb=144 overlap=6,enc_lds_kb=144 overlap=6 overlap=5 -}; do
  [ "$o" = "-" ] && o=""
  echo "== MON_OPTIONS=$o"; MON_OPTIONS="$o" MON_KT_DENSE_ONLY=${KT_DENSE_ONLY:-} timeout 200 python tools/kernel_times.py 2>&1 | tail -1
done | tee "$OUT/kernel_times.txt"
for o in ${TRACE_SET:-overlap=5,enc_lds_kb=144 overlap=6,enc_lds_kb=144}; do
  tag=$(echo "$o" | tr ',=' '__')
  (cd /tmp && MON_OPTIONS="$o" timeout 300 rocprofv3 --kernel-trace -d "$OUT/trace_$tag" -o trace -- python $REPO/tools/profile_window.py --warmup 5 --steps 20 > "$OUT/trace_$tag.log" 2>&1); tail -1 "$OUT/trace_$tag.log"
  python tools/rocpd_timeline.py "$OUT/trace_$tag" 10 2 | tee "$OUT/timeline_$tag.md"
  rm -rf "$OUT/trace_$tag"
done
