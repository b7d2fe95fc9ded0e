#!/usr/bin/env bash
# writes one synthetic sequence, then runs the headless offline driver N times and reports runs with a NaN loss
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
ss = ge.load_tools()
sc = ss.make_scene(n_views=40, H=480, W=640, f=525.0, n_objects=8, seed=5)
ss.write_sequence(sc, "/tmp/oj_seq")
PY
f=0
for i in $(seq 1 ${N:-20}); do
  MON_OPTIONS=${OPTS:-} ${EXE:-./ro-map_amd/offline_nerf} ro-map_amd/configs/base.json /tmp/oj_seq 0 8 /tmp/oj_out > /tmp/oj_run.txt 2>&1
  if grep -q "nan" /tmp/oj_run.txt; then f=$((f+1)); echo "run $i: NaN"; grep "nan" /tmp/oj_run.txt | head -3; grep "^Id: 0 " /tmp/oj_run.txt | head -12; fi
done
echo "NaN runs: $f of ${N:-20} (OPTS=${OPTS:-})"
