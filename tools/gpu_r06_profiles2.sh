#!/usr/bin/env bash
cd $GRAFT_REPO_ROOT
WINDOW_ARGS="--occupancy" bash tools/gpu_profile_window.sh r06_sparse_occ 800 > gpurun_out/r06_sparse_occ.log 2>&1; head -12 gpurun_out/r06_sparse_occ/kernel_window.md | cut -c1-150
