#!/usr/bin/env bash
# round 6: the evidence windows on the round's final kernels (dense / sparse / sparse + occupancy grid; T = 2^22 early and late)
cd $GRAFT_REPO_ROOT
bash tools/gpu_profile_window.sh r06_dense 0 > gpurun_out/r06_dense.log 2>&1; tail -3 gpurun_out/r06_dense.log
bash tools/gpu_profile_window.sh r06_sparse 800 > gpurun_out/r06_sparse.log 2>&1; tail -3 gpurun_out/r06_sparse.log
WINDOW_ARGS="--occupancy" bash tools/gpu_profile_window.sh r06_sparse_occ 800 > gpurun_out/r06_sparse_occ.log 2>&1; tail -3 gpurun_out/r06_sparse_occ.log
WINDOW_ARGS="--log2-hashmap-size 22" bash tools/gpu_profile_window.sh r06_T22_init 15 > gpurun_out/r06_T22_init.log 2>&1; tail -3 gpurun_out/r06_T22_init.log
WINDOW_ARGS="--log2-hashmap-size 22" bash tools/gpu_profile_window.sh r06_T22_late 795 > gpurun_out/r06_T22_late.log 2>&1; tail -3 gpurun_out/r06_T22_late.log
