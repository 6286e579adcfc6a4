#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "$1"; shift; env "$@" timeout -k 10 100 python3 tools/quick_perf.py --reps 3 2>&1 | grep -E "render" || exit 1; }
run "default sched (max-memory-clause)" X=1
run "no sched flag" RT_AMD_LIBRARY=ray-tracer_amd/librt_amd_nosched.so
run "max-ilp" RT_AMD_LIBRARY=ray-tracer_amd/librt_amd_ilp.so
run "node 4,4" RT_TUNE_NODE=4,4
run "node 12,8" RT_TUNE_NODE=12,8
run "node 16,10" RT_TUNE_NODE=16,10
run "ranges 16" RT_TUNE_RANGES=16
run "ranges 64" RT_TUNE_RANGES=64
run "votes paths 24,32,56" RT_TUNE_VOTES2_PATHS=24,32,56
run "votes paths 32,32,32" RT_TUNE_VOTES2_PATHS=32,32,32
run "votes paths 28,36,44" RT_TUNE_VOTES2_PATHS=28,36,44
run "votes paths 20,28,40" RT_TUNE_VOTES2_PATHS=20,28,40
