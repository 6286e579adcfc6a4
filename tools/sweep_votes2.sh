#!/bin/bash
# vote thresholds of the two-rays-per-lane walk: hippo batch and 1080p render
cd $GRAFT_REPO_ROOT
for v in "32,32,24" "16,16,12" "24,24,16" "40,40,32" "48,48,32" "24,32,48" "32,40,56" "16,24,32" "40,32,16" "20,20,60"; do
  echo "votes2 $v"
  RT_TUNE_VOTES2=$v RT_TUNE_VOTES2_PATHS=$v timeout -k 10 100 python3 tools/quick_perf.py --reps 3 2>&1 | grep -E "batch|render" || exit 1
done
