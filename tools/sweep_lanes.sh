#!/bin/bash
# 1080p hippo render with one pass at a time against two passes in flight on two streams, for several splits of
# a CU's block slots between the two lanes' persistent walks.   bash tools/sweep_lanes.sh > gpurun_out/lanes.txt
cd $GRAFT_REPO_ROOT
echo "lanes=1"; RT_FILM_LANES=1 timeout -k 10 120 python3 tools/quick_perf.py --reps 3 2>&1 | grep render || exit 1
for sh in "3 2" "5 5" "4 4" "3 3" "4 2" "2 3" "4 1" "4 3"; do
  set -- $sh
  echo "lanes=2 share $1,$2"
  RT_TUNE_WALK_SHARE0=$1 RT_TUNE_WALK_SHARE1=$2 timeout -k 10 120 python3 tools/quick_perf.py --reps 3 2>&1 | grep render || exit 1
done
