#!/bin/bash
# quick_perf.py over a list of "exact,leaf,refill" vote thresholds (applied to both the batch and the
# path tracer's launches); results into gpurun_out/sweep_votes.txt
out=gpurun_out/sweep_votes.txt
: > $out
for v in "$@"; do
  echo "== votes $v" >> $out
  RT_TUNE_VOTES=$v RT_TUNE_VOTES_PATHS=$v timeout -k 10 120 python -u tools/quick_perf.py 2>&1 | grep -v amdgpu.ids >> $out || exit 1
done
