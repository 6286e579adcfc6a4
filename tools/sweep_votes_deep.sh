#!/bin/bash
# quick_perf.py --deep over a list of "exact,leaf,refill" vote thresholds (batch and path launches alike);
# results into gpurun_out/sweep_votes_deep.txt
out=gpurun_out/sweep_votes_deep.txt
: > $out
for v in "$@"; do
  echo "== votes $v" >> $out
  RT_TUNE_VOTES=$v RT_TUNE_VOTES_PATHS=$v timeout -k 10 200 python -u tools/quick_perf.py --deep --no-hippo 2>&1 | grep -v amdgpu.ids >> $out || exit 1
done
