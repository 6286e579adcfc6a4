#!/bin/bash
# vote thresholds of the parked-ray walk ("switch,prep,stuck"): hippo batch and 1080p render
out=gpurun_out/${1:-sweep_w3.txt}; shift
: > $out
for v in "$@"; do
  echo "== votes3 $v" >> $out
  RT_WALK3=3 RT_TUNE_VOTES3=$v RT_TUNE_VOTES3_PATHS=$v timeout -k 10 200 python -u tools/quick_perf.py --reps 3 2>&1 | grep -v amdgpu.ids >> $out || exit 1
done
