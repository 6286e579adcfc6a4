#!/bin/bash
# order in which a film lists its pixels (RT_TUNE_PIXEL_BLOCK): 1080p hippo render
out=gpurun_out/${1:-sweep_pixel_block.txt}; shift
: > $out
for v in "$@"; do
  echo "== pixel block $v" >> $out
  RT_TUNE_PIXEL_BLOCK=$v timeout -k 10 200 python -u tools/quick_perf.py --reps 3 --rays 65536 2>&1 | grep -v amdgpu.ids | grep render >> $out || exit 1
done
