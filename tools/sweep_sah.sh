#!/bin/bash
# quick_perf.py over a list of RT_TUNE_SAH settings ("leaf_base,per_prim,step,leaf_max"); results into gpurun_out/sweep_sah.txt
out=gpurun_out/sweep_sah.txt
: > $out
for v in "$@"; do
  echo "== sah $v" >> $out
  RT_TUNE_SAH=$v timeout -k 10 120 python -u tools/quick_perf.py --check 2>&1 | grep -v amdgpu.ids >> $out || exit 1
done
