#!/bin/bash
# quick_perf.py over a list of RT_TUNE_NODE settings ("iterations,drop sixteenths"); results into gpurun_out/sweep_node.txt
out=gpurun_out/sweep_node.txt
: > $out
for v in "$@"; do
  echo "== node $v" >> $out
  RT_TUNE_NODE=$v timeout -k 10 120 python -u tools/quick_perf.py 2>&1 | grep -v amdgpu.ids >> $out || exit 1
done
