#!/bin/bash
# tools/sweep_env_deep.sh VAR value...: quick_perf.py --deep --no-hippo with VAR set to each value in turn;
# results into gpurun_out/sweep_env_deep.txt
out=gpurun_out/sweep_env_deep.txt
var=$1; shift
: > $out
for v in "$@"; do
  echo "== $var=$v" >> $out
  env $var=$v timeout -k 10 200 python -u tools/quick_perf.py --deep --no-hippo 2>&1 | grep -v amdgpu.ids >> $out || exit 1
done
