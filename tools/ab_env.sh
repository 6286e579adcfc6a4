#!/bin/bash
# tools/ab_env.sh <out-file> VAR value... -- lib...: quick_perf.py (3 repetitions) for every library (names as in
# tools/abn.sh) under every value of the environment variable VAR; everything into gpurun_out/<out-file>
out=gpurun_out/$1; var=$2; shift 2
vals=()
while [ "$1" != "--" ]; do vals+=("$1"); shift; done
shift
: > $out
for v in "${vals[@]}"; do
  for l in "$@"; do
    lib=ray-tracer_amd/librt_amd_$l.so
    [ "$l" = new ] && lib=ray-tracer_amd/librt_amd.so
    echo "== $var=$v $l" >> $out
    env $var=$v RT_AMD_LIBRARY=$lib timeout -k 10 200 python -u tools/quick_perf.py --reps 3 --check 2>&1 | grep -v amdgpu.ids >> $out || exit 1
  done
done
