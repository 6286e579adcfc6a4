#!/bin/bash
# tools/abn.sh <out-file> <quick_perf args> -- lib1 lib2 ...: quick_perf.py with each named library in turn
# (names as in ray-tracer_amd/librt_amd_<name>.so; "new" = ray-tracer_amd/librt_amd.so), all into one file
set -e
out=gpurun_out/$1; shift
args=()
while [ "$1" != "--" ]; do args+=("$1"); shift; done
shift
: > $out
for v in "$@"; do
  lib=ray-tracer_amd/librt_amd_$v.so
  [ "$v" = new ] && lib=ray-tracer_amd/librt_amd.so
  echo "== $v" >> $out
  RT_AMD_LIBRARY=$lib timeout -k 10 200 python -u tools/quick_perf.py "${args[@]}" 2>&1 | grep -v amdgpu.ids >> $out
done
