#!/bin/bash
# A/B of two builds of the library on one GPU box: tools/ab.sh <out-prefix> [quick_perf args]
# runs ray-tracer_amd/librt_amd_base.so (a build of the commit to compare against) and then the
# current ray-tracer_amd/librt_amd.so through tools/quick_perf.py and writes gpurun_out/<prefix>_{base,new}.txt
set -e
p=$1; shift
RT_AMD_LIBRARY=ray-tracer_amd/librt_amd_base.so timeout -k 10 280 python -u tools/quick_perf.py "$@" > gpurun_out/${p}_base.txt 2>&1
timeout -k 10 280 python -u tools/quick_perf.py "$@" > gpurun_out/${p}_new.txt 2>&1
