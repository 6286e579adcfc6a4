#!/bin/bash
# Everything profiles/rNN needs, on the GPU box: rocprofv3 kernel stats and PMC passes of the hippo workload
# (1080p render + 16.7 M-ray batch) and of the config-5 batch, then the counting build's visits per ray.
#   bash tools/prof_round.sh <prefix, e.g. r02>      -> gpurun_out/<prefix>_hippo, gpurun_out/<prefix>_deep
set -e
P=${1:-r02}
R=$GRAFT_REPO_ROOT
cd $R
bash tools/prof_passes.sh hippo ${P}_hippo 1
bash tools/prof_passes.sh deep ${P}_deep 0
COUNT=ray-tracer_amd/librt_amd_count.so
RT_AMD_LIBRARY=$COUNT timeout -k 10 200 python3 tools/prof_workload.py --render 1 --batch 0 > /dev/null 2> gpurun_out/${P}_hippo/visits_render.txt
RT_AMD_LIBRARY=$COUNT timeout -k 10 200 python3 tools/prof_workload.py --render 0 --batch 1 > /dev/null 2> gpurun_out/${P}_hippo/visits_batch.txt
RT_AMD_LIBRARY=$COUNT timeout -k 10 400 python3 tools/prof_workload.py --scene deep --render 0 --batch 1 > /dev/null 2> gpurun_out/${P}_deep/visits_batch.txt
echo prof_round done
