#!/bin/bash
# Memory-side counter passes (FETCH_SIZE; WRITE_SIZE + L2 hits and misses) over tools/prof_workload.py for the
# current environment:   bash tools/prof_mem.sh <out dir under gpurun_out/> [scene]
set -e
NAME=${1:-mem}
SCENE=${2:-hippo}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$NAME
mkdir -p $O
cd $R
W="python3 tools/prof_workload.py --scene $SCENE --render 1 --batch 1"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $W > $O/fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_write -- $W > $O/write.log 2>&1
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_summary.txt
find $O -name "*.csv" -size +2M -delete
