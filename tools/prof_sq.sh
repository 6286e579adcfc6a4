#!/bin/bash
# One SQ counter pass (instruction counts, busy and wait cycles) over tools/prof_workload.py for the current
# environment (RT_WALK3, RT_TUNE_*, ...):   bash tools/prof_sq.sh <out dir under gpurun_out/> [scene]
set -e
NAME=${1:-sq}
SCENE=${2:-hippo}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$NAME
mkdir -p $O
cd $R
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -- python3 tools/prof_workload.py --scene $SCENE --render 1 --batch 1 > $O/sq.log 2>&1
python3 tools/pmc_summary.py $O/pmc_sq > $O/pmc_summary.txt
find $O -name "*.csv" -size +2M -delete
