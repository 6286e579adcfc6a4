#!/bin/bash
# Diagnostic PMC passes over the fast walk (one rocprofv3 process per counter group).
#   bash tools/prof_diag.sh <out dir under gpurun_out/> [prof_workload.py arguments]
set -e
NAME=${1:-diag}
shift || true
ARGS=${@:---render 1 --batch 1}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$NAME
mkdir -p $O
cd $R
W="python3 tools/prof_workload.py $ARGS"
pass() {
  n=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$n -- $W > $O/$n.log 2>&1
}
pass sqa SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM
pass sqb SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_THREAD_CYCLES_VALU
pass sqc SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
python3 tools/pmc_summary.py $O/sqa $O/sqb $O/sqc $O/tcp > $O/pmc_summary.txt
find $O -name "*.csv" -size +1M -delete
echo done
