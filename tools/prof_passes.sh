#!/bin/bash
# rocprofv3 passes over tools/prof_workload.py on the GPU box: kernel stats, then one PMC pass per
# counter group (rocprofv3 does not split counters over passes).
#   bash tools/prof_passes.sh <scene: hippo|hippo_spheres|deep> <out dir under gpurun_out/> [render passes]
set -e
SCENE=${1:-hippo}
NAME=${2:-prof}
RENDER=${3:-1}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$NAME
mkdir -p $O
cd $R
W="python3 tools/prof_workload.py --scene $SCENE"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $W --render $((RENDER * 2)) --batch 2 > $O/stats.log 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $W --render $RENDER --batch 1 > $O/fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_write -- $W --render $RENDER --batch 1 > $O/write.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -- $W --render $RENDER --batch 1 > $O/sq.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F32 SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc_sq2 -- $W --render $RENDER --batch 1 > $O/sq2.log 2>&1
timeout -k 10 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_tcp -- $W --render $RENDER --batch 1 > $O/tcp.log 2>&1
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_sq2 $O/pmc_tcp > $O/pmc_summary.txt
python3 tools/make_traffic_json.py $O $O/hbm_traffic.json > /dev/null
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O -name "*.csv" -size +2M -delete
echo done
