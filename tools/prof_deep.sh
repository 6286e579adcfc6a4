set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/deep
mkdir -p $O
cd $R
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/prof_workload.py --scene deep --render 1 --batch 2 > $O/stats.log 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 tools/prof_workload.py --scene deep --render 0 --batch 1 > $O/fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/write -- python3 tools/prof_workload.py --scene deep --render 0 --batch 1 > $O/write.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/sq -- python3 tools/prof_workload.py --scene deep --render 0 --batch 1 > $O/sq.log 2>&1
timeout -k 10 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/tcp -- python3 tools/prof_workload.py --scene deep --render 0 --batch 1 > $O/tcp.log 2>&1
python3 tools/pmc_summary.py $O/fetch $O/write $O/sq $O/tcp > $O/pmc_summary.txt
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O -name "*.csv" -size +3M -delete
