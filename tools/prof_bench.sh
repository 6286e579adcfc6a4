#!/bin/bash
# The bench itself under rocprofv3's kernel trace, on the GPU box: gpurun_out/<prefix>_bench/kernel_stats.csv and the
# line the bench printed in that run (bench_under_profiler.json).   bash tools/prof_bench.sh <prefix, e.g. r03>
set -e
P=${1:-r03}
cd $GRAFT_REPO_ROOT
out=gpurun_out/${P}_bench
rm -rf $out && mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 8 --warmup 1 > $out/bench_under_profiler.json 2> $out/bench.err
cp "$(ls -t $(find $out/stats -name "*kernel_stats.csv") | head -1)" $out/kernel_stats.csv
echo prof_bench done
