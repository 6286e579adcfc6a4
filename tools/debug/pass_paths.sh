out=gpurun_out/pass_paths.txt; : > $out
for v in 44000000 92274688 180000000; do
  echo "== RT_TUNE_PASS_PATHS=$v" >> $out
  RT_TUNE_PASS_PATHS=$v timeout -k 10 200 python -u tools/quick_perf.py --reps 3 --rays 65536 2>&1 | grep render >> $out || exit 1
done
