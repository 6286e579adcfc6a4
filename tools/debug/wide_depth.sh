out=gpurun_out/wide_depth.txt; : > $out
for d in 10 11 12 13 14; do
  echo "== RT_TUNE_WIDE_DEPTH=$d walk3" >> $out
  RT_TUNE_WIDE_DEPTH=$d RT_WALK3=3 timeout -k 10 200 python -u tools/quick_perf.py --reps 3 2>&1 | grep -v amdgpu.ids >> $out || exit 1
  RT_TUNE_WIDE_DEPTH=$d RT_WALK3=3 RT_AMD_LIBRARY=ray-tracer_amd/librt_amd_count.so timeout -k 10 200 python3 tools/prof_workload.py --render 1 --batch 0 2>&1 >/dev/null | grep node_steps >> $out
done
echo "== RT_TUNE_WIDE_DEPTH=11 old" >> $out
RT_TUNE_WIDE_DEPTH=11 RT_WALK3=0 timeout -k 10 200 python -u tools/quick_perf.py --reps 3 2>&1 | grep -v amdgpu.ids >> $out
