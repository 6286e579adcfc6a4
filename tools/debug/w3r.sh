out=gpurun_out/w3r.txt; : > $out
for m in 0 12; do
  echo "== RT_WALK3=$m hippo" >> $out
  RT_WALK3=$m timeout -k 10 200 python -u tools/quick_perf.py --reps 3 --check 2>&1 | grep -v amdgpu.ids >> $out || exit 1
done
for m in 0 12; do
  echo "== RT_WALK3=$m deep" >> $out
  RT_WALK3=$m timeout -k 10 400 python -u tools/quick_perf.py --reps 3 --check --deep --no-hippo 2>&1 | grep -v amdgpu.ids >> $out || exit 1
done
