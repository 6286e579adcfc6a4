import json, sys, subprocess, os
out = subprocess.run([sys.executable, "bench.py", "--no-cpu", "--steps", "2"], capture_output=True, text=True).stdout
d = json.loads(out.strip().splitlines()[-1])
print(os.environ.get("RT_AMD_LIBRARY", "default").split("/")[-1], os.environ.get("RT_FAST_TREE", "sah"), "render %.0f" % d["value"], "micro %.0f" % d["intersect_microbench"]["fast"]["mrays_per_s"], flush=True)
