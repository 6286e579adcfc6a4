#!/usr/bin/env python3
"""Like tune_sah.py, on the subdivided scene (3 levels: 3 M triangles + 100 k spheres)."""
import json
import os
import subprocess
import sys

for s in sys.argv[1:]:
    env = dict(os.environ)
    s, _, extra = s.partition("@")
    for kv in filter(None, extra.split(",")):
        k, _, v = kv.partition("=")
        env[k] = v
    sah, _, votes = s.partition("/")
    if sah:
        env["RT_TUNE_SAH"] = sah
    if votes:
        env["RT_TUNE_VOTES"] = votes
    out = subprocess.run([sys.executable, "bench.py", "--scene", "deep", "--deep-levels", "4", "--steps", "1", "--warmup", "1"],
                         capture_output=True, text=True, env=env)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print(f"{s + extra:40s} render {d['value']:7.0f}  micro {d['intersect_microbench']['fast']['mrays_per_s']:7.0f}", flush=True)
    except Exception as e:  # noqa: BLE001
        print(s, "failed", e, out.stderr[-300:], flush=True)
        break
