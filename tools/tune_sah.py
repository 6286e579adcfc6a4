#!/usr/bin/env python3
"""Experiment driver: microbenchmark + short render for several builder / vote settings.
Each setting runs in a child process because the knobs are read from the environment."""
import json
import os
import subprocess
import sys

SETTINGS = [s for s in (sys.argv[1:] or ["0.7,0.3,1.0,8"])]
for s in SETTINGS:
    env = dict(os.environ)
    s, _, extra = s.partition("@")  # "@VAR=value,VAR=value": further environment knobs
    for kv in filter(None, extra.split(",")):
        k, _, v = kv.partition("=")
        env[k] = v
    sah, _, votes = s.partition("/")
    if sah:
        env["RT_TUNE_SAH"] = sah
    if votes:
        env["RT_TUNE_VOTES"] = votes
    out = subprocess.run([sys.executable, "bench.py", "--no-cpu", "--steps", "2"], capture_output=True, text=True, env=env)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print(f"{s + extra:44s} render {d['value']:7.0f}  micro {d['intersect_microbench']['fast']['mrays_per_s']:7.0f}", flush=True)
    except Exception as e:  # noqa: BLE001
        print(s, "failed", e, out.stderr[-300:], flush=True)
        break
