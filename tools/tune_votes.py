#!/usr/bin/env python3
"""Sweeps RT_TUNE_VOTES (exact,leaf,refill thresholds of the traversal vote) on the
intersection microbenchmark and one render pass; prints ms per launch for each setting."""
import itertools
import os
import subprocess
import sys

combos = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or \
    [(16, 16, 12), (8, 8, 8), (24, 24, 16), (32, 32, 16), (16, 32, 16), (32, 16, 16), (24, 16, 24), (1, 1, 1)]
code = r'''
import sys, time
sys.path.insert(0, ".")
import torch, ray_tracer_amd as rta
bvh = rta.init_environment()
n = 1 << 23
dev = torch.device("cuda:0")
rays = torch.from_numpy(rta.make_ray_batch(bvh, n)).to(dev)
ids = torch.empty(n, dtype=torch.int32, device=dev); dist = torch.empty(n, dtype=torch.float64, device=dev)
lib = rta._lib.lib()
def run():
    rta._lib.check(lib.rt_intersect_batch_device(bvh.scene(), rays.data_ptr(), n, 0, ids.data_ptr(), dist.data_ptr(), None, None, None))
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
film = rta.Film(bvh, 1920, 1080); film.set_camera(rta.reference_camera(), 1)
film.render(0, 1, seed=1); torch.cuda.synchronize(); film.clear()
t0 = time.perf_counter(); film.render(0, 4, seed=1); torch.cuda.synchronize(); t1 = time.perf_counter()
st = film.stats()
print("batch %.3f ms (%.0f Mrays/s)  render 4spp %.1f ms, traverse %.1f ms (%.0f Mrays/s in traversal)" % (ms, n / ms / 1e3, (t1 - t0) * 1e3, st["traverse_ms"], st["rays"] / st["traverse_ms"] / 1e3))
'''
for c in combos:
    env = dict(os.environ, RT_TUNE_VOTES="%d,%d,%d" % c)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    print(c, out.stdout.strip() or out.stderr.strip()[-300:], flush=True)
