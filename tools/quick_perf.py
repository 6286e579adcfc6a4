#!/usr/bin/env python3
"""Quick A/B numbers for kernel work: hippo batch (16.7 M rays), hippo 1080p render, optionally the
deep scene's batch. Prints one line per measurement.
    python tools/quick_perf.py [--deep] [--no-render] [--reps N]
Environment variables (RT_TUNE_*, RT_FAST_TREE, RT_AMD_LIBRARY) select the variant."""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import ray_tracer_amd as rta  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--deep", action="store_true")
ap.add_argument("--no-hippo", action="store_true")
ap.add_argument("--no-render", action="store_true")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--rays", type=int, default=1 << 24)
ap.add_argument("--sort", type=int, default=0, help="experiment: order the batch by an origin/direction key of this many bits per axis first")
ap.add_argument("--check", action="store_true", help="compare the fast walk's ids with the reference-order walk")
args = ap.parse_args()
dev = torch.device("cuda:0")
lib = rta._lib.lib()


def batch(bvh, name):
    n = args.rays
    rays = torch.from_numpy(rta.make_ray_batch(bvh, n)).to(dev)
    if args.sort:
        b = args.sort
        o = rays[:, :3]
        lo, hi = o.min(0).values, o.max(0).values
        q = ((o - lo) / (hi - lo) * ((1 << b) - 1e-6)).long()
        d = (rays[:, 3:] > 0).long()
        key = (d[:, 0] << 2 | d[:, 1] << 1 | d[:, 2])
        for a in range(3):
            key = key << b | q[:, a]
        rays = rays[torch.argsort(key)].contiguous()
    ids = torch.empty(n, dtype=torch.int32, device=dev)
    d = torch.empty(n, dtype=torch.float64, device=dev)
    call = lambda m: rta._lib.check(lib.rt_intersect_batch_device(bvh.scene(), rays.data_ptr(), n, m, ids.data_ptr(),
                                                                  d.data_ptr(), None, None, None))
    call(rta.TRAVERSE_FAST)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        call(rta.TRAVERSE_FAST)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.reps
    info = bvh.scene_info()
    line = f"{name} batch: {n / ms / 1e3:8.0f} Mrays/s  {ms:7.3f} ms  wide depth {info['fast_depth']} nodes {info['fast_nodes']}"
    if args.check:
        fast = ids.clone()
        fd = d.clone()
        call(rta.TRAVERSE_REFERENCE_ORDER)
        torch.cuda.synchronize()
        differ = int((fast != ids).sum())
        hit = ids != -1
        dd = int((fd[hit] != d[hit]).sum())
        line += f"  ids differing from reference order: {differ}, distances: {dd}"
    print(line, flush=True)


def render(bvh, name, spp=16, passes=3):
    film = rta.Film(bvh, 1920, 1080)
    film.set_camera(rta.reference_camera(), 1)
    film.render(0, spp, seed=1)
    torch.cuda.synchronize()
    film.clear()
    t0 = time.perf_counter()
    for k in range(passes):
        film.render(spp * k, spp, seed=1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = film.stats()
    print(f"{name} render: {st['rays'] / dt / 1e6:8.0f} Mrays/s  {1920 * 1080 * spp * passes / dt / 1e6:6.1f} Msamples/s  "
          f"traverse {st['traverse_ms'] / st['traverse_launches']:7.3f} ms/launch", flush=True)
    film.close()


if not args.no_hippo:
    bvh = rta.init_environment()
    batch(bvh, "hippo")
    if not args.no_render:
        render(bvh, "hippo")
    bvh.close()
if args.deep:
    t0 = time.perf_counter()
    bvh = rta.scenes.build(rta.scenes.subdivided_scene(4, 100_000))
    bvh.scene()
    print(f"deep scene built in {time.perf_counter() - t0:.1f} s", flush=True)
    batch(bvh, "deep")
    if not args.no_render:
        render(bvh, "deep", spp=8, passes=2)
    bvh.close()
