#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 passes: a few render passes of the hippo scene at
1920x1080 (16 spp each) and a few launches of the 16,777,216-ray intersection batch.
    rocprofv3 --kernel-trace --stats -d OUT -- python3 tools/prof_workload.py [--render N] [--batch N]
"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
import ray_tracer_amd as rta  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--render", type=int, default=2, help="render passes of 16 spp")
ap.add_argument("--batch", type=int, default=2, help="launches of the ray batch per traversal flavour")
ap.add_argument("--rays", type=int, default=1 << 24)
ap.add_argument("--reference-order", action="store_true")
ap.add_argument("--scene", choices=("hippo", "hippo_spheres", "deep"), default="hippo")
ap.add_argument("--deep-levels", type=int, default=4)
args = ap.parse_args()

if args.scene == "hippo":
    bvh = rta.init_environment()
elif args.scene == "hippo_spheres":
    bvh = rta.scenes.build(rta.scenes.hippo_with_spheres())
else:
    bvh = rta.scenes.build(rta.scenes.subdivided_scene(args.deep_levels))
if args.render:
    film = rta.Film(bvh, 1920, 1080)
    film.set_camera(rta.reference_camera(), 1)
    for k in range(args.render):
        film.render(16 * k, 16, seed=1,
                    mode=rta.TRAVERSE_REFERENCE_ORDER if args.reference_order else rta.TRAVERSE_FAST)
    torch.cuda.synchronize()
    print(film.stats())
if args.batch:
    dev = torch.device("cuda:0")
    rays = torch.from_numpy(rta.make_ray_batch(bvh, args.rays)).to(dev)
    ids = torch.empty(args.rays, dtype=torch.int32, device=dev)
    dist = torch.empty(args.rays, dtype=torch.float64, device=dev)
    modes = [rta.TRAVERSE_REFERENCE_ORDER] if args.reference_order else [rta.TRAVERSE_FAST]
    for m in modes:
        for _ in range(args.batch):
            rta._lib.check(rta._lib.lib().rt_intersect_batch_device(bvh.scene(), rays.data_ptr(), args.rays, m,
                                                                    ids.data_ptr(), dist.data_ptr(), None, None, None))
    torch.cuda.synchronize()
    print("hits", int((ids != -1).sum()))
if args.render:
    film.close()
bvh.close()  # an instrumented library prints its visit counts here
