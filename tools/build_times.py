#!/usr/bin/env python3
"""Scene-creation time and walk speed for the three hierarchies of RT_TRAVERSE_FAST."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import ray_tracer_amd as rta  # noqa: E402

levels = int(sys.argv[1]) if len(sys.argv) > 1 else 0
objs = rta.scenes.subdivided_scene(levels) if levels else None
for kind in ("sah", "lbvh", "reference"):
    bvh = rta.scenes.build(objs, fast_tree=kind) if objs else rta.init_environment()
    if not objs:
        bvh.fast_tree = rta.scene.FAST_TREES[kind]
    t0 = time.perf_counter()
    bvh.scene()
    dt = time.perf_counter() - t0
    info = bvh.scene_info()
    n = 1 << 22
    rays = torch.from_numpy(rta.make_ray_batch(bvh, n)).cuda()
    ids = torch.empty(n, dtype=torch.int32, device="cuda")
    d = torch.empty(n, dtype=torch.float64, device="cuda")
    L = rta._lib.lib()
    for rep in range(3):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        rta._lib.check(L.rt_intersect_batch_device(bvh.scene(), rays.data_ptr(), n, rta.TRAVERSE_FAST, ids.data_ptr(),
                                                   d.data_ptr(), None, None, None))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t1) * 1e3
    print(f"{kind:10s} prims {info['n_prims']:>9d}  rt_scene_create {dt:7.2f} s  fast depth {info['fast_depth']:3d}  "
          f"nodes {info['fast_nodes']:>9d}  used kind {info['fast_tree']}  {n / ms / 1e3:7.0f} Mrays/s "
          f"hits {int((ids != -1).sum())}", flush=True)
    bvh.close()
