#!/usr/bin/env python3
"""Reference-order visit counts (nodes popped, polygons and spheres tested per ray) of the CPU
oracle on the two benchmark ray populations of the hippo scene and on the miniature of the deep
scene (96,000 subdivided triangles + 3,000 spheres, the fixture of tests/test_scenes.py; the oracle
cannot hold the 11.9 M primitive scene); written to profiles/reference_visit_counts.json.
CPU only and deterministic: the populations are seeded, so the counts are the same on any host.
bench.py recomputes them in its cpu_baseline leg and falls back to this file when that leg is
skipped (multi-GPU runs, --no-cpu)."""
import ctypes as C
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
sys.path.insert(0, str(ROOT / "tests"))
import oracle as O  # noqa: E402
import ray_tracer_amd as rta  # noqa: E402

SEED, W, H, SPP, STRIDE, T = 20261111, 1920, 1080, 16, 331, 8


def main():
    O.build()
    L = O.lib()
    v, f = O.load_rtmesh(ROOT / "ray-tracer_amd" / "assets" / "hippo.rtmesh")
    cam = O.reference_camera()
    rgba = np.zeros(W * H * 4, dtype=np.uint8)
    n_rays, st = C.c_uint64(0), O.Stats()
    L.o_time_render(O.dp(v), v.shape[0], O.u32p(f), f.shape[0], C.byref(cam), W, H, SPP, C.c_uint64(SEED), T, STRIDE,
                    rgba.ctypes.data_as(O.c_u8_p), C.byref(n_rays), C.byref(st))
    render = dict(nodes_per_ray=st.nodes_popped / n_rays.value, polys_per_ray=st.poly_tests / n_rays.value,
                  spheres_per_ray=st.sphere_tests / n_rays.value, rays=int(n_rays.value),
                  workload=f"hippo {W}x{H} {SPP} spp, pixel index % {STRIDE} < {T}, seed {SEED}")
    bvh = rta.init_environment()
    k = 1 << 17
    rays = rta.make_ray_batch(bvh, 1 << 20)[:k]  # first 131072 rays of the generator's first 1Mi
    ref = O.hippo_scene()
    st2 = O.Stats()
    ref.intersect_batch(rays, stats=st2)
    micro = dict(nodes_per_ray=st2.nodes_popped / k, polys_per_ray=st2.poly_tests / k,
                 spheres_per_ray=st2.sphere_tests / k, rays=k,
                 workload="first 131072 rays of make_ray_batch(hippo, 2^20, seed 0x5EED0001)")
    from conftest import oracle_scene_from_objects
    objs = rta.scenes.subdivided_scene(levels=3, n_spheres=3000, seed=5, max_base_faces=1500)
    mini = rta.scenes.build(objs)
    mini_ref = oracle_scene_from_objects(O, objs)
    mrays = rta.make_ray_batch(mini, k)
    st3 = O.Stats()
    mini_ref.intersect_batch(mrays, stats=st3)
    deep = dict(nodes_per_ray=st3.nodes_popped / k, polys_per_ray=st3.poly_tests / k,
                spheres_per_ray=st3.sphere_tests / k, rays=k,
                workload="deep-scene miniature (1,500 base faces subdivided 3x = 96,000 triangles + 3,000 spheres, "
                         "seed 5), make_ray_batch(miniature, 131072, seed 0x5EED0001)")
    out = ROOT / "profiles" / "reference_visit_counts.json"
    out.parent.mkdir(exist_ok=True)
    out.write_text(json.dumps({"render": render, "microbench": micro, "deep_miniature": deep}, indent=1) + "\n")
    print(out.read_text())


if __name__ == "__main__":
    main()
