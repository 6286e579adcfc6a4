#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry points: rt_intersect_batch on the 16,777,216-ray
hippo batch (rays and results in pageable host memory) and rt_render of a 1080p frame."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import ray_tracer_amd as rta  # noqa: E402

bvh = rta.init_environment()
n = 1 << 24
rays = rta.make_ray_batch(bvh, n)
bvh.intersect_batch(rays[:4096])
for _ in range(2):
    t0 = time.perf_counter()
    st = rta.Stats()
    ids, dist = bvh.intersect_batch(rays, stats=st)
    dt = time.perf_counter() - t0
    print(f"rt_intersect_batch host buffers: {n / dt / 1e6:.0f} Mrays/s end to end ({dt * 1e3:.1f} ms; "
          f"kernel {st.traverse_ms:.1f} ms; {n * 48 / 1e6:.0f} MB in, {n * 12 / 1e6:.0f} MB out)")
t0 = time.perf_counter()
mean, rgba, st = rta.render(bvh, rta.reference_camera(), 1920, 1080, 16, seed=1)
dt = time.perf_counter() - t0
print(f"rt_render 1920x1080x16spp host frame: {st['rays'] / dt / 1e6:.0f} Mrays/s end to end ({dt * 1e3:.1f} ms)")
