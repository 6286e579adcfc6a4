"""Debug helper: the scene and rays of tests/test_gpu_parity.py::test_rays_through_edges_and_vertices,
walked by the oracle and by both device walks; prints where they differ."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]
import numpy as np
import oracle
import ray_tracer_amd as rta
from conftest import build_mixed

oracle.build()
rng = np.random.default_rng(77)
items = []
for n_side in (3, 4, 5, 6):
    for _ in range(40):
        c = rng.uniform(-1.0, 1.0, 3)
        nrm = rng.standard_normal(3)
        nrm /= np.linalg.norm(nrm)
        a = np.cross(nrm, [1.0, 0.0, 0.0])
        if np.linalg.norm(a) < 0.1:
            a = np.cross(nrm, [0.0, 1.0, 0.0])
        a /= np.linalg.norm(a)
        b = np.cross(nrm, a)
        ang = np.sort(rng.uniform(0, 2 * np.pi, n_side))
        pts = c + rng.uniform(0.2, 0.6) * (np.cos(ang)[:, None] * a + np.sin(ang)[:, None] * b)
        items.append(("p", pts, nrm, rng.uniform(0.2, 0.9, 3), 0.0))
bvh = build_mixed(rta, items, False)
orc = build_mixed(oracle, items, True)
rays, prim = [], []
for k, (_, pts, nrm, _, _) in enumerate(items):
    n = len(pts)
    targets = [pts[i] for i in range(n)]
    for i in range(n):
        p, q = pts[i], pts[(i + 1) % n]
        for w in (0.5, 0.25, rng.random()):
            targets.append(p + w * (q - p))
        e = (pts[1] - pts[0]) / np.linalg.norm(pts[1] - pts[0])
        targets.append(p + rng.uniform(-0.3, 0.3) * e)
    for t in targets:
        for _ in range(3):
            side = nrm if rng.random() < 0.9 else -nrm
            o = t + side * rng.uniform(0.5, 2.0) + rng.standard_normal(3) * 0.3
            d = t - o
            rays.append(np.concatenate([o, d / np.linalg.norm(d)]))
            prim.append(k)
rays = np.asarray(rays)
ref_ids, ref_dist = orc.intersect_batch(rays)
b_ids, b_dist = bvh.brute_force_batch(rays)
print("brute vs oracle ids differ:", int((b_ids != ref_ids).sum()))
for name, mode in (("reference order", rta.TRAVERSE_REFERENCE_ORDER), ("fast", rta.TRAVERSE_FAST)):
    ids, dist = bvh.intersect_batch(rays, mode=mode)
    bad = np.nonzero((ids != ref_ids) | (dist.view(np.uint64) != ref_dist.view(np.uint64)))[0]
    print(name, "differ:", len(bad))
    for i in bad[:8]:
        print("  ray", i, "aimed at", prim[i], "device", ids[i], dist[i], "oracle", ref_ids[i], ref_dist[i], "brute", b_ids[i], b_dist[i])
