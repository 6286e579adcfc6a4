#!/usr/bin/env python3
"""Random scenes, both device walks on the same rays: ids and distance bits must agree.
    python tools/fuzz_walks.py [--scenes 40] [--rays 200000] [--seed 0] [--circle] [--tiny]
Scenes mix spheres, quadrilaterals, pentagons and triangles at random sizes, counts and scales (1e-3 .. 1e3),
some with all primitives on one plane or in one cluster; rays come from the benchmark's generator (origins on a
shell around the scene and on its surfaces) plus origins inside the scene. --tiny: polygons 1e-5 .. 1e-2 of the
scene's size (the 11.9 M primitive scene's are 6e-5), and half of the rays aimed at points within a few f32 roundings
of an outline's edges and corners, up to grazing incidence: the band in which the margins of the fast walk's f32
screen and the pad of its boxes decide. Prints one line per scene."""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import numpy as np  # noqa: E402
import ray_tracer_amd as rta  # noqa: E402
from conftest import random_rays  # noqa: E402


def scene(rng, tiny=False, outlines=None):
    scale = 10.0 ** rng.uniform(-3, 3)
    centre = rng.uniform(-5, 5, 3) * scale
    flat = rng.random() < 0.2  # everything near one plane
    objs = []
    n_s = int(rng.integers(0, 60)) if rng.random() < 0.7 else 0
    for _ in range(n_s):
        c = rng.uniform(-1.5, 1.5, 3)
        if flat:
            c[2] *= 0.01
        objs.append(rta.Sphere(centre + c * scale, rng.uniform(0.02, 0.5) * scale, rta.Brdf(rng.uniform(0.2, 0.9, 3))))
    for n_side in (3, 4, 5):
        for _ in range(int(rng.integers(1, 400 if n_side == 3 else 60))):
            c = rng.uniform(-1.5, 1.5, 3)
            nrm = rng.standard_normal(3)
            if flat:
                c[2] *= 0.01
                nrm = np.array([0.0, 0.0, 1.0]) + 0.05 * rng.standard_normal(3)
            nrm /= np.linalg.norm(nrm)
            a = np.cross(nrm, [1.0, 0.0, 0.0])
            if np.linalg.norm(a) < 0.1:
                a = np.cross(nrm, [0.0, 1.0, 0.0])
            a /= np.linalg.norm(a)
            b = np.cross(nrm, a)
            rad = 10.0 ** rng.uniform(-5, -2) if tiny else rng.uniform(0.02, 0.6)
            ang = np.sort(rng.uniform(0, 2 * np.pi, n_side))
            pts = c + rad * (np.cos(ang)[:, None] * a + np.sin(ang)[:, None] * b)
            if n_side > 3 and rng.random() < 0.2:
                pts = pts + nrm * rng.uniform(-0.02, 0.02, (n_side, 1))  # warped: off its own plane
            e = 10.0 if rng.random() < 0.2 else 0.0
            objs.append(rta.Polygon.new_cached(centre + pts * scale, rta.Brdf(rng.uniform(0.2, 0.9, 3), e), nrm))
            if outlines is not None:
                outlines.append((centre + pts * scale, nrm))
    return objs, centre, scale


def edge_rays(rng, outlines, n, extent):
    """Rays aimed at points on, or within a few 1e-7 of the scene's extent of, the edges and corners of outlines."""
    rays = np.empty((n, 6))
    for r in range(n):
        pts, nrm = outlines[int(rng.integers(len(outlines)))]
        k = int(rng.integers(len(pts)))
        a, b = pts[k], pts[(k + 1) % len(pts)]
        w = rng.choice([0.0, 1.0, 0.5, rng.random()])
        edge = (b - a) / np.linalg.norm(b - a)
        out = np.cross(edge, nrm)
        target = a + w * (b - a) + out * rng.choice([0.0, 1.0, -1.0]) * 10.0 ** rng.uniform(-8, -5) * extent
        tilt = rng.choice([0.3, 1.0, 5.0, 50.0])  # tangent of the angle to the normal: up to grazing
        side = rng.standard_normal(3)
        side -= nrm * (side @ nrm)
        d = -nrm + tilt * side / np.linalg.norm(side)
        d /= np.linalg.norm(d)
        rays[r, :3] = target - d * 10.0 ** rng.uniform(-3, 0) * extent
        rays[r, 3:] = d
    return rays


def run(scenes, n_rays, seed=0, log=print, circle_bounds=False, tiny=False):
    """Number of scenes on which the two walks differ."""
    bad = 0
    for k in range(scenes):
        rng = np.random.default_rng(seed * 1000 + k)
        outlines = [] if tiny else None
        objs, centre, scale = scene(rng, tiny, outlines)
        tree = ("sah", "reference", "lbvh")[k % 3]
        bvh = rta.BVH.from_list(objs, fast_tree=tree, circle_bounds=circle_bounds)
        rays = np.vstack([rta.make_ray_batch(bvh, n_rays // 2, seed=k),
                          random_rays(n_rays // 4, seed=k, center=centre, spread=1.0 * scale, dist=1.0 * scale),
                          random_rays(n_rays // 4, seed=k + 1, center=centre, spread=0.5 * scale, dist=30.0 * scale)])
        if tiny:
            rays = np.vstack([rays[: n_rays // 2], edge_rays(rng, outlines, n_rays // 2, 10.0 * scale)])
        rid, rdist = bvh.intersect_batch(rays, mode=rta.TRAVERSE_REFERENCE_ORDER)
        fid, fdist = bvh.intersect_batch(rays, mode=rta.TRAVERSE_FAST)
        differ = int(((rid != fid) | (rdist.view(np.uint64) != fdist.view(np.uint64))).sum())
        info = bvh.scene_info()
        log(f"scene {k}: {len(objs)} primitives, scale {scale:.3g}, tree {tree}, wide depth {info['fast_depth']}, "
            f"hits {(rid != 0xFFFFFFFF).mean():.2f}, differing rays {differ}")
        bad += differ != 0
        bvh.close()
    return bad


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=40)
    ap.add_argument("--rays", type=int, default=200_000)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--circle", action="store_true", help="scenes built with RT_FAST_BOUNDS_CIRCLE")
    ap.add_argument("--tiny", action="store_true", help="small polygons and rays along their outlines")
    args = ap.parse_args()
    n_bad = run(args.scenes, args.rays, args.seed, log=lambda m: print(m, flush=True), circle_bounds=args.circle,
                tiny=args.tiny)
    print("scenes with differences:", n_bad)
    sys.exit(1 if n_bad else 0)
