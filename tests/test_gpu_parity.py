"""GPU parity tests: every call goes through the C ABI of librt_amd.so and is
checked against the CPU oracle on the same seeded inputs.

Bars: closest-hit primitive ids identical outside near-ties
(hits whose distances differ by < 1e-6 relative); distances within 1e-5 relative.
RT_TRAVERSE_REFERENCE_ORDER is held to the stricter bar of bit-identical ids,
distances, positions and normals.
"""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN, build_mixed, random_rays

pytestmark = pytest.mark.gpu

MISS = 0xFFFFFFFF
ID_TIE_BAND = 1e-6   # relative distance gap below which two candidate hits count as a tie
DIST_RTOL = 1e-5


def assert_hits_match(got_ids, got_dist, ref_ids, ref_dist, what):
    """ids exact outside the near-tie band, distances within DIST_RTOL."""
    both = (got_ids != MISS) & (ref_ids != MISS)
    assert np.array_equal(got_ids == MISS, ref_ids == MISS), f"{what}: hit/miss pattern differs"
    rel = np.abs(got_dist[both] - ref_dist[both]) / np.maximum(np.abs(ref_dist[both]), 1e-300)
    assert (rel <= DIST_RTOL).all(), f"{what}: distance off by {rel.max()}"
    diff = both & (got_ids != ref_ids)
    if diff.any():
        gap = np.abs(got_dist[diff] - ref_dist[diff]) / np.maximum(np.abs(ref_dist[diff]), 1e-300)
        assert (gap < ID_TIE_BAND).all(), f"{what}: {diff.sum()} id mismatches outside the near-tie band"
    return int(diff.sum())


def bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


# ---------------------------------------------------------------------------
# closest hit
# ---------------------------------------------------------------------------
def test_golden_hippo_hits(rta, hippo_bvh):
    g = np.load(GOLDEN / "hippo_hits_4k.npz")
    for mode in (rta.TRAVERSE_REFERENCE_ORDER, rta.TRAVERSE_FAST):
        ids, dist, pos, nrm = hippo_bvh.intersect_batch(g["rays"], mode=mode, want_pos=True)
        assert np.array_equal(ids, g["prim_id"])
        hit = ids != MISS
        assert np.array_equal(bits(dist[hit]), bits(g["distance"][hit]))
        assert np.array_equal(bits(pos[hit]), bits(g["pos"][hit]))
        assert np.array_equal(bits(nrm[hit]), bits(g["normal"][hit]))


def test_golden_mixed_hits(rta, mixed_bvh):
    g = np.load(GOLDEN / "mixed_hits_4k.npz")
    ids, dist, pos, nrm = mixed_bvh.intersect_batch(g["rays"], mode=rta.TRAVERSE_REFERENCE_ORDER, want_pos=True)
    hit = ids != MISS
    assert np.array_equal(ids, g["prim_id"])
    assert np.array_equal(bits(dist[hit]), bits(g["distance"][hit]))
    assert np.array_equal(bits(pos[hit]), bits(g["pos"][hit]))
    assert np.array_equal(bits(nrm[hit]), bits(g["normal"][hit]))
    ids2, dist2 = mixed_bvh.intersect_batch(g["rays"], mode=rta.TRAVERSE_FAST)
    assert_hits_match(ids2, dist2, g["prim_id"], g["distance"], "mixed fast")


def test_hippo_batch_vs_oracle(rta, hippo_bvh, hippo_oracle):
    """200k rays of the microbenchmark's generator: reference-order mode bit for bit,
    fast mode to the north-star bar (and, in practice, also bit for bit)."""
    rays = rta.make_ray_batch(hippo_bvh, 200_000, seed=0xABCDEF)
    ref_ids, ref_dist, ref_pos, ref_nrm = hippo_oracle.intersect_batch(rays, want_pos=True)
    ids, dist, pos, nrm = hippo_bvh.intersect_batch(rays, mode=rta.TRAVERSE_REFERENCE_ORDER, want_pos=True)
    hit = ref_ids != MISS
    assert np.array_equal(ids, ref_ids)
    assert np.array_equal(bits(dist[hit]), bits(ref_dist[hit]))
    assert np.array_equal(bits(pos[hit]), bits(ref_pos[hit]))
    assert np.array_equal(bits(nrm[hit]), bits(ref_nrm[hit]))
    ids, dist = hippo_bvh.intersect_batch(rays, mode=rta.TRAVERSE_FAST)
    ties = assert_hits_match(ids, dist, ref_ids, ref_dist, "hippo fast")
    assert ties == 0
    assert np.array_equal(bits(dist[hit]), bits(ref_dist[hit]))
    assert 0.3 < hit.mean() < 0.8


def test_lattice_aligned_rays_with_circle_bounds(rta, oracle):
    """Rays that are NOT in general position: a field of axis-aligned trapezoids (0,0) (1,0) (1,1) (0,1/2) and rays
    straight down whose y is level with the fourth vertex to the bit. geometry.rs:86,95 then skip the two edges
    that meet at that vertex, the edge x = 1 is the only crossing, and the reference reports a hit for every x up
    to 1 that lies inside the bounding circle (primitives.rs:231), also to the left of the trapezoid: a hit in thin
    air. Reference-order mode returns those hits; the fast mode returns them when the scene is built with
    RT_FAST_BOUNDS_CIRCLE, and the default fast mode, whose boxes bound the outlines, does not (shown here)."""
    items = []
    for a in range(8):
        for b in range(8):
            pts = np.array([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [1.0, 1.0, 0.0], [0.0, 0.5, 0.0]]) + [3.0 * a, 3.0 * b, 0.0]
            items.append(("p", pts, np.array([0.0, 0.0, 1.0]), (0.5, 0.5, 0.5), 0.0))
    orc = build_mixed(oracle, items, True)
    rng = np.random.default_rng(5)
    n = 8000
    a, b = rng.integers(0, 8, n), rng.integers(0, 8, n)
    x = 3.0 * a + rng.uniform(-0.4, 1.3, n)
    y = 3.0 * b + 0.5
    y[::5] += rng.uniform(-0.7, 0.7, len(y[::5]))  # some rays in general position too
    rays = np.stack([x, y, np.full(n, 3.0), np.zeros(n), np.zeros(n), -np.ones(n)], axis=1)
    ref_ids, ref_dist = orc.intersect_batch(rays)
    left_of_it = (x < 3.0 * a) & (ref_ids != MISS)
    assert left_of_it.sum() > 200  # hits in thin air, as the reference reports them
    plain = build_mixed(rta, items, False)
    ids, dist = plain.intersect_batch(rays, mode=rta.TRAVERSE_REFERENCE_ORDER)
    assert np.array_equal(ids, ref_ids) and np.array_equal(bits(dist), bits(ref_dist))
    ids, dist = plain.intersect_batch(rays, mode=rta.TRAVERSE_FAST)
    assert (ids != ref_ids).sum() > 200 and (ids[~left_of_it] == ref_ids[~left_of_it]).all()
    plain.close()
    wide = build_mixed(rta, items, False, circle_bounds=True)
    ids, dist = wide.intersect_batch(rays, mode=rta.TRAVERSE_FAST)
    assert np.array_equal(ids, ref_ids) and np.array_equal(bits(dist), bits(ref_dist))
    wide.close()


def test_self_hit_rule(rta, hippo_bvh, hippo_oracle):
    """Secondary-like rays start ON a polygon; the reference rejects that polygon because
    (o - p0).n <= 1e-8 (primitives.rs:217), with no epsilon push (tracer.rs:47)."""
    rays = rta.make_ray_batch(hippo_bvh, 20_000, seed=99)[1::2]
    ref_ids, ref_dist = hippo_oracle.intersect_batch(rays)
    for mode in (rta.TRAVERSE_REFERENCE_ORDER, rta.TRAVERSE_FAST):
        ids, dist = hippo_bvh.intersect_batch(rays, mode=mode)
        assert np.array_equal(ids, ref_ids)
        h = ids != MISS
        assert (dist[h] > 1e-7).all()


def test_mixed_scene_quirks(rta, mixed_bvh, mixed_oracle, mixed_items):
    """Origins inside spheres give hits BEHIND the origin (primitives.rs:75) and make the
    reference's result depend on its visiting order; reference-order mode reproduces that."""
    rng = np.random.default_rng(3)
    spheres = [it for it in mixed_items if it[0] == "s"]
    o = np.array([s[1] + s[2] * 0.5 * rng.standard_normal(3) / 2 for s in spheres for _ in range(50)])
    d = rng.standard_normal(o.shape)
    d /= np.linalg.norm(d, axis=1)[:, None]
    rays = np.vstack([np.hstack([o, d]), random_rays(20000, seed=8, center=(0, 0, 0), spread=1.2, dist=0.5)])
    ref_ids, ref_dist, ref_pos, _ = mixed_oracle.intersect_batch(rays, want_pos=True)
    ids, dist, pos, _ = mixed_bvh.intersect_batch(rays, mode=rta.TRAVERSE_REFERENCE_ORDER, want_pos=True)
    assert np.array_equal(ids, ref_ids)
    assert np.array_equal(bits(dist), bits(ref_dist))  # misses carry 0.0, NaNs compare by bits
    hit = ids != MISS
    assert np.array_equal(bits(pos[hit]), bits(ref_pos[hit]))
    # fast mode: it repeats in reference order every ray for which a sphere reported a hit behind the
    # origin or a NaN, so it returns the reference's answer everywhere, order-dependent cases included
    fids, fdist, fpos, _ = mixed_bvh.intersect_batch(rays, mode=rta.TRAVERSE_FAST, want_pos=True)
    assert np.array_equal(fids, ref_ids)
    assert np.array_equal(bits(fdist), bits(ref_dist))
    assert np.array_equal(bits(fpos[hit]), bits(ref_pos[hit]))
    # and the cases are really there: hits that lie BEHIND the ray's origin
    behind = hit & (np.einsum("ij,ij->i", ref_pos - rays[:, :3], rays[:, 3:]) < 0)
    assert behind.sum() > 100


def test_single_leaf_and_empty_and_ragged(rta, sphere_bvh, sphere_oracle):
    rays = random_rays(3001, seed=4, center=(0, 0, -0.7), spread=0.8, dist=2.5)
    ref_ids, ref_dist = sphere_oracle.intersect_batch(rays)
    for mode in (rta.TRAVERSE_REFERENCE_ORDER, rta.TRAVERSE_FAST):
        for n in (1, 63, 64, 65, 257, 3001):
            ids, dist = sphere_bvh.intersect_batch(rays[:n], mode=mode)
            assert np.array_equal(ids, ref_ids[:n]) and np.array_equal(bits(dist), bits(ref_dist[:n]))
        ids, dist = sphere_bvh.intersect_batch(np.zeros((0, 6)), mode=mode)
        assert ids.shape == (0,)
    assert {0, 1, 2} <= set(ref_ids.tolist())
    empty = rta.BVH.from_list([])
    for mode in (rta.TRAVERSE_REFERENCE_ORDER, rta.TRAVERSE_FAST):
        ids, _ = empty.intersect_batch(rays[:100], mode=mode)
        assert (ids == MISS).all()


def test_guard_leaf_scene(rta, oracle):
    """A leaf with more than 8 primitives (coincident centres) is split for the fast walk."""
    radii = np.linspace(0.05, 1.0, 20)
    s = oracle.Scene()
    for r in radii:
        s.add_sphere((0, 0, 0), r, oracle.make_brdf((1, 1, 1)))
    s.build()
    b = rta.BVH.from_list([rta.Sphere((0, 0, 0), r, rta.Brdf((1, 1, 1))) for r in radii])
    rays = random_rays(5000, seed=6, center=(0, 0, 0), spread=0.7, dist=3.0)
    ref_ids, ref_dist = s.intersect_batch(rays)
    for mode in (rta.TRAVERSE_REFERENCE_ORDER, rta.TRAVERSE_FAST):
        ids, dist = b.intersect_batch(rays, mode=mode)
        assert np.array_equal(ids, ref_ids) and np.array_equal(bits(dist), bits(ref_dist))


def test_exact_ties_go_to_the_first_visited(rta, oracle):
    """Duplicated and overlapping coplanar polygons hit at bit-identical distances; the
    reference keeps the one it visits first (strict '<', bvh.rs:116), and so must the fast
    walk although its own hierarchy stores the primitives in a different order."""
    rng = np.random.default_rng(21)
    quads = []
    for k in range(60):
        cx, cy = rng.uniform(-1, 1, 2)
        z = float(rng.integers(0, 3)) * 0.25
        h = rng.uniform(0.2, 0.6)
        pts = [[cx - h, cy - h, z], [cx + h, cy - h, z], [cx + h, cy + h, z], [cx - h, cy + h, z]]
        quads.append(pts)
        if k % 3 == 0:
            quads.append(pts)  # an exact duplicate
    s = oracle.Scene()
    objs = []
    for q in quads:
        s.add_polygon(q, oracle.make_brdf((0.5, 0.5, 0.5)), (0, 0, 1))
        objs.append(rta.Polygon.new_cached(q, rta.Brdf((0.5, 0.5, 0.5)), (0, 0, 1)))
    s.build()
    b = rta.BVH.from_list(objs)
    n = 20000
    o = np.column_stack([rng.uniform(-1.2, 1.2, n), rng.uniform(-1.2, 1.2, n), rng.uniform(1.0, 2.0, n)])
    d = np.column_stack([rng.normal(0, 0.2, n), rng.normal(0, 0.2, n), -np.ones(n)])
    d /= np.linalg.norm(d, axis=1)[:, None]
    rays = np.ascontiguousarray(np.hstack([o, d]))
    ref_ids, ref_dist = s.intersect_batch(rays)
    for mode in (rta.TRAVERSE_REFERENCE_ORDER, rta.TRAVERSE_FAST):
        ids, dist = b.intersect_batch(rays, mode=mode)
        assert np.array_equal(ids, ref_ids) and np.array_equal(bits(dist), bits(ref_dist))
    assert (ref_ids != MISS).mean() > 0.5
    # the duplicates make real ties: some winners have a twin at the same distance
    dup_of = {k + 1 for k in range(len(quads) - 1) if quads[k] is quads[k + 1]} | \
             {k for k in range(len(quads) - 1) if quads[k] is quads[k + 1]}
    assert np.isin(ref_ids[ref_ids != MISS], list(dup_of)).any()


def test_far_origins_and_axis_aligned_rays(rta, hippo_bvh, hippo_oracle):
    """Fast mode culls in f32: origins far from the scene and exactly axis-aligned
    directions (zero components -> infinite slab slopes) must not lose hits."""
    rng = np.random.default_rng(12)
    n = 20000
    tgt = np.array([0.0, 0.0, -0.6]) + 0.25 * rng.standard_normal((n, 3))
    d = rng.standard_normal((n, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    far = np.hstack([tgt - d * rng.choice([1e2, 1e4, 1e6], size=(n, 1)), d])
    axis = np.zeros((6000, 6))
    for k in range(6):
        blk = axis[k * 1000:(k + 1) * 1000]
        sign = 1.0 if k % 2 == 0 else -1.0
        blk[:, :3] = np.array([0.0, 0.0, -0.6]) + 0.3 * rng.standard_normal((1000, 3))
        blk[:, k // 2] = -3.0 * sign
        blk[:, 3 + k // 2] = sign
    rays = np.vstack([far, axis])
    ref_ids, ref_dist = hippo_oracle.intersect_batch(rays)
    ids, dist = hippo_bvh.intersect_batch(rays, mode=rta.TRAVERSE_FAST)
    assert_hits_match(ids, dist, ref_ids, ref_dist, "far/axis fast")
    assert np.array_equal(ids, ref_ids)
    assert (ref_ids != MISS).mean() > 0.3


def test_intersect_primitives(rta, mixed_bvh, mixed_oracle, oracle):
    """Primitive::intersect / normal_at, one (ray, primitive) pair per thread."""
    rays = random_rays(6000, seed=31, center=(0, 0, 0), spread=1.0, dist=4.0)
    ref_ids, _ = mixed_oracle.intersect_batch(rays)
    rng = np.random.default_rng(1)
    prim = np.where(ref_ids != MISS, ref_ids, rng.integers(0, 90, size=len(rays))).astype(np.uint32)
    prim[::3] = rng.integers(0, 90, size=len(prim[::3]))
    hit, pos, nrm = mixed_bvh.intersect_primitives(rays, prim)
    L = oracle.lib()
    p3, n3 = (C.c_double * 3)(), (C.c_double * 3)()
    n_hit = 0
    for i in range(len(rays)):
        o, d = oracle.vec(*rays[i, :3]), oracle.vec(*rays[i, 3:])
        ok = L.o_prim_intersect(mixed_oracle.prims[prim[i]], o, d, p3)
        assert bool(ok) == bool(hit[i])
        if ok:
            n_hit += 1
            L.o_prim_normal_at(mixed_oracle.prims[prim[i]], p3, n3)
            assert tuple(p3) == tuple(pos[i]) and tuple(n3) == tuple(nrm[i])
    assert n_hit > 1000
    with pytest.raises(rta.RtError):
        mixed_bvh.intersect_primitives(rays[:4], [0, 1, 2, 4000])


def test_rays_through_edges_and_vertices(rta, oracle):
    """Polygon::intersect where poly_inclusion_2d decides by a hair (geometry.rs:86-103): rays aimed at
    vertices, at points on the edges and at points level with a vertex, for outlines of 3 to 6 vertices.
    Each (ray, polygon) pair goes through the device's exact test and through the oracle's; hit flags
    and hit positions must agree to the bit, and both outcomes must occur."""
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
            # level with vertex i along the first edge's direction (basis_1 = norm(p1 - p0), primitives.rs:160)
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
    prim = np.asarray(prim, dtype=np.uint32)
    hit, pos, _ = bvh.intersect_primitives(rays, prim)
    L = oracle.lib()
    p3 = (C.c_double * 3)()
    want = np.zeros(len(rays), dtype=bool)
    for i in range(len(rays)):
        ok = L.o_prim_intersect(orc.prims[prim[i]], oracle.vec(*rays[i, :3]), oracle.vec(*rays[i, 3:]), p3)
        want[i] = bool(ok)
        if ok:
            assert tuple(p3) == tuple(pos[i]), i
    assert np.array_equal(want, hit.astype(bool)), int((want != hit.astype(bool)).sum())
    assert 0.15 < want.mean() < 0.85
    # The reference-order walk agrees with the oracle's walk on every one of these rays.
    ref_ids, ref_dist, ref_pos, _ = orc.intersect_batch(rays, want_pos=True)
    ids, dist = bvh.intersect_batch(rays, mode=rta.TRAVERSE_REFERENCE_ORDER)
    assert np.array_equal(ids, ref_ids) and np.array_equal(bits(dist), bits(ref_dist))
    # The fast walk does too, with one kind of exception that this batch provokes on purpose: when the
    # hit point is EXACTLY level with a vertex (same basis_2 coordinate), geometry.rs:86,95 skip the two
    # edges that meet there, and a point far outside the outline (but inside the bounding circle of
    # primitives.rs:231) can come out with an odd crossing count: the reference reports a hit in thin
    # air. The fast walk's boxes and screen bound the outline itself, so it does not see those.
    ids, dist = bvh.intersect_batch(rays, mode=rta.TRAVERSE_FAST)
    differ = np.nonzero((ids != ref_ids) | (bits(dist) != bits(ref_dist)))[0]
    assert len(differ) < 0.005 * len(rays)
    for i in differ:
        pts, nrm = items[ref_ids[i]][1], items[ref_ids[i]][2]
        centre = pts.mean(axis=0)
        outside = 0.0
        for a in range(len(pts)):
            p, q = pts[a], pts[(a + 1) % len(pts)]
            inward = np.cross(nrm, q - p)
            inward *= np.sign(np.dot(inward, centre - p)) / np.linalg.norm(inward)
            outside = max(outside, -np.dot(ref_pos[i] - p, inward))
        assert outside > 1e-6, (i, outside)  # the oracle's "hit" lies clearly outside the polygon it names
    assert len(differ) > 0  # the batch does provoke the case
    bvh.close()
    # With RT_FAST_BOUNDS_CIRCLE the fast walk bounds every polygon by the bounding circle that primitives.rs:231
    # tests against, and returns those hits as well: the reference's answer for every ray of the batch.
    wide = build_mixed(rta, items, False, circle_bounds=True)
    assert wide.scene_info()["fast_tree"] & 0x100
    ids, dist = wide.intersect_batch(rays, mode=rta.TRAVERSE_FAST)
    assert np.array_equal(ids, ref_ids) and np.array_equal(bits(dist), bits(ref_dist))
    wide.close()


def test_random_scenes_both_walks_agree(rta):
    """tools/fuzz_walks.py: random mixes of spheres, triangles, (warped) quadrilaterals and pentagons at scales
    from 1e-3 to 1e3, the three hierarchies in turn, rays from outside, from the surfaces and from inside: the fast
    walk returns the reference-order walk's primitive and distance bits on every ray."""
    import importlib.util
    import pathlib
    spec = importlib.util.spec_from_file_location("fuzz_walks", pathlib.Path(__file__).resolve().parent.parent / "tools" / "fuzz_walks.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lines = []
    assert mod.run(18, 60_000, seed=3, log=lines.append) == 0, "\n".join(lines)
    # polygons of 1e-5 .. 1e-2 of the scene's size, half of the rays aimed along their outlines up to grazing
    # incidence: where the margins of the f32 screen and the pad of the boxes decide
    assert mod.run(9, 40_000, seed=4, log=lines.append, tiny=True) == 0, "\n".join(lines)


def test_overlapping_launches_on_several_streams(rta, hippo_bvh):
    """Fast-walk launches on one scene from several streams at once: each launch draws its ranges of rays from
    a counter of its own, so the launches neither lose nor repeat a ray."""
    import torch
    dev = torch.device("cuda:0")
    n = 1 << 20
    rays = torch.from_numpy(rta.make_ray_batch(hippo_bvh, n, seed=5)).to(dev)
    lib = rta._lib.lib()
    want_ids = torch.empty(n, dtype=torch.int32, device=dev)
    want_dist = torch.empty(n, dtype=torch.float64, device=dev)
    rta._lib.check(lib.rt_intersect_batch_device(hippo_bvh.scene(), rays.data_ptr(), n, rta.TRAVERSE_REFERENCE_ORDER,
                                                 want_ids.data_ptr(), want_dist.data_ptr(), None, None, None))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(4)]
    outs = []
    for rep in range(3):
        for st in streams:
            ids = torch.full((n,), -7, dtype=torch.int32, device=dev)
            dist = torch.full((n,), -1.0, dtype=torch.float64, device=dev)
            st.wait_stream(torch.cuda.current_stream())
            rta._lib.check(lib.rt_intersect_batch_device(hippo_bvh.scene(), rays.data_ptr(), n, rta.TRAVERSE_FAST,
                                                         ids.data_ptr(), dist.data_ptr(), None, None, st.cuda_stream))
            outs.append((ids, dist))
    torch.cuda.synchronize()
    for ids, dist in outs:
        assert torch.equal(ids, want_ids) and torch.equal(dist, want_dist)


def test_device_pointer_entry_point(rta, hippo_bvh):
    """rt_intersect_batch_device on torch-owned HBM buffers and a non-default stream."""
    import torch
    g = np.load(GOLDEN / "hippo_hits_4k.npz")
    dev = torch.device("cuda:0")
    rays = torch.from_numpy(g["rays"]).to(dev)
    ids = torch.empty(len(rays), dtype=torch.int32, device=dev)
    dist = torch.empty(len(rays), dtype=torch.float64, device=dev)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        rta._lib.check(rta._lib.lib().rt_intersect_batch_device(
            hippo_bvh.scene(), rays.data_ptr(), len(rays), rta.TRAVERSE_FAST, ids.data_ptr(), dist.data_ptr(),
            None, None, stream.cuda_stream))
    stream.synchronize()
    got = ids.cpu().numpy().view(np.uint32)
    assert np.array_equal(got, g["prim_id"])
    hit = got != MISS
    assert np.array_equal(bits(dist.cpu().numpy()[hit]), bits(g["distance"][hit]))


def test_full_size_batch_properties(rta, hippo_bvh):
    """16,777,216 rays (the microbenchmark's size): properties that need no oracle.
    Both traversal flavours agree ray for ray; every hit lies on its primitive's plane,
    in front of the origin, at the reported distance."""
    import torch
    n = 1 << 24
    dev = torch.device("cuda:0")
    rays_h = rta.make_ray_batch(hippo_bvh, n)
    rays = torch.from_numpy(rays_h).to(dev)
    out = {}
    for mode in (rta.TRAVERSE_FAST, rta.TRAVERSE_REFERENCE_ORDER):
        ids = torch.empty(n, dtype=torch.int32, device=dev)
        dist = torch.empty(n, dtype=torch.float64, device=dev)
        pos = torch.empty((n, 3), dtype=torch.float64, device=dev)
        rta._lib.check(rta._lib.lib().rt_intersect_batch_device(
            hippo_bvh.scene(), rays.data_ptr(), n, mode, ids.data_ptr(), dist.data_ptr(), pos.data_ptr(), None,
            None))
        torch.cuda.synchronize()
        out[mode] = (ids, dist, pos)
    ids, dist, pos = out[rta.TRAVERSE_FAST]
    rids, rdist, rpos = out[rta.TRAVERSE_REFERENCE_ORDER]
    assert torch.equal(ids, rids)
    hit = ids != -1
    assert torch.equal(dist[hit], rdist[hit]) and torch.equal(pos[hit], rpos[hit])
    assert 0.3 < hit.double().mean().item() < 0.8
    # geometry of the hits
    seg = pos[hit] - rays[hit][:, :3]
    assert torch.allclose(seg.norm(dim=1), dist[hit], rtol=1e-12, atol=0)
    along = (seg * rays[hit][:, 3:]).sum(dim=1)
    assert (along > 0).all() and torch.allclose(along, dist[hit], rtol=1e-9)
    polys = hippo_bvh.flat()["polygons"]
    order = np.argsort(polys["prim_id"])
    normal = torch.from_numpy(polys["normal"][order]).to(dev)
    p0 = torch.from_numpy(polys["p0"][order]).to(dev)
    hid = ids[hit].long()
    off = ((pos[hit] - p0[hid]) * normal[hid]).sum(dim=1).abs()
    assert off.max().item() < 1e-12
    assert ((rays[hit][:, 3:] * normal[hid]).sum(dim=1) < 0).all()  # front faces only


# ---------------------------------------------------------------------------
# camera, tone map, integrator, film
# ---------------------------------------------------------------------------
def test_ray_bundle_vs_oracle(rta, oracle):
    """camera::ray_bundle: identical arithmetic, device libm for sqrt/sin/cos of the lens sample."""
    cam_o, cam = oracle.reference_camera(), rta.reference_camera()
    for (w, h) in ((32, 32), (48, 20), (7, 13)):
        ref = oracle.ray_bundle(cam_o, w, h, seed=5)
        got = rta.ray_bundle_cam(cam, w, h, seed=5)
        assert got.shape == ref.shape
        assert np.allclose(got, ref, rtol=0, atol=1e-15)
        assert (got == ref).mean() > 0.95
    # the reference's own signature: square resolution
    sq = rta.ray_bundle([0.0, -1.7, -0.65], [[0, 1, 0], [0, 0, 1], [1, 0, 0]], 2.0, 0.039, 0.05, 0.000001, 32, seed=5)
    assert np.array_equal(sq, rta.ray_bundle_cam(cam, 32, 32, seed=5))


def test_tone_map_vs_oracle(rta, oracle):
    """camera::spectral_radiance_to_color byte for byte. floor(255*f(x)) turns a last-bit difference of the
    power into a whole byte next to an integer, so the device evaluates pow(E, -2.5x) correctly rounded
    (double-double), as the host's libm does: the committed curve, 380,000 random radiances and, the hard part,
    radiances placed within a few ulps of every one of the 254 byte boundaries."""
    g = np.load(GOLDEN / "tone_curve.npz")
    x = g["x"]
    got = rta.spectral_radiance_to_color(np.stack([x, x, x], axis=1))
    assert (got[:, 3] == 255).all()
    assert np.array_equal(got[:, 0], g["u8"]) and np.array_equal(got[:, 1], g["u8"])
    assert tuple(rta.spectral_radiance_to_color([0.0, 1.0, 1e3])) == (0, 216, 255, 255)
    assert rta.spectral_radiance_to_color([float("nan"), -1.0, 0.0])[0] == 0
    assert tuple(rta.spectral_radiance_to_color([float("inf"), -float("inf"), 1e308])) == (255, 0, 255, 255)
    rng = np.random.default_rng(31)
    j = np.arange(1, 255, dtype=np.float64)
    edge = -np.log(2.0 / (1.0 + j / 255.0) - 1.0) / 2.5              # 255*f(edge) = j
    near = (edge[:, None] * (1.0 + np.arange(-40, 41)[None, :] * 2.0 ** -52)).ravel()
    xs = np.concatenate([near, rng.uniform(-1.0, 12.0, 300_000), rng.uniform(13.0, 16.0, 30_000),
                         10.0 ** rng.uniform(-12, 3, 30_000), -10.0 ** rng.uniform(-12, 3, 3_000)])
    xs = xs[: len(xs) // 3 * 3].reshape(-1, 3)
    want = np.array([oracle.tone_map(tuple(r)) for r in xs], dtype=np.uint8)  # the C oracle: libm's pow
    got = rta.spectral_radiance_to_color(xs)
    assert np.array_equal(got, want), int((got != want).any(axis=1).sum())


def test_trace_paths_vs_oracle(rta, hippo_bvh):
    """tracer::trace_ray with the same counter-based uniforms: radiance per path. Device
    sin/cos/acos differ from libm in the last bits, so directions agree to ~1e-15 and a
    rare path changes its hit sequence; the bar is per-path agreement for > 99.5 % of
    paths at 1e-9 relative and identical total energy within 1 %."""
    g = np.load(GOLDEN / "hippo_paths.npz")
    rays = np.vstack([g["bundle"].reshape(-1, 6)] * 2)
    st = rta.Stats()
    for mode in (rta.TRAVERSE_FAST, rta.TRAVERSE_REFERENCE_ORDER):
        rad = hippo_bvh.trace_paths(rays, g["keys"], seed=int(g["seed"]), mode=mode, stats=st)
        ref = g["radiance"]
        close = np.isclose(rad, ref, rtol=1e-9, atol=1e-12).all(axis=1)
        assert close.mean() > 0.995, close.mean()
        assert abs(rad.sum() - ref.sum()) <= 0.01 * ref.sum()
        assert abs(int(st.rays) - int(g["n_rays"])) <= 0.002 * int(g["n_rays"])
        assert st.paths == len(rays)
    # the reference's single-ray entry point
    one = rta.trace_ray(rays[0, :3], rays[0, 3:], hippo_bvh, seed=int(g["seed"]), pixel=int(g["keys"][0, 0]),
                        sample=int(g["keys"][0, 1]), sub_ray=int(g["keys"][0, 2]))
    assert np.allclose(one, g["radiance"][0], rtol=1e-9, atol=1e-12)


def test_emitter_first_fold(rta):
    """L = rho*(E + cos*L_next) (tracer.rs:63-78): first hit on an emitter, bounce escapes."""
    b = rta.BVH.from_list([rta.Polygon.new_cached([[-50, -50, 0], [50, -50, 0], [50, 50, 0], [-50, 50, 0]],
                                                  rta.Brdf((0.5, 0.25, 1.0), 4.0), (0, 0, 1))])
    st = rta.Stats()
    rad = b.trace_paths([[0, 0, 1, 0, 0, -1]], [[0, 0, 0]], seed=1, stats=st)
    assert tuple(rad[0]) == (2.0, 1.0, 4.0) and st.rays == 2
    dark = rta.BVH.from_list([rta.Polygon.new_cached([[2, 2, -1], [-2, 2, -1], [-2, -2, -1], [2, -2, -1]],
                                                     rta.Brdf((0.75, 0.95, 0.5)), (0, 0, 1))])
    assert tuple(dark.trace_paths([[0, 0, 1, 0, 0, -1]], [[0, 0, 0]], seed=1)[0]) == (0.0, 0.0, 0.0)


def render_rmse(a, b):
    return float(np.sqrt(np.mean((a - b) ** 2)))


@pytest.mark.parametrize("w,h", [(40, 24), (37, 23)])  # pixel counts: a multiple of 64, and not one
def test_film_render_vs_oracle_same_streams(rta, sphere_bvh, sphere_oracle, oracle, w, h):
    """Config 1 (sphere on the reference floor, emissive wall), small frame. With identical
    camera rays and identical random streams the GPU image equals the oracle's up to the
    rare paths that diverge on last-bit libm differences: per-pixel mean radiance RMSE
    below 2 % of the mean radiance, and > 97 % of RGBA bytes identical."""
    spp, seed = 8, 11
    cam_o, cam = oracle.reference_camera(), rta.reference_camera()
    bundle = oracle.ray_bundle(cam_o, w, h, seed)
    ref_mean, ref_rgba, ref_rays = sphere_oracle.render(cam_o, w, h, spp, seed, rays=bundle)
    film = rta.Film(sphere_bvh, w, h)
    film.set_rays(bundle)
    film.render(0, 3, seed=seed)
    film.render(3, spp - 3, seed=seed)  # split passes continue the same sample streams
    mean, rgba = film.resolve()
    st = film.stats()
    assert ref_mean.mean() > 0.05
    assert render_rmse(mean, ref_mean) < 0.02 * ref_mean.mean()
    assert (rgba == ref_rgba).mean() > 0.97
    assert np.isclose(mean, ref_mean, rtol=1e-9, atol=1e-12).all(axis=2).mean() > 0.97
    assert abs(st["rays"] - ref_rays) <= 0.002 * ref_rays and st["paths"] == w * h * 5 * spp
    # on-device camera: same image
    film2 = rta.Film(sphere_bvh, w, h)
    film2.set_camera(cam, seed)
    film2.render(0, spp, seed=seed)
    mean2, _ = film2.resolve()
    assert render_rmse(mean2, ref_mean) < 0.02 * ref_mean.mean()


def test_glossy_sphere_vs_oracle(rta, oracle):
    """Config 1 with the JavaScript tracer's variable-glossy lobe on the sphere and a glossy
    floor (geometry.js:172-184,300-318; the Rust crate is diffuse only, so this is checked
    against the oracle's restatement of the published algorithm): identical Philox blocks give
    the same lobe samples up to last-bit differences of device log/sin/cos."""
    from conftest import oracle_scene_from_objects
    objs = rta.scenes.sphere_on_ground(gloss=0.8)
    objs[1] = rta.Polygon.new_cached(objs[1].points, rta.Brdf.variable_glossy(0.8, (0.75, 0.95, 0.5)), (0, 0, 1))
    bvh = rta.scenes.build(objs)
    ref = oracle_scene_from_objects(oracle, objs)
    mats = bvh.flat()["materials"]
    assert sorted(round(float(g), 6) for g in mats["gloss"]) == [0.0, 0.2, 0.8]
    w, h, spp, seed = 40, 24, 8, 5
    cam_o = oracle.reference_camera()
    bundle = oracle.ray_bundle(cam_o, w, h, seed)
    # per path
    rays = np.vstack([bundle.reshape(-1, 6)] * 4)
    keys = np.array([[p, smp, j] for smp in range(4) for p in range(w * h) for j in range(5)], dtype=np.uint32)
    want, n_rays = ref.trace_batch(rays, keys, seed=seed)
    st = rta.Stats()
    got = bvh.trace_paths(rays, keys, seed=seed, stats=st)
    assert (want.sum(axis=1) > 0).sum() > 300  # most camera rays of this small frame leave the scene
    assert np.isclose(got, want, rtol=1e-9, atol=1e-12).all(axis=1).mean() > 0.99
    assert abs(int(st.rays) - n_rays) <= 0.005 * n_rays
    # per image
    ref_mean, ref_rgba, _ = ref.render(cam_o, w, h, spp, seed, rays=bundle)
    film = rta.Film(bvh, w, h)
    film.set_rays(bundle)
    film.render(0, spp, seed=seed)
    mean, rgba = film.resolve()
    assert render_rmse(mean, ref_mean) < 0.03 * ref_mean.mean()
    assert (rgba == ref_rgba).mean() > 0.95
    # the lobe changes the picture: a diffuse sphere renders differently
    diffuse = rta.scenes.build(rta.scenes.sphere_on_ground())
    film_d = rta.Film(diffuse, w, h)
    film_d.set_rays(bundle)
    film_d.render(0, spp, seed=seed)
    mean_d, _ = film_d.resolve()
    assert render_rmse(mean_d, ref_mean) > 0.1 * ref_mean.mean()
    with pytest.raises(RuntimeError, match="gloss"):
        rta.scenes.build([rta.Sphere((0, 0, 0), 1.0, rta.Brdf((1, 1, 1), gloss=1.5))]).scene()


def test_render_converges_to_oracle_image(rta, hippo_bvh, hippo_oracle, oracle):
    """Hippo scene at equal spp with DIFFERENT random streams: the two images are
    independent Monte Carlo estimates of the same integrand, so their per-pixel RMSE is
    about sqrt(2) x the per-image noise; noise is measured from two oracle renders."""
    w, h, spp = 32, 32, 16
    cam_o, cam = oracle.reference_camera(), rta.reference_camera()
    a, _, _ = hippo_oracle.render(cam_o, w, h, spp, seed=1)
    b, _, _ = hippo_oracle.render(cam_o, w, h, spp, seed=2)
    noise = render_rmse(a, b)
    got, rgba, st = rta.render(hippo_bvh, cam, w, h, spp, seed=3)
    assert noise > 0
    assert render_rmse(got, a) < 1.35 * noise and render_rmse(got, b) < 1.35 * noise
    assert abs(got.mean() - a.mean()) < 0.1 * a.mean()
    assert st["paths"] == w * h * 5 * spp and st["rays"] > st["paths"]


def test_config1_full_size_image_rmse(rta, sphere_bvh, sphere_oracle, oracle):
    """BASELINE config 1 at its stated size: one sphere on the reference floor lit by the reference's
    emissive wall, 512 x 512, 64 spp (main.rs:50-76, tracer.rs:8-81), GPU against the CPU oracle at equal
    spp with INDEPENDENT random streams. Stated thresholds: per-pixel RMSE of the mean-radiance images
    <= 1.35 x the oracle's own seed-to-seed RMSE (two independent estimates differ by sqrt(2) x the
    per-image noise, and so do two oracle renders), and frame-mean radiance within 1 %."""
    import os
    w = h = 512
    spp = 64
    threads = max(1, min(16, os.cpu_count() or 1))
    cam_o, cam = oracle.reference_camera(), rta.reference_camera()
    a, _, rays_a = sphere_oracle.render(cam_o, w, h, spp, seed=101, threads=threads)
    b, _, _ = sphere_oracle.render(cam_o, w, h, spp, seed=202, threads=threads)
    noise = render_rmse(a, b)
    got, rgba, st = rta.render(sphere_bvh, cam, w, h, spp, seed=303)
    ra, rb = render_rmse(got, a), render_rmse(got, b)
    rel_mean = abs(got.mean() - a.mean()) / a.mean()
    print(f"config 1, 512x512, 64 spp: oracle seed-to-seed RMSE {noise:.6e}; GPU vs oracle RMSE {ra:.6e} / {rb:.6e} "
          f"(ratio {ra / noise:.3f} / {rb / noise:.3f}, threshold 1.35); frame-mean radiance GPU {got.mean():.6f} "
          f"oracle {a.mean():.6f} (relative difference {rel_mean:.2e}, threshold 1e-2)")
    assert noise > 0 and a.mean() > 0.05
    assert ra <= 1.35 * noise and rb <= 1.35 * noise
    assert rel_mean <= 0.01 and abs(got.mean() - b.mean()) / b.mean() <= 0.01
    assert st["paths"] == w * h * 5 * spp
    assert abs(st["rays"] - rays_a) <= 0.005 * rays_a  # the same expected path length
    assert (rgba[..., 3] == 255).all()


def test_sharded_films_tile_the_frame(rta, hippo_bvh):
    """Tile sharding: the shards' frames are disjoint and their sum is bit-identical to the
    unsharded frame for any shard count (random streams are keyed by global pixel)."""
    spp, seed = 2, 5
    cam = rta.reference_camera()
    for w, h, shards, tile in ((80, 48, 2, 16), (80, 48, 3, 8), (80, 48, 8, 32), (75, 45, 3, 8), (61, 37, 2, 5)):
        full, full_rgba, _ = rta.render(hippo_bvh, cam, w, h, spp, seed=seed)
        acc = np.zeros_like(full)
        acc_rgba = np.zeros_like(full_rgba, dtype=np.uint32)
        covered = np.zeros((h, w), dtype=int)
        for r in range(shards):
            film = rta.Film(hippo_bvh, w, h, shard_index=r, shard_count=shards, tile_size=tile)
            film.set_camera(cam, seed)
            film.render(0, spp, seed=seed)
            mean, rgba = film.resolve()
            covered += (rgba[..., 3] == 255)
            acc += mean
            acc_rgba += rgba
            film.close()
        assert (covered == 1).all()
        assert np.array_equal(bits(acc), bits(full))
        assert np.array_equal(acc_rgba.astype(np.uint8), full_rgba)


def test_full_frame_properties(rta, hippo_bvh):
    """The headline frame size (1920x1080), where the oracle would take minutes: properties that
    hold at any size. Two tile shards sum to the unsharded frame bit for bit; a frame rendered as
    sample ranges 0..1 and 1..3 equals the frame rendered as 0..3 in one call (the fold adds paths
    in (sample, sub-ray) order either way); every pixel is covered; a second render is identical."""
    w, h, spp, seed = 1920, 1080, 3, 17
    cam = rta.reference_camera()
    film = rta.Film(hippo_bvh, w, h)
    film.set_camera(cam, seed)
    film.render(0, spp, seed=seed)
    full, full_rgba = film.resolve()
    st = film.stats()
    assert st["paths"] == w * h * 5 * spp and 2.5 < st["rays"] / st["paths"] < 4.0
    assert (full_rgba[..., 3] == 255).all() and np.isfinite(full).all() and full.min() >= 0.0
    # split sample ranges on the same film object
    film.clear()
    film.render(0, 1, seed=seed)
    film.render(1, 2, seed=seed)
    again, again_rgba = film.resolve()
    assert np.array_equal(bits(again), bits(full)) and np.array_equal(again_rgba, full_rgba)
    film.close()
    acc = np.zeros_like(full)
    for r in range(2):
        part = rta.Film(hippo_bvh, w, h, shard_index=r, shard_count=2, tile_size=32)
        part.set_camera(cam, seed)
        part.render(0, spp, seed=seed)
        mean, _ = part.resolve()
        acc += mean
        part.close()
    assert np.array_equal(bits(acc), bits(full))
    # the picture is the reference scene: lit floor, darker hippo, something in every colour channel
    assert full[..., 0].mean() > 0.05 and full[..., 1].mean() > 0.05 and full[..., 2].mean() > 0.02


def test_adversarial_triangles(rta):
    """The fast walk culls and screens in f32. Slivers, specks, big triangles, a scene far from the
    coordinate origin, and rays aimed exactly at edges and corners: the f32 stages may only ever
    pass a doubtful case on to the f64 test, so both modes must agree (outside exact-tie bands)."""
    rng = np.random.default_rng(77)
    for offset, scale in (((0.0, 0.0, 0.0), 1.0), ((700.0, -300.0, 150.0), 1.0), ((0.0, 0.0, 0.0), 1e-3)):
        off = np.array(offset)
        tris = []
        for k in range(1500):
            c = rng.uniform(-1, 1, 3)
            size = 10 ** rng.uniform(-4, 0.3)
            a = c + size * rng.normal(size=3)
            b = c + size * rng.normal(size=3)
            if k % 3 == 0:  # a sliver: third corner almost on the line a-b
                t = rng.uniform(0, 1)
                cc = a + t * (b - a) + 1e-3 * size * rng.normal(size=3)
            else:
                cc = c + size * rng.normal(size=3)
            tris.append(np.array([a, b, cc]) * scale + off)
        objs = []
        for p in tris:
            n = np.cross(p[1] - p[0], p[2] - p[1])
            if np.linalg.norm(n) < 1e-300:
                continue
            objs.append(rta.Polygon.new_cached(p, rta.Brdf((0.5, 0.5, 0.5)), n))
        bvh = rta.BVH.from_list(objs)
        m = 120_000
        which = rng.integers(0, len(tris), m)
        p = np.array(tris)[which]
        kind = rng.integers(0, 4, m)
        w = rng.dirichlet((1, 1, 1), m)
        w[kind == 1, 2] = 0.0                       # exactly on the edge p0-p1
        w[kind == 2] = np.eye(3)[rng.integers(0, 3, (kind == 2).sum())]  # exactly a corner
        w /= w.sum(axis=1, keepdims=True)
        target = np.einsum("mk,mkc->mc", w, p)
        nrm = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 1])
        nrm /= np.linalg.norm(nrm, axis=1)[:, None]
        d = -nrm + rng.normal(0, 0.8, (m, 3))       # from the front side, up to grazing
        d /= np.linalg.norm(d, axis=1)[:, None]
        dist = 10 ** rng.uniform(-3, 1, m) * scale
        o = target - d * dist[:, None]
        rays = np.ascontiguousarray(np.hstack([o, d]))
        ref_ids, ref_dist = bvh.intersect_batch(rays, mode=rta.TRAVERSE_REFERENCE_ORDER)
        ids, dd = bvh.intersect_batch(rays, mode=rta.TRAVERSE_FAST)
        assert_hits_match(ids, dd, ref_ids, ref_dist, f"offset {offset} scale {scale}")
        same = (ids == ref_ids) & (ids != MISS)
        assert np.array_equal(bits(dd[same]), bits(ref_dist[same]))
        assert (ids != MISS).mean() > 0.4


def test_warped_polygons(rta, oracle):
    """Quads and pentagons whose vertices do not lie in one plane. The reference puts the hit on the plane
    through the bounding centre, not through points[0] (primitives.rs:221-228), so such a polygon is hit
    off its outline's plane; an origin between the two planes is even mirrored by the |.| of
    primitives.rs:223. The fast walk's boxes and screen must not lose any of those hits: both walks
    against the oracle and against the scan over every primitive, rays aimed at edges at grazing angles."""
    # the worked example: the hit lies at x = 0.0005 on the z = 0.0025 plane, outside the z = 0 outline's box
    quad = [[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0.01]]
    one = rta.BVH.from_list([rta.Polygon.new_cached(quad, rta.Brdf((0.5, 0.5, 0.5)), (0, 0, 1))])
    ray = np.array([[0.998, 0.5, 1.0, -np.sqrt(0.5), 0.0, -np.sqrt(0.5)]])
    for mode in (rta.TRAVERSE_REFERENCE_ORDER, rta.TRAVERSE_FAST):
        ids, dist, pos, _ = one.intersect_batch(ray, mode=mode, want_pos=True)
        assert ids[0] == 0 and abs(pos[0, 2] - 0.0025) < 1e-12 and abs(pos[0, 0] - 0.0005) < 1e-9

    rng = np.random.default_rng(2024)
    polys = []
    for k in range(400):
        n_side = 4 if k % 2 == 0 else 5
        c = rng.uniform(-1, 1, 3)
        nrm = rng.normal(size=3)
        nrm /= np.linalg.norm(nrm)
        a = np.cross(nrm, [1.0, 0.0, 0.0])
        if np.linalg.norm(a) < 0.1:
            a = np.cross(nrm, [0.0, 1.0, 0.0])
        a /= np.linalg.norm(a)
        b = np.cross(nrm, a)
        rad = 10 ** rng.uniform(-1.5, -0.3)
        ang = np.sort(rng.uniform(0, 2 * np.pi, n_side))
        pts = c + rad * (np.cos(ang)[:, None] * a + np.sin(ang)[:, None] * b)
        warp = rad * 10 ** rng.uniform(-6, -0.5)        # from invisible to a third of the size
        pts = pts + warp * rng.normal(size=(n_side, 1)) * nrm
        polys.append((pts, nrm))
    brdf, obrdf = rta.Brdf((0.5, 0.5, 0.5)), oracle.make_brdf((0.5, 0.5, 0.5))
    bvh = rta.BVH.from_list([rta.Polygon.new_cached(p, brdf, n) for p, n in polys])
    ref = oracle.Scene()
    for p, n in polys:
        ref.add_polygon(p, obrdf, n)
    ref = ref.build()
    m = 150_000
    which = rng.integers(0, len(polys), m)
    o = np.empty((m, 3))
    d = np.empty((m, 3))
    for j, w in enumerate(which):
        pts, nrm = polys[w]
        e = rng.integers(0, len(pts))
        t = rng.uniform(0, 1)
        on_edge = pts[e] + t * (pts[(e + 1) % len(pts)] - pts[e])
        target = on_edge + (pts.mean(axis=0) - on_edge) * rng.choice([0.0, 1e-9, 1e-3, 0.3, -1e-3])
        dd = -nrm * 10 ** rng.uniform(-2.5, 0) + rng.normal(size=3)   # mostly grazing
        dd /= np.linalg.norm(dd)
        # some origins sit between the outline's plane and the centre plane
        back = 10 ** rng.uniform(-4, 0.5)
        o[j], d[j] = target - dd * back, dd
    rays = np.ascontiguousarray(np.hstack([o, d]))
    o_ids, o_dist = ref.intersect_batch(rays)
    r_ids, r_dist = bvh.intersect_batch(rays, mode=rta.TRAVERSE_REFERENCE_ORDER)
    assert np.array_equal(r_ids, o_ids) and np.array_equal(bits(r_dist), bits(o_dist))
    f_ids, f_dist = bvh.intersect_batch(rays, mode=rta.TRAVERSE_FAST)
    assert np.array_equal(f_ids, r_ids) and np.array_equal(bits(f_dist), bits(r_dist))
    b_ids, b_dist = bvh.brute_force_batch(rays)
    assert np.array_equal(b_ids == MISS, f_ids == MISS) and np.array_equal(bits(b_dist), bits(f_dist))
    assert 0.2 < (f_ids != MISS).mean() < 0.95


def test_tangent_rays_on_spheres(rta):
    """Rays that graze spheres at exactly their radius (give or take a few ulps), spheres from
    1e-3 to 1 in radius, also far from the coordinate origin: the f32 sphere screen must hand every
    doubtful ray to the f64 test (primitives.rs:45-81)."""
    rng = np.random.default_rng(78)
    for offset in ((0.0, 0.0, 0.0), (500.0, 200.0, -300.0)):
        off = np.array(offset)
        n_s = 800
        c = rng.uniform(-1, 1, (n_s, 3)) + off
        r = 10 ** rng.uniform(-3, 0, n_s)
        bvh = rta.BVH.from_list([rta.Sphere(c[k], r[k], rta.Brdf((1, 1, 1))) for k in range(n_s)])
        m = 100_000
        which = rng.integers(0, n_s, m)
        d = rng.normal(size=(m, 3))
        d /= np.linalg.norm(d, axis=1)[:, None]
        u = np.cross(d, rng.normal(size=(m, 3)))
        u /= np.linalg.norm(u, axis=1)[:, None]
        miss_by = rng.choice([0.0, 1e-15, -1e-15, 1e-9, -1e-9, 1e-4, -1e-4, -0.5], m)
        tangent = c[which] + u * (r[which] * (1.0 + miss_by))[:, None]
        o = tangent - d * rng.uniform(6.0, 12.0, m)[:, None]  # outside every sphere
        rays = np.ascontiguousarray(np.hstack([o, d]))
        ref_ids, ref_dist = bvh.intersect_batch(rays, mode=rta.TRAVERSE_REFERENCE_ORDER)
        ids, dd = bvh.intersect_batch(rays, mode=rta.TRAVERSE_FAST)
        assert_hits_match(ids, dd, ref_ids, ref_dist, f"offset {offset}")
        same = (ids == ref_ids) & (ids != MISS)
        assert np.array_equal(bits(dd[same]), bits(ref_dist[same]))
        assert (ids != MISS).mean() > 0.3


def test_walks_against_brute_force(rta, hippo_bvh, hippo_oracle, mixed_bvh, mixed_oracle):
    """Something that cannot miss a primitive: every ray against every primitive on the GPU
    (rt_intersect_brute_force). It equals the oracle's linear scan bit for bit, and on 1,048,576 hippo
    rays (4.8e10 primitive tests) both walks find exactly its hits: same hit/miss pattern, distances
    bit-identical, ids identical outside exact ties (the walks keep the first visited, the scan the
    first listed)."""
    small = rta.make_ray_batch(hippo_bvh, 3000, seed=41)
    o_ids, o_dist = hippo_oracle.brute_force_batch(small)
    g_ids, g_dist = hippo_bvh.brute_force_batch(small)
    assert np.array_equal(g_ids, o_ids) and np.array_equal(bits(g_dist), bits(o_dist))
    rays = rta.make_ray_batch(hippo_bvh, 1 << 20, seed=42)
    b_ids, b_dist = hippo_bvh.brute_force_batch(rays)
    for mode in (rta.TRAVERSE_FAST, rta.TRAVERSE_REFERENCE_ORDER):
        ids, dist = hippo_bvh.intersect_batch(rays, mode=mode)
        assert np.array_equal(ids == MISS, b_ids == MISS)
        hit = ids != MISS
        assert np.array_equal(bits(dist[hit]), bits(b_dist[hit]))
        assert (ids[hit] != b_ids[hit]).mean() < 1e-3  # only exact ties may name another primitive
    assert (b_ids != MISS).mean() > 0.4
    # spheres and polygons mixed, origins outside every sphere
    rays = random_rays(20_000, seed=43, center=(0, 0, 0), spread=1.0, dist=30.0)
    b_ids, b_dist = mixed_bvh.brute_force_batch(rays)
    ids, dist = mixed_bvh.intersect_batch(rays, mode=rta.TRAVERSE_FAST)
    assert_hits_match(ids, dist, b_ids, b_dist, "mixed scene vs brute force")


def test_directions_that_are_not_unit_vectors(rta, hippo_bvh):
    """The interface asks for unit directions, and the fast walk relies on them (entry distances against hit
    distances, the margins of its f32 stages). A ray whose direction is not a unit vector, or not a number, gets
    the reference-order walk's answer in fast mode too, to the bit, in every walk variant."""
    rng = np.random.default_rng(5)
    n = 30000
    tgt = np.array([0.0, 0.0, -0.6]) + 0.3 * rng.standard_normal((n, 3))
    o = np.array([1.5, -1.0, 0.5]) + 0.5 * rng.standard_normal((n, 3))
    d = tgt - o
    d /= np.linalg.norm(d, axis=1)[:, None]
    scale = rng.choice([1.0, 0.5, 2.0, 1.0 + 1e-3, 1.0 - 1e-5, 37.0, 1e-3], size=n)
    rays = np.hstack([o, d * scale[:, None]])
    rays[::997, 3] = np.nan
    rays[5::991, 5] = np.inf
    ref_ids, ref_dist = hippo_bvh.intersect_batch(rays, mode=rta.TRAVERSE_REFERENCE_ORDER)
    for env in ({}, {"RT_TWO_RAYS": "0"}, {"RT_WALK3": "3"}):
        os.environ.update(env)
        try:
            bvh = rta.init_environment() if env else hippo_bvh
            ids, dist = bvh.intersect_batch(rays, mode=rta.TRAVERSE_FAST)
        finally:
            for k in env:
                del os.environ[k]
        odd = scale != 1.0
        assert np.array_equal(ids[odd], ref_ids[odd]) and np.array_equal(bits(dist[odd]), bits(ref_dist[odd])), env
        assert_hits_match(ids[~odd], dist[~odd], ref_ids[~odd], ref_dist[~odd], f"unit rays {env}")
    assert (ref_ids[scale != 1.0] != MISS).mean() > 0.3
