"""Seeded synthetic ray batches for the intersection microbenchmark.

Half of a batch is "primary-like": origins uniform on a sphere of twice the
root shell's radius around its centre, aimed at a uniform point inside the
shell. The other half is "secondary-like": origins on the surface of a
uniformly chosen primitive, directions uniform over that primitive's outward
hemisphere; these start exactly on a surface and exercise the rule that a ray
must not hit the polygon it leaves (primitives.rs:217). Directions are
normalised in f64. numpy's Philox bit generator makes the batch reproducible.
"""
import numpy as np

DEFAULT_SEED = 0x5EED0001


def _unit(rng, n):
    v = rng.standard_normal((n, 3))
    v /= np.linalg.norm(v, axis=1)[:, None]
    return v


def make_ray_batch(bvh, n, seed=DEFAULT_SEED):
    """Returns an (n, 6) float64 array of rays for the scene behind `bvh`."""
    rng = np.random.Generator(np.random.Philox(seed))
    flat = bvh.flat()
    shell = flat["nodes"]["shell"][0]
    c, r = shell[:3], float(shell[3])
    n_sec = n // 2 if (len(flat["polygons"]) + len(flat["spheres"])) else 0
    n_pri = n - n_sec
    rays = np.empty((n, 6), dtype=np.float64)

    # primary-like
    o = c + 2.0 * r * _unit(rng, n_pri)
    tgt = c + r * _unit(rng, n_pri) * np.cbrt(rng.random((n_pri, 1)))
    d = tgt - o
    d /= np.linalg.norm(d, axis=1)[:, None]
    if n_sec:
        rays[0::2] = np.hstack([o, d])  # ceil(n/2) rows
    else:
        rays[:] = np.hstack([o, d])

    # secondary-like: interleaved with the primary-like rays so that every wave sees both kinds
    if n_sec:
        polys, pts2d, sph = flat["polygons"], flat["points2d"], flat["spheres"]
        n_poly, n_sph = len(polys), len(sph)
        pick = rng.integers(0, n_poly + n_sph, size=n_sec)
        so = np.empty((n_sec, 3))
        sn = np.empty((n_sec, 3))
        is_poly = pick < n_poly
        if is_poly.any():
            p = polys[pick[is_poly]]
            k = is_poly.sum()
            # a point inside the fan triangle (p0, v_i, v_{i+1}) of the outline
            tri = rng.integers(1, np.maximum(p["n_points"] - 1, 2))
            off = p["points_offset"].astype(np.int64)
            a2, b2 = pts2d[off + tri], pts2d[off + tri + 1]
            u, v = rng.random(k), rng.random(k)
            flip = u + v > 1.0
            u[flip], v[flip] = 1.0 - u[flip], 1.0 - v[flip]
            xy = u[:, None] * a2 + v[:, None] * b2
            so[is_poly] = p["p0"] + xy[:, :1] * p["basis_1"] + xy[:, 1:] * p["basis_2"]
            sn[is_poly] = p["normal"]
        if (~is_poly).any():
            s = sph[pick[~is_poly] - n_poly]
            nn = _unit(rng, (~is_poly).sum())
            so[~is_poly] = s["center"] + s["radius"][:, None] * nn
            sn[~is_poly] = nn
        sd = _unit(rng, n_sec)
        below = np.einsum("ij,ij->i", sd, sn) < 0.0
        sd[below] -= 2.0 * np.einsum("ij,ij->i", sd[below], sn[below])[:, None] * sn[below]
        rays[1::2] = np.hstack([so, sd])
    return rays
