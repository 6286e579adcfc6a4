"""The benchmark scenes, built through the same constructors the reference uses.

  sphere_on_ground     config 1: one sphere on the reference floor, lit by the reference's wall
  init_environment     config 2/3: the reference scene itself (scene.py, main.rs:117-170)
  hippo_with_spheres   config 4: the reference scene plus a handful of spheres
  subdivided_scene     config 5: a recursively subdivided mesh (millions of polygons) plus
                       many spheres: a deep tree whose node and primitive arrays no longer
                       fit in the caches

Each returns the list of objects (so that the CPU oracle can be fed the same primitives in
tests) and, via build(), a BVH.
"""
import numpy as np

from .scene import BVH, HIPPO_MESH, Brdf, Mesh, Polygon, Sphere, _ReferenceRoom, load_rtmesh

ROOM_X, ROOM_Y, ROOM_Z, ROOM_H = 2.0, 2.0, -1.0, 0.5


def reference_room_specs():
    """The four quads of main.rs:117-161 as (points, rho, emissivity, direc_vec)."""
    x, y, z, h = ROOM_X, ROOM_Y, ROOM_Z, ROOM_H
    return [
        ([[x, y, z], [-x, y, z], [-x, -y, z], [x, -y, z]], (0.75, 0.95, 0.5), 0.0, (0.0, 0.0, 1.0)),
        ([[-x - 1.0, y, z + h], [-x - 1.0, -y, z + h], [-x - 1.0, -y, -z + h], [-x - 1.0, y, -z + h]],
         (1.0, 1.0, 1.0), 25.0, (1.0, 0.0, 0.0)),
        ([[x, y, z], [x, -y, z], [x, -y, z + 3.0], [x, y, z + 3.0]], (0.95, 0.95, 0.95), 0.0, (-1.0, 0.0, 0.0)),
        ([[x, y, z], [-x, y, z], [-x, y, -z + 1.0], [x, y, -z + 1.0]], (0.95, 0.95, 0.95), 0.0, (0.0, -1.0, 0.0)),
    ]


def sphere_on_ground(gloss=0.0):
    """Config 1. The sphere rests on the floor (z = -1) in front of the reference camera.
    gloss = 0 is the Rust crate's material (uniform diffuse, brdf.rs:14-29); gloss > 0 gives the
    sphere the JavaScript tracer's variable-glossy lobe (geometry.js:172-184)."""
    specs = reference_room_specs()
    objs = [Sphere((0.0, 0.0, -0.7), 0.3, Brdf((0.9, 0.1, 0.1), gloss=gloss))]
    for pts, rho, e, dv in specs[:2]:  # floor + emissive wall
        objs.append(Polygon.new_cached(pts, Brdf(rho, e), dv))
    return objs


def sphere_specs(n_spheres, seed, radius=(0.05, 0.25)):
    """Seeded spheres inside the room: (centre, radius, rho) triples."""
    rng = np.random.Generator(np.random.Philox(seed))
    c = np.stack([rng.uniform(-1.8, 1.8, n_spheres), rng.uniform(-1.8, 1.8, n_spheres),
                  rng.uniform(-0.9, 1.5, n_spheres)], axis=1)
    r = rng.uniform(radius[0], radius[1], n_spheres)
    rho = rng.uniform(0.2, 0.95, (n_spheres, 3))
    return c, r, rho


def hippo_with_spheres(n_spheres=12, seed=4, mesh_path=None):
    """Config 4: floor, emissive wall, two walls, the hippo mesh and a few spheres."""
    v, f = load_rtmesh(mesh_path or HIPPO_MESH)
    objs = [_ReferenceRoom(), Mesh(v, f, Brdf((0.65, 0.65, 0.65)), z_shift=-0.717)]
    c, r, rho = sphere_specs(n_spheres, seed)
    for k in range(n_spheres):
        objs.append(Sphere(c[k], r[k], Brdf(rho[k])))
    return objs


def subdivide(vertices, faces, levels, jitter, seed):
    """Each triangle -> 4 per level (edge midpoints), every new midpoint displaced by a seeded
    jitter proportional to its edge length, so no two bounding centres coincide (more than 4
    coincident centres would make the reference's split rule recurse forever, bvh.rs:56-67).
    Returns an (n, 3, 3) array of triangle corners."""
    rng = np.random.Generator(np.random.Philox(seed))
    tri = np.asarray(vertices, dtype=np.float64)[np.asarray(faces, dtype=np.int64)]  # (n,3,3)
    for _ in range(levels):
        a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]

        def mid(p, q):
            m = 0.5 * (p + q)
            if jitter:
                m = m + jitter * np.linalg.norm(q - p, axis=1)[:, None] * rng.uniform(-1.0, 1.0, m.shape)
            return m

        ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
        tri = np.concatenate([np.stack([a, ab, ca], axis=1), np.stack([ab, b, bc], axis=1),
                              np.stack([ca, bc, c], axis=1), np.stack([ab, bc, ca], axis=1)], axis=0)
    return tri


def subdivided_scene(levels=4, n_spheres=100_000, seed=5, max_base_faces=None, mesh_path=None, jitter=0.02):
    """Config 5: the room, the hippo mesh subdivided `levels` times (46,202 x 4^levels triangles:
    11.8 M at 4 levels) and `n_spheres` small spheres. max_base_faces trims the base mesh (tests)."""
    v, f = load_rtmesh(mesh_path or HIPPO_MESH)
    if max_base_faces is not None:
        f = f[:max_base_faces]
    v = v.copy()
    v[:, 2] -= 0.717
    tri = subdivide(v, f, levels, jitter, seed)
    n = tri.shape[0]
    verts = tri.reshape(-1, 3)
    faces = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    objs = [_ReferenceRoom(), Mesh(verts, faces, Brdf((0.65, 0.65, 0.65)), z_shift=0.0)]
    if n_spheres:
        c, r, rho = sphere_specs(n_spheres, seed + 1, radius=(0.002, 0.02))
        # a handful of distinct materials keeps the material table small
        palette = [Brdf(tuple(p)) for p in np.round(rho[:16], 2)]
        for k in range(n_spheres):
            objs.append(Sphere(c[k], r[k], palette[k % len(palette)]))
    return objs


def build(objects, device=0, fast_tree="sah", circle_bounds=False):
    return BVH.from_list(objects, device=device, fast_tree=fast_tree, circle_bounds=circle_bounds)
