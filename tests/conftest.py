import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = ROOT / "tests" / "golden"
ASSETS = ROOT / "ray-tracer_amd" / "assets"  # the hippo mesh ships with the package


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def rta():
    import ray_tracer_amd
    return ray_tracer_amd


@pytest.fixture(scope="session")
def hippo_oracle(oracle):
    return oracle.hippo_scene(ASSETS / "hippo.rtmesh")


@pytest.fixture(scope="session")
def hippo_bvh(rta):
    return rta.init_environment(ASSETS / "hippo.rtmesh")


def sphere_scene_specs():
    """Config 1: one sphere on the reference floor, lit by the reference's emissive wall."""
    x, y, z, h = 2.0, 2.0, -1.0, 0.5
    return dict(
        sphere=dict(center=(0.0, 0.0, -0.7), radius=0.3, rho=(0.9, 0.1, 0.1)),
        floor=dict(points=[[x, y, z], [-x, y, z], [-x, -y, z], [x, -y, z]], rho=(0.75, 0.95, 0.5), e=0.0,
                   dv=(0.0, 0.0, 1.0)),
        wall=dict(points=[[-x - 1.0, y, z + h], [-x - 1.0, -y, z + h], [-x - 1.0, -y, -z + h], [-x - 1.0, y, -z + h]],
                  rho=(1.0, 1.0, 1.0), e=25.0, dv=(1.0, 0.0, 0.0)),
    )


@pytest.fixture(scope="session")
def sphere_oracle(oracle):
    sp = sphere_scene_specs()
    s = oracle.Scene()
    s.add_sphere(sp["sphere"]["center"], sp["sphere"]["radius"], oracle.make_brdf(sp["sphere"]["rho"]))
    for k in ("floor", "wall"):
        s.add_polygon(sp[k]["points"], oracle.make_brdf(sp[k]["rho"], sp[k]["e"]), sp[k]["dv"])
    return s.build()


@pytest.fixture(scope="session")
def sphere_bvh(rta):
    sp = sphere_scene_specs()
    objs = [rta.Sphere(sp["sphere"]["center"], sp["sphere"]["radius"], rta.Brdf(sp["sphere"]["rho"]))]
    for k in ("floor", "wall"):
        objs.append(rta.Polygon.new_cached(sp[k]["points"], rta.Brdf(sp[k]["rho"], sp[k]["e"]), sp[k]["dv"]))
    return rta.BVH.from_list(objs)


def mixed_scene_objects(mod, seed=7, n_spheres=40, n_quads=30, n_pent=20, is_oracle=False):
    """A random scene with spheres, quads and pentagons inside an emissive box; same
    construction for the oracle (`mod` = oracle module) and the product (`mod` = package)."""
    rng = np.random.default_rng(seed)
    items = []
    for _ in range(n_spheres):
        items.append(("s", rng.uniform(-1.5, 1.5, 3), rng.uniform(0.05, 0.4), rng.uniform(0.2, 0.9, 3), 0.0))
    for n_side, count in ((4, n_quads), (5, n_pent)):
        for _ in range(count):
            c = rng.uniform(-1.5, 1.5, 3)
            nrm = rng.standard_normal(3)
            nrm /= np.linalg.norm(nrm)
            a = np.cross(nrm, [1.0, 0.0, 0.0])
            if np.linalg.norm(a) < 0.1:
                a = np.cross(nrm, [0.0, 1.0, 0.0])
            a /= np.linalg.norm(a)
            b = np.cross(nrm, a)
            rad = rng.uniform(0.1, 0.6)
            ang = np.sort(rng.uniform(0, 2 * np.pi, n_side))
            pts = c + rad * (np.cos(ang)[:, None] * a + np.sin(ang)[:, None] * b)
            e = 10.0 if rng.random() < 0.2 else 0.0
            items.append(("p", pts, nrm, rng.uniform(0.2, 0.9, 3), e))
    return items


def build_mixed(mod, items, oracle_side, **from_list_args):
    if oracle_side:
        s = mod.Scene()
        for it in items:
            if it[0] == "s":
                s.add_sphere(it[1], it[2], mod.make_brdf(it[3], it[4]))
            else:
                s.add_polygon(it[1], mod.make_brdf(it[3], it[4]), it[2])
        return s.build()
    objs = []
    for it in items:
        if it[0] == "s":
            objs.append(mod.Sphere(it[1], it[2], mod.Brdf(it[3], it[4])))
        else:
            objs.append(mod.Polygon.new_cached(it[1], mod.Brdf(it[3], it[4]), it[2]))
    return mod.BVH.from_list(objs, **from_list_args)


@pytest.fixture(scope="session")
def mixed_items():
    return mixed_scene_objects(None)


@pytest.fixture(scope="session")
def mixed_oracle(oracle, mixed_items):
    return build_mixed(oracle, mixed_items, True)


@pytest.fixture(scope="session")
def mixed_bvh(rta, mixed_items):
    return build_mixed(rta, mixed_items, False)


def random_rays(n, seed, center=(0.0, 0.0, -0.5), spread=0.6, dist=3.0):
    rng = np.random.default_rng(seed)
    o = rng.standard_normal((n, 3))
    o /= np.linalg.norm(o, axis=1)[:, None]
    o = o * dist + np.asarray(center)
    tgt = rng.standard_normal((n, 3)) * spread + np.asarray(center)
    d = tgt - o
    d /= np.linalg.norm(d, axis=1)[:, None]
    return np.hstack([o, d])


def oracle_scene_from_objects(oracle, objects):
    """Feeds the CPU oracle the same primitive list a package-side object list describes."""
    import ray_tracer_amd as rta
    from ray_tracer_amd import scenes
    s = oracle.Scene()

    def brdf(b):
        return oracle.make_brdf(b.reflection_spectrum, b.emissivity, b.inside_index_of_refraction, b.gloss)

    for o in objects:
        if isinstance(o, rta.Sphere):
            s.add_sphere(o.center, o.radius, brdf(o.brdf))
        elif isinstance(o, rta.Polygon):
            s.add_polygon(o.points, brdf(o.brdf), o.direc_vec)
        elif isinstance(o, rta.Mesh):
            v = o.vertices.copy()
            v[:, 2] += o.z_shift
            b = brdf(o.brdf)
            for f in o.faces:
                p = v[f]
                s.add_polygon(p, b, np.cross(p[1] - p[0], p[2] - p[1]))
        elif type(o).__name__ == "_ReferenceRoom":
            for pts, rho, e, dv in scenes.reference_room_specs():
                s.add_polygon(pts, oracle.make_brdf(rho, e), dv)
        else:
            raise TypeError(type(o))
    return s.build()
