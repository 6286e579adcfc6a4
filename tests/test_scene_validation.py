"""rt_scene_create checks every link and range of a flattened scene before anything reaches a
kernel (a kernel that indexes outside its arrays can fault the whole GPU). The checks run before
the device is touched, so corrupt descriptions are rejected here on the CPU as well."""
import ctypes as C

import numpy as np
import pytest


RT_ERR_INVALID_ARGUMENT, RT_ERR_NO_DEVICE = 1, 2


def desc_from(rta, flat, fast_tree=0):
    """An rt_scene_desc over numpy arrays (kept alive by the returned list)."""
    L = rta._lib
    keep = [np.ascontiguousarray(flat[k]) for k in ("nodes", "slots", "polygons", "points2d", "spheres", "materials")]
    nodes, slots, polys, pts, sph, mats = keep
    d = L.SceneDesc()
    d.struct_size = C.sizeof(L.SceneDesc)
    d.n_nodes, d.n_slots, d.n_polygons = len(nodes), len(slots), len(polys)
    d.n_points2d, d.n_spheres, d.n_materials = len(pts), len(sph), len(mats)
    d.nodes = C.cast(nodes.ctypes.data, type(d.nodes))
    d.slots = C.cast(slots.ctypes.data, type(d.slots))
    d.polygons = C.cast(polys.ctypes.data, type(d.polygons))
    d.points2d = C.cast(pts.ctypes.data, type(d.points2d))
    d.spheres = C.cast(sph.ctypes.data, type(d.spheres))
    d.materials = C.cast(mats.ctypes.data, type(d.materials))
    d.fast_tree = fast_tree
    return d, keep


def create(rta, d):
    out = C.c_void_p()
    st = rta._lib.lib().rt_scene_create(C.byref(d), 0, C.byref(out))
    msg = rta._lib.lib().rt_last_error().decode()
    if st == 0:
        rta._lib.lib().rt_scene_destroy(out)
    return st, msg


@pytest.fixture(scope="module")
def flat(mixed_bvh):
    return mixed_bvh.flat()


def test_intact_description_passes_validation(rta, flat):
    d, keep = desc_from(rta, flat)
    st, msg = create(rta, d)
    assert st in (0, RT_ERR_NO_DEVICE), msg  # 0 on a GPU box; validation itself passed either way
    out = (C.c_uint32 * 8)()
    assert rta._lib.lib().rt_fast_tree_check(C.byref(d), 0, 2, out) == 0


CORRUPTIONS = {
    "skip link past the end": lambda f: f["nodes"]["skip"].__setitem__(1, len(f["nodes"]) + 5),
    "skip link backwards": lambda f: f["nodes"]["skip"].__setitem__(2, 1),
    "root skip": lambda f: f["nodes"]["skip"].__setitem__(0, len(f["nodes"]) - 1),
    "left child inside right subtree": lambda f: f["nodes"]["left"].__setitem__(0, 1),
    "leaf slot range": lambda f: f["nodes"]["slot_count"].__setitem__(int(np.flatnonzero(f["nodes"]["left"] == 0)[0]), 999),
    "inner node with primitives": lambda f: f["nodes"]["slot_count"].__setitem__(0, 1),
    "slot twice": lambda f: f["slots"].__setitem__(1, f["slots"][0]),
    "slot out of range": lambda f: f["slots"].__setitem__(0, 0x7FFFFFF0),
    "polygon material": lambda f: f["polygons"]["material"].__setitem__(0, 1000),
    "polygon with two points": lambda f: f["polygons"]["n_points"].__setitem__(0, 2),
    "polygon points out of range": lambda f: f["polygons"]["points_offset"].__setitem__(0, 10 ** 6),
    "sphere material": lambda f: f["spheres"]["material"].__setitem__(0, 1000),
    "empty leaf": lambda f: f["nodes"]["slot_count"].__setitem__(int(np.flatnonzero(f["nodes"]["left"] == 0)[-1]), 0),
    "prim_id twice": lambda f: f["polygons"]["prim_id"].__setitem__(0, f["spheres"]["prim_id"][0]),
    "prim_id out of range": lambda f: f["spheres"]["prim_id"].__setitem__(0, 0xFFFFFFFF),
    "prim_id huge": lambda f: f["polygons"]["prim_id"].__setitem__(1, 0xF0000000),
    "gloss": lambda f: f["materials"]["gloss"].__setitem__(0, -0.5),
    "gloss nan": lambda f: f["materials"]["gloss"].__setitem__(0, np.nan),
}


@pytest.mark.parametrize("name", sorted(CORRUPTIONS))
def test_corrupt_descriptions_are_rejected(rta, flat, name):
    f = {k: v.copy() for k, v in flat.items()}
    CORRUPTIONS[name](f)
    d, keep = desc_from(rta, f)
    st, msg = create(rta, d)
    assert st == RT_ERR_INVALID_ARGUMENT, (name, st, msg)
    assert msg.startswith("rt_scene_create: ")
    out = (C.c_uint32 * 8)()
    assert rta._lib.lib().rt_fast_tree_check(C.byref(d), 0, 1, out) == RT_ERR_INVALID_ARGUMENT


def test_header_fields_are_checked(rta, flat):
    d, keep = desc_from(rta, flat)
    d.struct_size = 8
    assert create(rta, d)[0] == RT_ERR_INVALID_ARGUMENT
    d, keep = desc_from(rta, flat, fast_tree=7)
    st, msg = create(rta, d)
    assert st == RT_ERR_INVALID_ARGUMENT and "fast_tree" in msg
    for bits in (0x200, 0x103, 0x80000000):  # unknown flag, unknown kind under a known flag
        d, keep = desc_from(rta, flat, fast_tree=bits)
        st, msg = create(rta, d)
        assert st == RT_ERR_INVALID_ARGUMENT and "fast_tree" in msg, hex(bits)
    for bits in (0x100, 0x101, 0x102):  # RT_FAST_BOUNDS_CIRCLE with every kind of hierarchy
        d, keep = desc_from(rta, flat, fast_tree=bits)
        st, msg = create(rta, d)
        assert st in (0, RT_ERR_NO_DEVICE), (hex(bits), msg)
    d, keep = desc_from(rta, flat)
    d.n_slots += 1
    assert create(rta, d)[0] == RT_ERR_INVALID_ARGUMENT
    assert rta._lib.lib().rt_scene_create(None, 0, C.byref(C.c_void_p())) == RT_ERR_INVALID_ARGUMENT
