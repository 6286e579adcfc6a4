"""ctypes bindings of librt_amd.so (include/rt_amd.h and include/rt_host.h).

The library is the product: there is no Python or CPU implementation behind
these calls. If the shared object is missing the import fails loudly.
"""
import ctypes as C
import os
from pathlib import Path

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "librt_amd.so"
if os.environ.get("RT_AMD_LIBRARY"):  # an instrumented or experimental build of the same sources
    LIB_PATH = Path(os.environ["RT_AMD_LIBRARY"])

MISS = 0xFFFFFFFF
MAX_RAY_DEPTH = 5
TRAVERSE_FAST = 0
TRAVERSE_REFERENCE_ORDER = 1
SHARD_TILES = 0
SHARD_SAMPLES = 1
SLOT_SPHERE = 0x80000000

RT_OK, RT_ERR_INVALID_ARGUMENT, RT_ERR_NO_DEVICE, RT_ERR_DEVICE, RT_ERR_OUT_OF_MEMORY, RT_ERR_UNSUPPORTED = range(6)
STATIC_MAX_PATHS = 65536  # RT_STATIC_MAX_PATHS


class RtError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"rt status {status}: {message}")
        self.status = status


class Material(C.Structure):  # rt_material
    _fields_ = [("reflection_spectrum", C.c_double * 3), ("emissivity", C.c_double),
                ("inside_index_of_refraction", C.c_double), ("gloss", C.c_double),
                ("reserved", C.c_double * 2)]


class SphereRec(C.Structure):  # rt_sphere
    _fields_ = [("center", C.c_double * 3), ("radius", C.c_double), ("material", C.c_uint32),
                ("prim_id", C.c_uint32), ("reserved", C.c_double)]


class PolygonRec(C.Structure):  # rt_polygon
    _fields_ = [("normal", C.c_double * 3), ("bounding_radius", C.c_double), ("p0", C.c_double * 3),
                ("bounding_center", C.c_double * 3), ("basis_1", C.c_double * 3), ("basis_2", C.c_double * 3),
                ("n_points", C.c_uint32), ("points_offset", C.c_uint32), ("material", C.c_uint32),
                ("prim_id", C.c_uint32)]


class NodeRec(C.Structure):  # rt_node
    _fields_ = [("shell", C.c_double * 4), ("skip", C.c_uint32), ("left", C.c_uint32),
                ("slot_begin", C.c_uint32), ("slot_count", C.c_uint32), ("reserved", C.c_uint32 * 4)]


class SceneDesc(C.Structure):  # rt_scene_desc
    _fields_ = [("struct_size", C.c_uint32), ("n_nodes", C.c_uint32), ("nodes", C.POINTER(NodeRec)),
                ("n_slots", C.c_uint32), ("n_polygons", C.c_uint32), ("slots", C.POINTER(C.c_uint32)),
                ("polygons", C.POINTER(PolygonRec)), ("points2d", C.POINTER(C.c_double)),
                ("n_points2d", C.c_uint32), ("n_spheres", C.c_uint32), ("spheres", C.POINTER(SphereRec)),
                ("materials", C.POINTER(Material)), ("n_materials", C.c_uint32), ("fast_tree", C.c_uint32)]


class Camera(C.Structure):  # rt_camera
    _fields_ = [("loc", C.c_double * 3), ("dir", C.c_double * 9), ("focal_length", C.c_double),
                ("sensor_to_lens", C.c_double), ("sensor_diagonal", C.c_double),
                ("aperture_radius", C.c_double)]


class Stats(C.Structure):  # rt_stats
    _fields_ = [("rays", C.c_uint64), ("paths", C.c_uint64), ("traverse_ms", C.c_double),
                ("total_ms", C.c_double), ("traverse_launches", C.c_uint64), ("reserved", C.c_uint64 * 3)]

    def as_dict(self):
        return dict(rays=int(self.rays), paths=int(self.paths), traverse_ms=float(self.traverse_ms),
                    total_ms=float(self.total_ms), traverse_launches=int(self.traverse_launches))


class SceneInfo(C.Structure):  # rt_scene_info
    _fields_ = [(n, C.c_uint32) for n in ("n_nodes", "n_leaves", "depth", "n_prims", "n_polygons", "n_spheres",
                                          "n_materials", "max_leaf")] + \
               [("device_bytes", C.c_uint64), ("device", C.c_int32), ("fast_tree", C.c_uint32),
                ("fast_depth", C.c_uint32), ("fast_nodes", C.c_uint32), ("reserved", C.c_uint32 * 2)]


class BuildInfo(C.Structure):  # rt_build_info
    _fields_ = [(n, C.c_uint32) for n in ("n_nodes", "n_leaves", "depth", "n_prims", "max_leaf", "guard_leaves")] + \
               [("reserved", C.c_uint32 * 2)]


assert C.sizeof(Material) == 64 and C.sizeof(SphereRec) == 48 and C.sizeof(PolygonRec) == 144
assert C.sizeof(NodeRec) == 64

_vp, _u32, _u64, _dbl, _int = C.c_void_p, C.c_uint32, C.c_uint64, C.c_double, C.c_int
_dp, _u32p, _u8p = C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)

# name -> argtypes; every function returns rt_status unless listed in _OTHER
SIGNATURES = {
    # rt_amd.h
    "rt_device_count": [C.POINTER(_int)],
    "rt_scene_create": [C.POINTER(SceneDesc), _int, C.POINTER(_vp)],
    "rt_scene_destroy": [_vp],
    "rt_scene_get_info": [_vp, C.POINTER(SceneInfo)],
    "rt_fast_tree_check": [C.POINTER(SceneDesc), _int, C.c_uint32, C.POINTER(C.c_uint32)],
    "rt_intersect_batch": [_vp, _dp, _u64, _int, _u32p, _dp, _dp, _dp, C.POINTER(Stats)],
    "rt_intersect_batch_device": [_vp, _vp, _u64, _int, _vp, _vp, _vp, _vp, _vp],
    "rt_intersect_brute_force": [_vp, _dp, _u64, _u32p, _dp],
    "rt_intersect_primitives": [_vp, _dp, _u32p, _u64, _u8p, _dp, _dp],
    "rt_trace_paths": [_vp, _dp, _u32p, _u64, _u64, _u32, _int, _dp, C.POINTER(Stats)],
    "rt_trace_paths_static": [_vp, _vp, _vp, _u64, _u64, _u32, _int, _vp, _vp],
    "rt_ray_bundle": [C.POINTER(Camera), _u32, _u32, _u64, _int, _dp],
    "rt_tone_map": [_dp, _u64, _int, _u8p],
    "rt_tone_map_device": [_vp, _u64, _vp, _vp],
    "rt_film_create": [_vp, _u32, _u32, _u32, _u32, _u32, C.POINTER(_vp)],
    "rt_film_destroy": [_vp],
    "rt_film_set_camera": [_vp, C.POINTER(Camera), _u64],
    "rt_film_set_rays": [_vp, _dp],
    "rt_film_clear": [_vp],
    "rt_film_render": [_vp, _u32, _u32, _u64, _u32, _int, _vp],
    "rt_film_resolve_device": [_vp, _vp, _vp, _vp],
    "rt_film_resolve": [_vp, _dp, _u8p],
    "rt_film_get_stats": [_vp, C.POINTER(Stats)],
    "rt_render": [_vp, C.POINTER(Camera), _u32, _u32, _u32, _u64, _int, _dp, _u8p, C.POINTER(Stats)],
    "rt_render_multi": [C.POINTER(SceneDesc), C.POINTER(_int), _u32, C.POINTER(Camera), _u32, _u32, _u32, _u64, _int,
                        _int, _dp, _u8p, C.POINTER(Stats)],
    # rt_host.h
    "rtb_create": [C.POINTER(_vp)],
    "rtb_destroy": [_vp],
    "rtb_add_sphere": [_vp, _dp, _dbl, C.POINTER(Material), _u32p],
    "rtb_add_polygon": [_vp, _dp, _u32, C.POINTER(Material), _dp, _u32p],
    "rtb_add_parallelpiped": [_vp, _dp, _dp, _dp, _dp, C.POINTER(Material), _u32p],
    "rtb_add_mesh": [_vp, _dp, _u32, _u32p, _u32, _u32, _dbl, C.POINTER(Material), _u32p],
    "rtb_add_obj": [_vp, C.c_char_p, C.c_size_t, _dbl, C.POINTER(Material), _u32p, _u32p],
    "rtb_add_reference_room": [_vp],
    "rtb_build": [_vp],
    "rtb_get_info": [_vp, C.POINTER(BuildInfo)],
    "rtb_get_desc": [_vp, C.POINTER(SceneDesc)],
    "rtb_package": [_vp, _u32, _dp],
    "rtb_write_png": [C.c_char_p, _u8p, _u32, _u32],
}
_OTHER = {"rt_last_error": (C.c_char_p, []), "rt_abi_version": (C.c_uint32, [])}
EXPORTS = sorted(list(SIGNATURES) + list(_OTHER))

_lib = None


def lib():
    """Loads librt_amd.so; raises if it has not been built (no fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process. PyTorch wheels bundle their own libamdhip64 / libhsa-runtime64; if this
    # library is loaded first it binds the system's copy, a later `import torch` brings the second copy,
    # and that one finds no usable GPU ("No HIP GPUs are available": measured on MI355X, ROCm 7.2 +
    # torch 2.10). With torch imported first, both use the copy torch ships. RT_AMD_NO_TORCH=1 skips this
    # for processes that never use torch.
    if not os.environ.get("RT_AMD_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not LIB_PATH.exists():
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C {HERE / 'csrc'}` "
                          "(or __graft_entry__.build()); there is no CPU implementation to fall back to")
    L = C.CDLL(str(LIB_PATH))
    for name, args in SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype = C.c_int
        fn.argtypes = args
    for name, (res, args) in _OTHER.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(status):
    if status != RT_OK:
        raise RtError(status, lib().rt_last_error().decode("utf-8", "replace"))
