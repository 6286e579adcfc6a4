"""Host-side mirror of the reference crate's scene, BVH and render entry points.

Same names and argument meaning as rust_tracing/src/{brdf,primitives,bvh,tracer,
camera,main}.rs; every query is answered by the GPU library (librt_amd.so).
"""
import ctypes as C
import struct
from pathlib import Path

import numpy as np

from . import _lib as L
from ._lib import (MAX_RAY_DEPTH, MISS, TRAVERSE_FAST, TRAVERSE_REFERENCE_ORDER, Camera, Material, RtError, Stats,
                   check)

_dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
_u32p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))
_u8p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


class Brdf:
    """brdf.rs:4-10"""

    def __init__(self, reflection_spectrum, emissivity=0.0, inside_index_of_refraction=1.0, gloss=0.0):
        self.reflection_spectrum = tuple(float(x) for x in reflection_spectrum)
        self.emissivity = float(emissivity)
        self.inside_index_of_refraction = float(inside_index_of_refraction)
        self.gloss = float(gloss)  # 0 = ideal diffuse (all brdf.rs implements); see variable_glossy

    @classmethod
    def variable_glossy(cls, n, reflection_spectrum, emissivity=0.0):
        """geometry.js:172-184: n = 1 perfectly diffuse ... n = 0 perfectly reflective."""
        if not 0.0 <= n <= 1.0:
            raise ValueError("n must lie in [0, 1]")
        return cls(reflection_spectrum, emissivity, 1.0, gloss=1.0 - n)

    def _rec(self):
        m = Material()
        m.reflection_spectrum[:] = self.reflection_spectrum
        m.emissivity = self.emissivity
        m.inside_index_of_refraction = self.inside_index_of_refraction
        m.gloss = self.gloss
        return m


class Sphere:
    """primitives.rs:35-39"""

    def __init__(self, center, radius, brdf):
        self.center = tuple(float(x) for x in center)
        self.radius = float(radius)
        self.brdf = brdf

    def _add(self, b):
        check(L.lib().rtb_add_sphere(b, (C.c_double * 3)(*self.center), self.radius, C.byref(self.brdf._rec()), None))


class Polygon:
    """primitives.rs:99-114; construct with new_cached (primitives.rs:144)."""

    def __init__(self, points, brdf, direc_vec):
        self.points = _f64(points, (-1, 3))
        self.brdf = brdf
        self.direc_vec = tuple(float(x) for x in direc_vec)

    @staticmethod
    def new_cached(points, brdf, direc_vec):
        return Polygon(points, brdf, direc_vec)

    @staticmethod
    def parallelpiped(origin, gen1, gen2, gen3, brdf):
        """primitives.rs:117-141: the six faces of a box."""
        return _Parallelpiped(origin, gen1, gen2, gen3, brdf)

    def _add(self, b):
        check(L.lib().rtb_add_polygon(b, _dp(self.points), self.points.shape[0], C.byref(self.brdf._rec()),
                                      (C.c_double * 3)(*self.direc_vec), None))


class _Parallelpiped:
    def __init__(self, origin, gen1, gen2, gen3, brdf):
        self.args = [tuple(float(x) for x in v) for v in (origin, gen1, gen2, gen3)]
        self.brdf = brdf

    def _add(self, b):
        o, g1, g2, g3 = [(C.c_double * 3)(*v) for v in self.args]
        check(L.lib().rtb_add_parallelpiped(b, o, g1, g2, g3, C.byref(self.brdf._rec()), None))


class Mesh:
    """A triangle (or fixed-length n-gon) mesh: one Polygon per face with
    direc_vec = cross(p1-p0, p2-p1), as polys_from_obj builds them (main.rs:216-236)."""

    def __init__(self, vertices, faces, brdf, z_shift=0.0):
        self.vertices = _f64(vertices, (-1, 3))
        self.faces = np.ascontiguousarray(faces, dtype=np.uint32)
        self.brdf = brdf
        self.z_shift = float(z_shift)

    def _add(self, b):
        check(L.lib().rtb_add_mesh(b, _dp(self.vertices), self.vertices.shape[0], _u32p(self.faces),
                                   self.faces.shape[0], self.faces.shape[1], self.z_shift,
                                   C.byref(self.brdf._rec()), None))


class ObjText:
    """polys_from_obj (main.rs:172-240) on OBJ text."""

    def __init__(self, text, brdf=None, z_shift=-0.717):
        self.text = text.encode() if isinstance(text, str) else bytes(text)
        self.brdf = brdf or Brdf((0.65, 0.65, 0.65))
        self.z_shift = float(z_shift)

    def _add(self, b):
        check(L.lib().rtb_add_obj(b, self.text, len(self.text), self.z_shift, C.byref(self.brdf._rec()), None, None))


def polys_from_obj(file_name, z_shift=-0.717):
    """main.rs:172: reads the file; raises like the reference's expect()."""
    try:
        data = Path(file_name).read_bytes()
    except OSError as e:
        raise RuntimeError(f"something went wrong with reading the file: {e}") from e
    return [ObjText(data, z_shift=z_shift)]


class _ReferenceRoom:
    def _add(self, b):
        check(L.lib().rtb_add_reference_room(b))


def load_rtmesh(path):
    """assets/hippo.rtmesh -> (vertices f64 [nv,3], faces u32 [nf,3])."""
    raw = Path(path).read_bytes()
    if raw[:8] != b"RTMESH1\0":
        raise ValueError("bad mesh magic")
    nv, nf = struct.unpack_from("<QQ", raw, 8)
    v = np.frombuffer(raw, dtype="<f8", count=nv * 3, offset=24).reshape(nv, 3).copy()
    f = np.frombuffer(raw, dtype="<u4", count=nf * 3, offset=24 + nv * 24).reshape(nf, 3).copy()
    return v, f


HIPPO_MESH = Path(__file__).resolve().parent / "assets" / "hippo.rtmesh"


def init_environment(mesh_path=None, device=0, circle_bounds=False):
    """main.rs:117-170: floor, emissive wall, two walls and the hippo mesh."""
    v, f = load_rtmesh(mesh_path or HIPPO_MESH)
    objects = [_ReferenceRoom(), Mesh(v, f, Brdf((0.65, 0.65, 0.65)), z_shift=-0.717)]
    return BVH.from_list(objects, device=device, circle_bounds=circle_bounds)


class Hit:
    __slots__ = ("pos", "normal", "brdf", "prim_id", "distance")

    def __init__(self, pos, normal, brdf, prim_id, distance):
        self.pos, self.normal, self.brdf, self.prim_id, self.distance = pos, normal, brdf, prim_id, distance


FAST_TREES = {"sah": 0, "reference": 1, "lbvh": 2}
FAST_BOUNDS_CIRCLE = 0x100


class BVH:
    """bvh.rs:10-18. Build with from_list (bvh.rs:21); the flattened tree is
    uploaded to the GPU lazily, on the first query."""

    def __init__(self):
        self._b = C.c_void_p()
        self._scene = C.c_void_p()
        self._materials = None
        self.device = 0

    @staticmethod
    def from_list(objects, device=0, fast_tree="sah", circle_bounds=False):
        """BVH::from_list (bvh.rs:21). fast_tree picks the hierarchy RT_TRAVERSE_FAST walks:
        "sah" (host build, default), "reference" (the reference topology with boxes) or "lbvh"
        (built on the GPU); the hits are the same. circle_bounds (RT_FAST_BOUNDS_CIRCLE): bound every
        polygon by its bounding circle, so that the fast walk also returns the hits the reference
        reports outside a polygon's outline when the plane point is level with a vertex to the bit."""
        self = BVH()
        self.device = device
        self.fast_tree = FAST_TREES[fast_tree] | (FAST_BOUNDS_CIRCLE if circle_bounds else 0)
        check(L.lib().rtb_create(C.byref(self._b)))
        for o in objects:
            o._add(self._b)
        check(L.lib().rtb_build(self._b))
        return self

    # ---- host-side structure ---------------------------------------------
    def info(self):
        i = L.BuildInfo()
        check(L.lib().rtb_get_info(self._b, C.byref(i)))
        return {n: int(getattr(i, n)) for n, _ in i._fields_ if n != "reserved"}

    def desc(self):
        d = L.SceneDesc()
        check(L.lib().rtb_get_desc(self._b, C.byref(d)))
        return d

    def flat(self):
        """The flattened arrays as numpy views/copies (for inspection and tests)."""
        d = self.desc()
        nodes = np.ctypeslib.as_array(C.cast(d.nodes, C.POINTER(C.c_uint8)), shape=(d.n_nodes * 64,)).copy()
        node_dt = np.dtype([("shell", "<f8", 4), ("skip", "<u4"), ("left", "<u4"), ("slot_begin", "<u4"),
                            ("slot_count", "<u4"), ("reserved", "<u4", 4)])
        poly_dt = np.dtype([("normal", "<f8", 3), ("bounding_radius", "<f8"), ("p0", "<f8", 3),
                            ("bounding_center", "<f8", 3), ("basis_1", "<f8", 3), ("basis_2", "<f8", 3),
                            ("n_points", "<u4"), ("points_offset", "<u4"), ("material", "<u4"), ("prim_id", "<u4")])
        sph_dt = np.dtype([("center", "<f8", 3), ("radius", "<f8"), ("material", "<u4"), ("prim_id", "<u4"),
                           ("reserved", "<f8")])
        mat_dt = np.dtype([("reflection_spectrum", "<f8", 3), ("emissivity", "<f8"),
                           ("inside_index_of_refraction", "<f8"), ("gloss", "<f8"), ("reserved", "<f8", 2)])

        def view(ptr, n, dt):
            if n == 0:
                return np.zeros(0, dtype=dt)
            raw = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n * dt.itemsize,)).copy()
            return raw.view(dt)

        return dict(nodes=nodes.view(node_dt), slots=view(d.slots, d.n_slots, np.dtype("<u4")),
                    polygons=view(d.polygons, d.n_polygons, poly_dt),
                    points2d=view(d.points2d, d.n_points2d * 2, np.dtype("<f8")).reshape(-1, 2),
                    spheres=view(d.spheres, d.n_spheres, sph_dt), materials=view(d.materials, d.n_materials, mat_dt))

    def package(self, prim_id):
        out = (C.c_double * 4)()
        check(L.lib().rtb_package(self._b, prim_id, out))
        return tuple(out)

    # ---- device ------------------------------------------------------------
    def scene(self):
        if not self._scene:
            d = self.desc()
            d.fast_tree = getattr(self, "fast_tree", 0)
            check(L.lib().rt_scene_create(C.byref(d), self.device, C.byref(self._scene)))
        return self._scene

    def scene_info(self):
        i = L.SceneInfo()
        check(L.lib().rt_scene_get_info(self.scene(), C.byref(i)))
        return {n: int(getattr(i, n)) for n, _ in i._fields_ if n != "reserved"}

    def materials(self):
        if self._materials is None:
            self._materials = self.flat()["materials"]
        return self._materials

    def intersect_batch(self, rays, mode=TRAVERSE_FAST, want_pos=False, stats=None):
        """BVH::intersect for n rays: returns (prim_id, distance[, pos, normal])."""
        r = _f64(rays, (-1, 6))
        n = r.shape[0]
        ids = np.empty(n, dtype=np.uint32)
        dist = np.empty(n, dtype=np.float64)
        pos = np.empty((n, 3), dtype=np.float64) if want_pos else None
        nrm = np.empty((n, 3), dtype=np.float64) if want_pos else None
        check(L.lib().rt_intersect_batch(self.scene(), _dp(r), n, mode, _u32p(ids), _dp(dist),
                                         _dp(pos) if want_pos else None, _dp(nrm) if want_pos else None,
                                         C.byref(stats) if stats is not None else None))
        return (ids, dist, pos, nrm) if want_pos else (ids, dist)

    def intersect(self, ray_origin, ray_dir, mode=TRAVERSE_FAST):
        """bvh.rs:97: Option<(pos, normal, &Brdf)> -> Hit or None."""
        ids, dist, pos, nrm = self.intersect_batch([list(ray_origin) + list(ray_dir)], mode, want_pos=True)
        if ids[0] == MISS:
            return None
        return Hit(pos[0], nrm[0], None, int(ids[0]), float(dist[0]))

    def brute_force_batch(self, rays):
        """Closest hits by testing every primitive (diagnostic; n x primitives tests)."""
        r = _f64(rays, (-1, 6))
        n = r.shape[0]
        ids = np.empty(n, dtype=np.uint32)
        dist = np.empty(n, dtype=np.float64)
        check(L.lib().rt_intersect_brute_force(self.scene(), _dp(r), n, _u32p(ids), _dp(dist)))
        return ids, dist

    def intersect_primitives(self, rays, prim_ids):
        """Primitive::intersect / normal_at for (ray i, primitive prim_ids[i]) pairs."""
        r = _f64(rays, (-1, 6))
        ids = np.ascontiguousarray(prim_ids, dtype=np.uint32)
        n = r.shape[0]
        hit = np.empty(n, dtype=np.uint8)
        pos = np.empty((n, 3), dtype=np.float64)
        nrm = np.empty((n, 3), dtype=np.float64)
        check(L.lib().rt_intersect_primitives(self.scene(), _dp(r), _u32p(ids), n, _u8p(hit), _dp(pos), _dp(nrm)))
        return hit.astype(bool), pos, nrm

    def trace_paths(self, rays, keys, seed=0, max_depth=MAX_RAY_DEPTH, mode=TRAVERSE_FAST, stats=None):
        """tracer::trace_ray for n paths; keys[i] = (pixel, sample, sub-ray)."""
        r = _f64(rays, (-1, 6))
        k = np.ascontiguousarray(keys, dtype=np.uint32).reshape(-1, 3)
        n = r.shape[0]
        out = np.empty((n, 3), dtype=np.float64)
        check(L.lib().rt_trace_paths(self.scene(), _dp(r), _u32p(k), n, C.c_uint64(seed), max_depth, mode, _dp(out),
                                     C.byref(stats) if stats is not None else None))
        return out

    def close(self):
        if self._scene:
            L.lib().rt_scene_destroy(self._scene)
            self._scene = C.c_void_p()
        if self._b:
            L.lib().rtb_destroy(self._b)
            self._b = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def trace_ray(ray_origin, ray_dir, bvh, seed=0, pixel=0, sample=0, sub_ray=0):
    """tracer.rs:8"""
    return bvh.trace_paths([list(ray_origin) + list(ray_dir)], [[pixel, sample, sub_ray]], seed=seed)[0]


def reference_camera():
    """The camera literals of main.rs:22-30."""
    cam = Camera()
    cam.loc[:] = [0.0, -1.7, -0.65]
    cam.dir[:] = [0.0, 1.0, 0.0, 0.0, 0.0, 1.0, 1.0, 0.0, 0.0]
    cam.focal_length = 2.0
    cam.sensor_to_lens = 0.039
    cam.sensor_diagonal = 0.05
    cam.aperture_radius = 0.000001
    return cam


def make_camera(camera_loc, camera_dir, camera_focal_length, camera_sensor_to_lens, sensor_diagonal, aperture_radius):
    cam = Camera()
    cam.loc[:] = [float(x) for x in camera_loc]
    cam.dir[:] = [float(x) for row in camera_dir for x in row]
    cam.focal_length = camera_focal_length
    cam.sensor_to_lens = camera_sensor_to_lens
    cam.sensor_diagonal = sensor_diagonal
    cam.aperture_radius = aperture_radius
    return cam


def ray_bundle(camera_loc, camera_dir, camera_focal_length, camera_sensor_to_lens, sensor_diagonal, aperture_radius,
               image_resolution, seed=0, device=0):
    """camera.rs:16; image_resolution may be an int (square, as the reference) or (width, height)."""
    cam = make_camera(camera_loc, camera_dir, camera_focal_length, camera_sensor_to_lens, sensor_diagonal,
                      aperture_radius)
    if isinstance(image_resolution, int):
        width = height = image_resolution
    else:
        width, height = image_resolution
    return ray_bundle_cam(cam, width, height, seed, device)


def ray_bundle_cam(cam, width, height, seed=0, device=0):
    out = np.empty(width * height * 30, dtype=np.float64)
    check(L.lib().rt_ray_bundle(C.byref(cam), width, height, C.c_uint64(seed), device, _dp(out)))
    return out


def spectral_radiance_to_color(spectral_radiance, device=0):
    """camera.rs:80; accepts one triple or an (n,3) array."""
    r = _f64(spectral_radiance, (-1, 3))
    out = np.empty((r.shape[0], 4), dtype=np.uint8)
    check(L.lib().rt_tone_map(_dp(r), r.shape[0], device, _u8p(out)))
    return out[0] if np.ndim(spectral_radiance) == 1 else out


class Film:
    """The per-pixel loop and thread block of main.rs:39-104 for one image shard."""

    def __init__(self, bvh, width, height, shard_index=0, shard_count=1, tile_size=32):
        self.bvh, self.width, self.height = bvh, width, height
        self._f = C.c_void_p()
        check(L.lib().rt_film_create(bvh.scene(), width, height, shard_index, shard_count, tile_size,
                                     C.byref(self._f)))

    def set_camera(self, cam, seed=0):
        check(L.lib().rt_film_set_camera(self._f, C.byref(cam), C.c_uint64(seed)))

    def set_rays(self, rays):
        r = _f64(rays)
        assert r.size == self.width * self.height * 30
        check(L.lib().rt_film_set_rays(self._f, _dp(r)))

    def clear(self):
        check(L.lib().rt_film_clear(self._f))

    def render(self, first_sample, n_samples, seed=0, max_depth=MAX_RAY_DEPTH, mode=TRAVERSE_FAST, stream=None):
        check(L.lib().rt_film_render(self._f, first_sample, n_samples, C.c_uint64(seed), max_depth, mode,
                                     C.c_void_p(stream or 0)))

    def resolve_device(self, d_radiance_ptr, d_rgba_ptr, stream=None):
        check(L.lib().rt_film_resolve_device(self._f, C.c_void_p(d_radiance_ptr or 0), C.c_void_p(d_rgba_ptr or 0),
                                             C.c_void_p(stream or 0)))

    def resolve(self):
        rad = np.empty((self.height, self.width, 3), dtype=np.float64)
        rgba = np.empty((self.height, self.width, 4), dtype=np.uint8)
        check(L.lib().rt_film_resolve(self._f, _dp(rad), _u8p(rgba)))
        return rad, rgba

    def stats(self):
        s = Stats()
        check(L.lib().rt_film_get_stats(self._f, C.byref(s)))
        return s.as_dict()

    def close(self):
        if self._f:
            L.lib().rt_film_destroy(self._f)
            self._f = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def render(bvh, cam, width, height, spp, seed=0, mode=TRAVERSE_FAST):
    """What main() does between main.rs:39 and :104: returns (mean radiance, rgba, stats)."""
    rad = np.empty((height, width, 3), dtype=np.float64)
    rgba = np.empty((height, width, 4), dtype=np.uint8)
    s = Stats()
    check(L.lib().rt_render(bvh.scene(), C.byref(cam), width, height, spp, C.c_uint64(seed), mode, _dp(rad),
                            _u8p(rgba), C.byref(s)))
    return rad, rgba, s.as_dict()


def render_multi(bvh, cam, width, height, spp, seed=0, devices=(0,), shard="tiles", mode=TRAVERSE_FAST):
    """main.rs:34-104 over a list of devices (rt_render_multi): one host thread per entry of `devices`, the
    scene replicated per distinct device, one combine per frame on devices[0]. shard = "tiles" (byte-identical
    to render()) or "samples" (equal up to summation order). Returns (mean radiance, rgba, stats)."""
    rad = np.empty((height, width, 3), dtype=np.float64)
    rgba = np.empty((height, width, 4), dtype=np.uint8)
    s = Stats()
    d = bvh.desc()
    d.fast_tree = getattr(bvh, "fast_tree", 0)
    devs = (C.c_int * len(devices))(*[int(x) for x in devices])
    check(L.lib().rt_render_multi(C.byref(d), devs, len(devices), C.byref(cam), width, height, spp, C.c_uint64(seed), mode,
                                  {"tiles": L.SHARD_TILES, "samples": L.SHARD_SAMPLES}[shard], _dp(rad), _u8p(rgba),
                                  C.byref(s)))
    return rad, rgba, s.as_dict()


def write_to_png(file_name, rgba):
    """main.rs:242-244: an (height, width, 4) uint8 frame as a PNG file."""
    a = np.ascontiguousarray(rgba, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 4:
        raise ValueError("write_to_png expects an (height, width, 4) uint8 array")
    check(L.lib().rtb_write_png(str(file_name).encode(), _u8p(a), a.shape[1], a.shape[0]))


def device_count():
    n = C.c_int(0)
    st = L.lib().rt_device_count(C.byref(n))
    return n.value if st == L.RT_OK else 0
