"""ray-tracer_amd — MI355X-native ray intersection and path tracing.

The package directory name contains a hyphen, so import it through the alias
module at the repository root: ``import ray_tracer_amd as rta``.
"""
from . import _lib, scenes, sharding
from ._lib import (MAX_RAY_DEPTH, MISS, TRAVERSE_FAST, TRAVERSE_REFERENCE_ORDER, Camera, RtError, Stats)
from .batches import make_ray_batch
from .scene import (BVH, Brdf, Film, Hit, Mesh, ObjText, Polygon, Sphere, device_count, init_environment,
                    load_rtmesh, make_camera, polys_from_obj, ray_bundle, ray_bundle_cam, reference_camera, render, render_multi,
                    spectral_radiance_to_color, trace_ray, write_to_png, HIPPO_MESH)

__all__ = [n for n in dir() if not n.startswith("_")]
