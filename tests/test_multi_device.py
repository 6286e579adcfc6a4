"""rt_render_multi: the thread fan-out and merge of main.rs:34-104 over a list of devices, behind the C ABI.

A GPU box for the tests has one device, so the list repeats device 0: the shards then share the GPU, but every
step is the multi-device code path (one host thread, stream, film and frame per shard, peer copies into the
first shard's device, one combine per frame).
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


@pytest.mark.parametrize("w,h,spp", [(96, 64, 5), (75, 45, 2)])
def test_tiles_over_a_device_list_equal_one_device(rta, hippo_bvh, w, h, spp):
    cam = rta.reference_camera()
    full, full_rgba, st1 = rta.render(hippo_bvh, cam, w, h, spp, seed=21)
    for devices in ((0,), (0, 0, 0), (0, 0, 0, 0, 0, 0, 0, 0)):
        got, rgba, st = rta.render_multi(hippo_bvh, cam, w, h, spp, seed=21, devices=devices, shard="tiles")
        assert np.array_equal(bits(got), bits(full)), devices
        assert np.array_equal(rgba, full_rgba), devices
        assert st["paths"] == st1["paths"] == w * h * 5 * spp and st["rays"] == st1["rays"]


def test_sample_ranges_over_a_device_list(rta, hippo_bvh):
    w, h, spp = 96, 64, 7  # 7 samples over 3 shards: ranges of 3, 2 and 2
    cam = rta.reference_camera()
    full, full_rgba, st1 = rta.render(hippo_bvh, cam, w, h, spp, seed=9)
    got, rgba, st = rta.render_multi(hippo_bvh, cam, w, h, spp, seed=9, devices=(0, 0, 0), shard="samples")
    # the same paths, added in another order: equal up to the rounding of the sums
    assert np.allclose(got, full, rtol=1e-12, atol=1e-15)
    assert (rgba == full_rgba).mean() > 0.999 and (rgba[..., 3] == 255).all()
    assert st["paths"] == st1["paths"] and st["rays"] == st1["rays"]
    one, one_rgba, _ = rta.render_multi(hippo_bvh, cam, w, h, spp, seed=9, devices=(0,), shard="samples")
    assert np.array_equal(bits(one), bits(full)) and np.array_equal(one_rgba, full_rgba)
    # more shards than samples: the empty ranges contribute nothing
    few, _, _ = rta.render_multi(hippo_bvh, cam, w, h, 2, seed=9, devices=(0, 0, 0), shard="samples")
    two, _, _ = rta.render(hippo_bvh, cam, w, h, 2, seed=9)
    assert np.allclose(few, two, rtol=1e-12, atol=1e-15)


def test_device_list_scene_with_spheres_and_reference_order(rta, mixed_bvh):
    cam = rta.reference_camera()
    w, h, spp = 64, 40, 3
    for mode in (rta.TRAVERSE_FAST, rta.TRAVERSE_REFERENCE_ORDER):
        full, full_rgba, _ = rta.render(mixed_bvh, cam, w, h, spp, seed=4, mode=mode)
        got, rgba, _ = rta.render_multi(mixed_bvh, cam, w, h, spp, seed=4, devices=(0, 0), shard="tiles", mode=mode)
        assert np.array_equal(bits(got), bits(full)) and np.array_equal(rgba, full_rgba)


def test_device_list_errors(rta, hippo_bvh):
    lib = rta._lib.lib()
    cam = rta.reference_camera()
    d = hippo_bvh.desc()
    out = np.zeros((8, 8, 3))
    rgba = np.zeros((8, 8, 4), dtype=np.uint8)

    def call(devices, shard=0, width=8, n=None):
        devs = (C.c_int * max(1, len(devices)))(*devices)
        return lib.rt_render_multi(C.byref(d), devs, len(devices) if n is None else n, C.byref(cam), width, 8, 1,
                                   C.c_uint64(1), 0, shard, out.ctypes.data_as(C.POINTER(C.c_double)),
                                   rgba.ctypes.data_as(C.POINTER(C.c_uint8)), None)

    assert call([0]) == 0
    assert call([]) == rta._lib.RT_ERR_INVALID_ARGUMENT
    assert call([0, 99]) == rta._lib.RT_ERR_INVALID_ARGUMENT and b"device 99" in lib.rt_last_error()
    assert call([0, -1]) == rta._lib.RT_ERR_INVALID_ARGUMENT
    assert call([0], shard=7) == rta._lib.RT_ERR_INVALID_ARGUMENT
    assert call([0], width=0) == rta._lib.RT_ERR_INVALID_ARGUMENT
    assert call([0], n=65) == rta._lib.RT_ERR_INVALID_ARGUMENT
