"""rt_trace_paths_static: the entry point that allocates nothing.

Its path state lives in a static device array instead of a freshly allocated pool, so it is
checked against rt_trace_paths (the allocating entry point it is a variant of), bit for bit, on
two calls in a row, the second with a smaller input: the array keeps the first call's contents
and whatever must start at zero has to be cleared by the call itself.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


def _static(rta, bvh, rays_d, keys_d, n, seed, depth, mode, out_d, stream=None):
    return rta._lib.lib().rt_trace_paths_static(bvh.scene(), rays_d.data_ptr(), keys_d.data_ptr(), n, C.c_uint64(seed),
                                                depth, mode, out_d.data_ptr(), C.c_void_p(stream or 0))


def test_static_scratch_entry_point_matches_the_allocating_one(rta, hippo_bvh):
    import torch
    dev = torch.device("cuda:0")
    g = np.load(GOLDEN / "hippo_paths.npz")
    seed = int(g["seed"])
    rays_big = np.ascontiguousarray(np.vstack([g["bundle"].reshape(-1, 6)] * 2))
    keys_big = np.ascontiguousarray(g["keys"].astype(np.uint32))
    assert len(rays_big) == len(keys_big) and len(rays_big) <= rta._lib.STATIC_MAX_PATHS
    hippo_bvh.scene()
    torch.cuda.synchronize()
    for mode in (rta.TRAVERSE_FAST, rta.TRAVERSE_REFERENCE_ORDER):
        # first the large call, then a smaller one whose paths are different rays with different keys
        for n, first in ((len(rays_big), 0), (1000, 777), (1, 5)):
            rays = np.ascontiguousarray(rays_big[first:first + n])
            keys = np.ascontiguousarray(keys_big[first:first + n])
            want = hippo_bvh.trace_paths(rays, keys, seed=seed, mode=mode)
            rays_d = torch.from_numpy(rays).to(dev)
            keys_d = torch.from_numpy(keys.view(np.int32)).to(dev)
            out_d = torch.full((n, 3), float("nan"), dtype=torch.float64, device=dev)
            torch.cuda.synchronize()
            free_before = torch.cuda.mem_get_info()[0]
            rta._lib.check(_static(rta, hippo_bvh, rays_d, keys_d, n, seed, 5, mode, out_d))
            torch.cuda.synchronize()
            assert torch.cuda.mem_get_info()[0] >= free_before, "the call took device memory"
            got = out_d.cpu().numpy()
            assert np.array_equal(bits(got), bits(want)), (mode, n, int((bits(got) != bits(want)).any(axis=1).sum()))
            assert got.sum() > 0.0 or n == 1


def test_static_scratch_on_a_stream_and_with_another_depth(rta, sphere_bvh):
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    n = 4096
    o = np.tile([0.0, -1.7, -0.65], (n, 1)) + 0.01 * rng.standard_normal((n, 3))
    d = np.array([0.0, 1.0, -0.1]) + 0.3 * rng.standard_normal((n, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    rays = np.ascontiguousarray(np.hstack([o, d]))
    keys = np.ascontiguousarray(np.stack([np.arange(n), np.arange(n) % 7, np.arange(n) % 5], axis=1).astype(np.uint32))
    rays_d, keys_d = torch.from_numpy(rays).to(dev), torch.from_numpy(keys.view(np.int32)).to(dev)
    stream = torch.cuda.Stream()
    for depth in (5, 9, 2):
        want = sphere_bvh.trace_paths(rays, keys, seed=3, max_depth=depth)
        out_d = torch.zeros((n, 3), dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            rta._lib.check(_static(rta, sphere_bvh, rays_d, keys_d, n, 3, depth, rta.TRAVERSE_FAST, out_d,
                                   stream.cuda_stream))
        stream.synchronize()
        assert np.array_equal(bits(out_d.cpu().numpy()), bits(want)), depth


def test_static_scratch_under_graph_capture(rta, hippo_bvh):
    """What the entry point exists for: recorded into a HIP graph and replayed. The fast walk's work counter
    of a captured launch is baked into the graph, so it comes from a range that live launches never use: the
    replays (one of them overlapping live launches of the same scene on another stream) return the bits of
    rt_trace_paths, and so do the live launches."""
    import torch
    dev = torch.device("cuda:0")
    g = np.load(GOLDEN / "hippo_paths.npz")
    seed = int(g["seed"])
    rays = np.ascontiguousarray(g["bundle"].reshape(-1, 6))
    keys = np.ascontiguousarray(g["keys"].astype(np.uint32)[:len(rays)])
    n = len(rays)
    want = hippo_bvh.trace_paths(rays, keys, seed=seed)
    rays_d, keys_d = torch.from_numpy(rays).to(dev), torch.from_numpy(keys.view(np.int32)).to(dev)
    out_d = torch.zeros((n, 3), dtype=torch.float64, device=dev)
    # once outside capture: occupancy queries and module loading happen here
    rta._lib.check(_static(rta, hippo_bvh, rays_d, keys_d, n, seed, 5, rta.TRAVERSE_FAST, out_d))
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(graph, stream=side):
        rta._lib.check(_static(rta, hippo_bvh, rays_d, keys_d, n, seed, 5, rta.TRAVERSE_FAST, out_d,
                               torch.cuda.current_stream().cuda_stream))
    # live launches on another stream while the graph replays: a 1 Mi-ray batch, several times
    batch = torch.from_numpy(rta.make_ray_batch(hippo_bvh, 1 << 20, seed=5)).to(dev)
    ids_ref, dist_ref = hippo_bvh.intersect_batch(batch.cpu().numpy()[: 1 << 16])
    ids = torch.empty(1 << 20, dtype=torch.int32, device=dev)
    dist = torch.empty(1 << 20, dtype=torch.float64, device=dev)
    live = torch.cuda.Stream()
    lib = rta._lib.lib()
    for replay in range(3):
        out_d.fill_(float("nan"))
        torch.cuda.synchronize()
        with torch.cuda.stream(live):
            for _ in range(4):
                rta._lib.check(lib.rt_intersect_batch_device(hippo_bvh.scene(), batch.data_ptr(), 1 << 20, rta.TRAVERSE_FAST,
                                                             ids.data_ptr(), dist.data_ptr(), None, None, live.cuda_stream))
        graph.replay()
        torch.cuda.synchronize()
        got = out_d.cpu().numpy()
        assert np.array_equal(bits(got), bits(want)), (replay, int((bits(got) != bits(want)).any(axis=1).sum()))
        got_ids = ids.cpu().numpy().view(np.uint32)[: 1 << 16]
        hit = ids_ref != rta.MISS
        assert np.array_equal(got_ids, ids_ref)
        assert np.array_equal(bits(dist.cpu().numpy()[: 1 << 16][hit]), bits(dist_ref[hit]))


def test_static_scratch_rejects_what_does_not_fit(rta, sphere_bvh):
    import torch
    dev = torch.device("cuda:0")
    lib = rta._lib.lib()
    n = rta._lib.STATIC_MAX_PATHS
    rays_d = torch.zeros((n + 64, 6), dtype=torch.float64, device=dev)
    rays_d[:, 5] = 1.0
    keys_d = torch.zeros((n + 64, 3), dtype=torch.int32, device=dev)
    out_d = torch.zeros((n + 64, 3), dtype=torch.float64, device=dev)
    # the maximum itself is accepted at the reference's depth ...
    rta._lib.check(_static(rta, sphere_bvh, rays_d, keys_d, n, 0, 5, rta.TRAVERSE_FAST, out_d))
    torch.cuda.synchronize()
    # ... one wavefront more, or the same count at a larger depth, is refused with a clear message
    for bad_n, depth in ((n + 1, 5), (n, 6), (1 << 33, 5)):
        st = _static(rta, sphere_bvh, rays_d, keys_d, bad_n, 0, depth, rta.TRAVERSE_FAST, out_d)
        assert st == rta._lib.RT_ERR_INVALID_ARGUMENT
        msg = lib.rt_last_error().decode()
        assert "static scratch holds" in msg and "rt_trace_paths" in msg
    assert _static(rta, sphere_bvh, rays_d, keys_d, 16, 0, 0, rta.TRAVERSE_FAST, out_d) == rta._lib.RT_ERR_INVALID_ARGUMENT
    assert _static(rta, sphere_bvh, rays_d, keys_d, 0, 0, 5, rta.TRAVERSE_FAST, out_d) == 0
