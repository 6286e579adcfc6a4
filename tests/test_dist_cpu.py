"""The N > 1 path on CPU: two gloo ranks run the same shard/merge logic bench.py uses
on the GPUs (ray-tracer_amd/sharding.py). No GPU and no tracing here: each rank fills
its shard with values that are a known function of (pixel, sample)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

W, H, TILE, SPP = 80, 48, 16, 6


def pixel_value(sample):
    """A stand-in for one sample's radiance at every pixel: distinct per pixel and sample."""
    y, x = np.mgrid[0:H, 0:W]
    return np.stack([(y * W + x) * 1e-3 + sample, np.sin(0.1 * x) + 0.25 * sample, (x + y) % 7 + 0.5 * sample],
                    axis=-1)


def worker(rank, world, port, out):
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import ray_tracer_amd as rta
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # tile shards: RGBA8 frames, zero outside the owned tiles
        mask = rta.sharding.owned_mask(W, H, rank, world, TILE)
        frame = np.zeros((H, W, 4), dtype=np.uint8)
        y, x = np.mgrid[0:H, 0:W]
        frame[..., 0], frame[..., 1], frame[..., 2], frame[..., 3] = (x * 3) % 256, (y * 5) % 256, (x + y) % 256, 255
        frame[~mask] = 0
        t = torch.from_numpy(frame)
        rta.sharding.combine_tiles(t, dist)
        # sample-range shards: each rank averages its own SPP samples of the full frame
        s0 = rta.sharding.sample_range(rank, SPP)
        mean = torch.from_numpy(np.mean([pixel_value(s) for s in range(s0, s0 + SPP)], axis=0))
        rta.sharding.combine_samples(mean, dist)
        covered = torch.from_numpy(mask.astype(np.int32))
        dist.all_reduce(covered)
        if rank == 0:
            np.savez(out, tiles=t.numpy(), mean=mean.numpy(), covered=covered.numpy())
    finally:
        dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_shard_and_merge_with_gloo(tmp_path, world):
    out = str(tmp_path / "merged.npz")
    mp.spawn(worker, args=(world, free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    assert (got["covered"] == 1).all()  # every pixel belongs to exactly one rank
    y, x = np.mgrid[0:H, 0:W]
    assert np.array_equal(got["tiles"][..., 0], (x * 3) % 256) and (got["tiles"][..., 3] == 255).all()
    want = np.mean([pixel_value(s) for s in range(world * SPP)], axis=0)
    assert np.allclose(got["mean"], want, rtol=1e-13)


def test_owned_mask_matches_round_robin(rta):
    m = [rta.sharding.owned_mask(100, 70, r, 8, 32) for r in range(8)]
    assert np.sum(m, axis=0).min() == 1 and np.sum(m, axis=0).max() == 1
    assert m[0][0, 0] and m[1][0, 32] and m[4][32, 0]  # 4 tiles per row: tile 4 starts row 2
