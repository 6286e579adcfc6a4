"""Splitting one image over the GPUs of a node and merging the results.

The reference splits pixels over 7 threads as (y*RES+x) % NUM_THREADS and merges
the threads' full-size RGBA frames by "non-zero wins" (main.rs:55, 87-100). Here
the unit is a tile_size x tile_size tile dealt round-robin over the ranks, or a
contiguous sample range per rank; either way the scene is replicated, no ray
crosses devices, and the ranks meet exactly once per frame in one collective.
The tensors may live on the GPU (backend nccl = RCCL) or on the CPU (gloo).
"""
import numpy as np


def owned_mask(width, height, shard_index, shard_count, tile_size):
    """Boolean (height, width) mask of the pixels rt_film_create assigns to a shard."""
    tiles_x = (width + tile_size - 1) // tile_size
    y, x = np.mgrid[0:height, 0:width]
    tile = (y // tile_size) * tiles_x + x // tile_size
    return (tile % shard_count) == shard_index


def sample_range(rank, samples_per_rank):
    """First sample index of a rank when the ranks render consecutive sample ranges."""
    return rank * samples_per_rank


def combine_tiles(rgba, dist=None, dst=0):
    """Tile shards: every rank holds a full-size RGBA8 frame that is zero outside its
    tiles, so a sum-reduce to `dst` is the merge."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(rgba, dst=dst, op=dist.ReduceOp.SUM)
    return rgba


def combine_samples(radiance_mean, dist=None, dst=0):
    """Sample-range shards with equal sample counts: the frame's mean radiance is the
    mean of the ranks' means. Returns the combined tensor (valid on `dst`)."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(radiance_mean, dst=dst, op=dist.ReduceOp.SUM)
        if dist.get_rank() == dst:
            radiance_mean.mul_(1.0 / dist.get_world_size())
    return radiance_mean
