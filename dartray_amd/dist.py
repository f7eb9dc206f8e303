"""Multi-GPU sharding of SamplerRenderer.render.

The reference shards a render by image sub-window, one web-worker isolate per task, each with
its own scene copy, and merges by copying the disjoint rectangles into the output
(lib/dartray_web/render_manager.dart:100-141; GetSubWindow, lib/core/common.dart:52-73).  Here:
one process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI), the full scene
replicated on every GPU, 32x32-pixel tiles (TilePixelSampler.tileSize, tile_pixel_sampler.dart:37)
dealt round-robin over ranks for load balance, every rank accumulating into a zero-initialised
full-frame (X, Y, Z, weight) film and ONE reduce(sum) of that film to rank 0 per render.  With
the box filter of radius 0.5 the tiles are disjoint, so the sum adds zeros and is exact; with
wider filters the sum also carries the splats across tile borders that the reference's
rectangle copy drops.  `sample_set` is the weak-scaling alternative (bench.py's default at N > 1): every rank
renders the whole image with its own sampler seed and the same reduce adds the sample sets up.  torch is plumbing here (device memory, streams, the collective).
"""
import os

import torch
import torch.distributed as dist


def init_process_group():
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard(renderer, rank, world, tile_size=32):
    """Give `renderer` this rank's share of the image: round-robin tiles."""
    renderer.tileRank, renderer.tileCount, renderer.tileSize = rank, world, tile_size
    return renderer


def sample_set(renderer, rank):
    """Weak scaling: every rank renders ALL pixels with its own sample set (sampler seed + rank); the summed
    films are the image at world x spp samples per pixel (ImageFilm keeps weighted sums, image_film.dart:99-185)."""
    renderer.sampler.seed = int(renderer.sampler.seed) + int(rank)
    return renderer


def reduce_film(film, dst=0):
    """Sum the per-rank (X, Y, Z, weight) films onto rank `dst` (one collective per render)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    return film


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
