"""Multi-GPU sharding of SamplerRenderer.render.

The reference shards a render by image sub-window, one web-worker isolate per task, each with
its own scene copy, and merges by copying the disjoint rectangles into the output
(lib/dartray_web/render_manager.dart:100-141; GetSubWindow, lib/core/common.dart:52-73).  Here:
one process per GPU, the full scene replicated on every GPU, 32x32-pixel tiles
(TilePixelSampler.tileSize, tile_pixel_sampler.dart:37) dealt round-robin over ranks for load balance,
every rank accumulating into a zero-initialised full-frame (X, Y, Z, weight) film and ONE reduce(sum)
of that film to rank 0 per render.  With the box filter of radius 0.5 the tiles are disjoint, so the
sum adds zeros and is exact; with wider filters the sum also carries the splats across tile borders
that the reference's rectangle copy drops.

The collective is the C ABI's: dr_comm_init / dr_film_reduce (include/dartray_hip.h) call librccl's
ncclReduce over xGMI directly, exactly what a Dart host would call through dart:ffi.  torchrun is only the
launcher here and torch.distributed (gloo) only the control plane: it carries the 128-byte RCCL unique id
from rank 0 to the other ranks and provides the host barrier around a timed region.  `sample_set` is an
alternative split (bench.py --scaling samples): every rank renders the whole image with its own sampler
seed and the same reduce adds the sample sets up.

CPU tensors (the world-2 gloo tests, which stand the CPU oracle in for the GPU) are reduced with gloo: test
plumbing for the sharding logic, never a product path.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _abi

_comm_ready = False
# bench.py only: if dr_comm_init fails on a multi-GPU box (the world > 1 path of dr_comm.cpp has never met hardware in
# development: one-GPU boxes), the film is reduced by torch.distributed's own RCCL group instead and the JSON line says
# so, with the error -- a measured scaling curve plus a bug report instead of no curve.  Never taken silently.
_fallback_group = None
comm_error = None
# DARTRAY_COMM_REHEARSAL=1 (development only, never a measurement): the N ranks SHARE the visible GPUs (rank r on device r mod
# count -- RCCL refuses two ranks on one device) and the film is summed through host memory by gloo.  It exists so that the
# N-rank code of bench.py and of this module -- tile split, per-rank renders, the reduce's call sites, the JSON line -- can
# run on a one-GPU box; the collective itself is NOT the product's and the line says so.
_rehearsal = False


def rehearsal():
    return os.environ.get("DARTRAY_COMM_REHEARSAL") == "1"


def local_device(local):
    """The device index of local rank `local` (identity, except under DARTRAY_COMM_REHEARSAL)."""
    if rehearsal():
        return local % max(1, torch.cuda.device_count())
    return local


def init_process_group(device_comm=None):
    """Initialise from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Returns
    (rank, world, local_rank).  world > 1: a gloo control-plane group; on a GPU box also dr_init(local_rank) and the
    RCCL communicator of the C ABI (device_comm=False skips it, True also builds it for world == 1)."""
    global _comm_ready
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    if device_comm is None:
        device_comm = world > 1 and torch.cuda.is_available()
    if device_comm and not _comm_ready and world > 1 and rehearsal():
        global _fallback_group, comm_error, _rehearsal
        _abi.init(local_device(local))
        comm_error = "REHEARSAL (DARTRAY_COMM_REHEARSAL=1): ranks share a GPU, the film is summed through host memory by gloo"
        _fallback_group = dist.new_group(backend="gloo")
        _rehearsal = True
    elif device_comm and not _comm_ready:
        if world > 1 and os.environ.get("DARTRAY_COMM_FALLBACK") == "1":
            _comm_init_or_fallback(rank, world, local)
        else:
            comm_init(rank, world, local)
    return rank, world, local


def _any_rank(flag):
    """gloo all-reduce(MAX) of a host flag: did ANY rank raise it?  Every rank calls this at the same point."""
    t = torch.tensor([1 if flag else 0], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(int(t.item()))


def _comm_init_or_fallback(rank, world, local):
    """dr_comm_init on every rank -- or, if ANY rank cannot, every rank together on a torch.distributed RCCL group (see
    _fallback_group).  The handshake is symmetric phase by phase: each phase is a LOCAL attempt on the ranks it
    concerns, then one gloo all-reduce of an error flag that every rank joins, and only a clean phase is followed by the
    next collective -- so no rank is ever left waiting in a broadcast or in ncclCommInitRank for a peer that has given up:
      1. every rank: dr_init + dr_comm_available (binds librccl, checks its version; talks to nobody)   -> agree
      2. rank 0: dr_comm_unique_id                                                                    -> agree
      3. every rank: broadcast of the 128-byte id (gloo), then dr_comm_init (ncclCommInitRank)         -> agree
    A failure INSIDE ncclCommInitRank on a subset of ranks can still block the others in RCCL: nothing on the host can
    undo that, and the launcher's timeout ends the job non-zero -- this process never re-executes itself."""
    global _fallback_group, comm_error, _comm_ready
    lib = None
    err = None

    def attempt(fn):
        nonlocal err
        if err is None:
            try:
                fn()
            except Exception as e:  # noqa: BLE001 -- reported on the JSON line
                err = "%s: %s" % (type(e).__name__, e)

    def fall_back(stage):
        global _fallback_group, comm_error
        comm_error = err or "%s failed on another rank" % stage
        if _comm_ready:
            comm_destroy()
        _fallback_group = dist.new_group(backend=_FALLBACK_BACKEND)

    def phase1():
        nonlocal lib
        lib = _abi.lib()
        _abi.init(local)
        _abi.check(lib.dr_comm_available())

    attempt(phase1)
    if _any_rank(err):
        return fall_back("dr_comm_available")
    ident = torch.zeros(_abi.DR_COMM_ID_BYTES, dtype=torch.uint8)
    if rank == 0:
        attempt(lambda: _abi.check(lib.dr_comm_unique_id(ident.data_ptr(), ident.numel())))
    if _any_rank(err):
        return fall_back("dr_comm_unique_id")
    dist.broadcast(ident, src=0)

    def phase3():
        global _comm_ready
        _abi.check(lib.dr_comm_init(rank, world, ident.data_ptr(), ident.numel()))
        _comm_ready = True

    attempt(phase3)
    if _any_rank(err):
        return fall_back("dr_comm_init")


_FALLBACK_BACKEND = "nccl"  # (the CPU test of the handshake swaps in gloo)


def comm_init(rank, world, local):
    """dr_comm_init on every rank: rank 0 draws the unique id, the control plane broadcasts it."""
    global _comm_ready
    lib = _abi.lib()
    _abi.init(local)
    ident = torch.zeros(_abi.DR_COMM_ID_BYTES, dtype=torch.uint8)
    if rank == 0:
        _abi.check(lib.dr_comm_unique_id(ident.data_ptr(), ident.numel()))
    if world > 1:
        dist.broadcast(ident, src=0)
    _abi.check(lib.dr_comm_init(rank, world, ident.data_ptr(), ident.numel()))
    _comm_ready = True


def comm_destroy():
    global _comm_ready
    if _comm_ready:
        _abi.check(_abi.lib().dr_comm_destroy())
        _comm_ready = False


def shard(renderer, rank, world, tile_size=32):
    """Give `renderer` this rank's share of the image: round-robin tiles."""
    renderer.tileRank, renderer.tileCount, renderer.tileSize = rank, world, tile_size
    return renderer


def sample_set(renderer, rank):
    """Every rank renders ALL pixels with its own sample set (sampler seed + rank); the summed
    films are the image at world x spp samples per pixel (ImageFilm keeps weighted sums, image_film.dart:99-185)."""
    renderer.sampler.seed = int(renderer.sampler.seed) + int(rank)
    return renderer


def share_pilot(dev, src=0):
    """Make the ranks agree on what rank `src`'s pilot picked.  Call it after rank `src` has rendered the scene once and BEFORE the
    other ranks render it: the (closest-hit, any-hit) traversal kernels and the path-state layout of `dev` on rank `src` travel over
    the control plane (three integers, gloo) and the other ranks store them with dr_scene_set_trace_kernels /
    dr_scene_set_state_layout -- their first render then runs no calibration batches of its own.  Every kernel and layout is
    bit-exact, so this changes no film; it changes the job's time: a step is the max over ranks (the reference waits for its slowest
    task too, lib/dartray_web/render_manager.dart:100-141), and a rank whose own pilot -- a few small launches, sharing the node with
    seven other ranks' -- picks the slower kernel (12 % on C2-class scenes) sets the step for everyone.  Returns the picks
    (closest, any_hit, layout); (0, 0, 0) = rank `src` ran no pilot (a small render): nothing is stored."""
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    t = torch.zeros(3, dtype=torch.int32)
    if rank == src:
        k = dev.trace_kernels()
        t[0], t[1], t[2] = int(k[0]), int(k[1]), int(dev.state_layout()[0])
    if world > 1:
        dist.broadcast(t, src=src)
    picks = (int(t[0]), int(t[1]), int(t[2]))
    if rank != src:
        if picks[0] and picks[1]:
            dev.trace_kernels(picks[:2])
        if picks[2]:
            dev.state_layout(picks[2])
    return picks


def gather_picks(dev):
    """What every rank's scene ended up with, as a list of (closest, any_hit, layout) per rank (control plane; diagnostics)."""
    k = dev.trace_kernels()
    mine = torch.tensor([int(k[0]), int(k[1]), int(dev.state_layout()[0])], dtype=torch.int32)
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if world == 1:
        return [tuple(int(v) for v in mine)]
    out = [torch.zeros(3, dtype=torch.int32) for _ in range(world)]
    dist.all_gather(out, mine)
    return [tuple(int(v) for v in o) for o in out]


def reduce_film(film, dst=0, stream=None):
    """Sum the per-rank (X, Y, Z, weight) films onto rank `dst` (one collective per render).  Device films go
    through dr_film_reduce (RCCL) on `stream` (default: torch's current stream)."""
    if film.is_cuda:
        if _fallback_group is not None and _rehearsal:  # (gloo reduces host tensors only)
            host = film.cpu()
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM, group=_fallback_group)
            if dist.get_rank() == dst:
                film.copy_(host)
            return film
        if _fallback_group is not None:
            dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM, group=_fallback_group)
            return film
        if not _comm_ready:
            if int(os.environ.get("WORLD_SIZE", "1")) > 1:
                raise _abi.DartRayHipError("reduce_film: dr_comm_init has not run (init_process_group on a GPU box does it)")
            return film
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        assert film.dtype == torch.float32 and film.is_contiguous() and film.shape[-1] == 4
        _abi.check(_abi.lib().dr_film_reduce(film.data_ptr(), film.numel() // 4, dst, stream))
        return film
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)  # CPU tensors: the gloo tests only
    return film


def max_over_ranks(value, stream=None):
    """max over ranks of a host float (the timed region of bench.py): ncclAllReduce(max) on the device when the
    RCCL communicator exists, gloo otherwise."""
    if _comm_ready and torch.cuda.is_available():
        t = torch.tensor([value], dtype=torch.float64, device="cuda")
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        _abi.check(_abi.lib().dr_comm_allreduce_f64(t.data_ptr(), 1, 1, stream))
        return float(t.item())
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([value], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return value


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
