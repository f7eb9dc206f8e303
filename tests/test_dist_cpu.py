"""The N > 1 path on CPU: world_size 2, gloo.  Each rank takes its round-robin tiles
(dartray_amd.dist.shard -> dr_enumerate_pixels), accumulates a full-frame film, and ONE reduce(sum)
merges them on rank 0.  The per-rank compute is done by the CPU oracle here (test infrastructure
standing in for the GPU), so what is tested is the product's sharding logic and the call sequence of
dartray_amd.dist; on GPUs the same reduce_film goes through dr_film_reduce (RCCL, tests/test_gpu_comm.py).
bench.py --gpus N runs exactly this split (on configs[2]'s 4096 x 4096 x 1024 spp image by default)."""
import os
import sys
import tempfile

import numpy as np
import pytest

from conftest import ROOT


def _worker(rank, world, init_file, out_file, weak=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from dartray_amd import scenes, dist as drdist
    import oracle.binding as ob
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    assert drdist.max_over_ranks(float(rank)) == float(world - 1)
    prims, mk = scenes.config("C2", xres=40, yres=40, spp=4, blob=(12, 6))
    r = drdist.sample_set(mk(), rank) if weak else drdist.shard(mk(), rank, world, tile_size=16)
    px = r.pixels()
    osc = ob.OracleScene(prims)
    film = osc.render(ob.render_desc(r, sampler_mode=1, pixels=px))["film"]
    t = torch.from_numpy(film.copy())
    drdist.barrier()
    drdist.reduce_film(t, 0)
    if rank == 0:
        np.save(out_file, t.numpy())
    dist.destroy_process_group()


def test_two_rank_tile_sharding_and_film_reduce(ob):
    import torch.multiprocessing as mp
    from dartray_amd import scenes
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, "init")
        out_file = os.path.join(tmp, "film.npy")
        mp.spawn(_worker, args=(2, init_file, out_file), nprocs=2, join=True)
        merged = np.load(out_file)
    prims, mk = scenes.config("C2", xres=40, yres=40, spp=4, blob=(12, 6))
    single = ob.OracleScene(prims).render(ob.render_desc(mk(), sampler_mode=1))["film"]
    # box filter radius 0.5: tiles are disjoint, the sum only adds zeros => bit-exact
    assert np.array_equal(merged, single)
    assert np.all(merged[..., 3] >= 4)


def test_two_rank_sample_sets_and_film_reduce(ob):
    """bench.py --scaling samples: each rank renders the whole image with sampler seed + rank, one reduce(sum)."""
    import torch.multiprocessing as mp
    from dartray_amd import scenes
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, "init")
        out_file = os.path.join(tmp, "film.npy")
        mp.spawn(_worker, args=(2, init_file, out_file, True), nprocs=2, join=True)
        merged = np.load(out_file)
    prims, mk = scenes.config("C2", xres=40, yres=40, spp=4, blob=(12, 6))
    osc = ob.OracleScene(prims)
    films = []
    for rank in range(2):
        r = mk()
        r.sampler.seed += rank
        films.append(osc.render(ob.render_desc(r, sampler_mode=1))["film"])
    assert not np.array_equal(films[0], films[1])
    assert np.array_equal(merged, films[0] + films[1])  # one f32 add per element either way
    assert np.all(merged[..., 3] == 8)


def test_init_process_group_single_rank_is_a_noop(monkeypatch):
    from dartray_amd import dist as drdist
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    assert drdist.init_process_group() == (0, 1, 0)
    import torch
    t = torch.ones(3)
    assert drdist.reduce_film(t) is t
