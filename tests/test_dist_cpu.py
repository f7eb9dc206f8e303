"""The N > 1 path on CPU: world_size 2, gloo.  Each rank takes its round-robin tiles
(dartray_amd.dist.shard -> dr_enumerate_pixels), accumulates a full-frame film, and ONE reduce(sum)
merges them on rank 0.  The per-rank compute is done by the CPU oracle here (test infrastructure
standing in for the GPU), so what is tested is the product's sharding logic and the call sequence of
dartray_amd.dist; on GPUs the same reduce_film goes through dr_film_reduce (RCCL, tests/test_gpu_comm.py).
bench.py --gpus N runs exactly this split -- by default on BASELINE configs[2]'s 4096 x 4096 x 1024 spp image (the N-rank
headline since round 5: bench.plan, tests/test_bench_launch.py), followed by the N = 1 workload per GPU as a short extra."""
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


class _FakeLib:
    """Stands in for libdartray_hip in the handshake test: every dr_comm_* call returns what the scenario says for this
    rank (0 = DR_OK) and records that it was made."""

    def __init__(self, rank, fail_at):
        self.rank, self.fail_at, self.calls = rank, fail_at, []

    def _rc(self, name):
        self.calls.append(name)
        return 7 if self.fail_at == (self.rank, name) else 0

    def dr_comm_available(self):
        return self._rc("dr_comm_available")

    def dr_comm_unique_id(self, ptr, n):
        return self._rc("dr_comm_unique_id")

    def dr_comm_init(self, rank, world, ptr, n):
        return self._rc("dr_comm_init")

    def dr_comm_destroy(self):
        return self._rc("dr_comm_destroy")


def _handshake_worker(rank, world, init_file, out_dir, fail_rank, fail_call):
    sys.path.insert(0, ROOT)
    import json
    import torch
    import torch.distributed as dist
    from dartray_amd import _abi, dist as drdist
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    fake = _FakeLib(rank, (fail_rank, fail_call))

    def check(rc):
        if rc != 0:
            raise _abi.DartRayHipError("fake error %d" % rc)

    _abi.lib, _abi.init, _abi.check = (lambda: fake), (lambda dev: None), check
    drdist._FALLBACK_BACKEND = "gloo"
    drdist._comm_init_or_fallback(rank, world, rank)
    ok = None
    if drdist._fallback_group is not None:  # the group every rank dropped to works, on every rank
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t, group=drdist._fallback_group)
        ok = float(t.item())
    json.dump({"calls": fake.calls, "fallback": drdist._fallback_group is not None, "error": drdist.comm_error,
               "ready": drdist._comm_ready, "sum": ok}, open(os.path.join(out_dir, "r%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("fail_rank,fail_call", [(0, "dr_comm_unique_id"), (1, "dr_comm_available"), (1, "dr_comm_init"), (0, "dr_comm_init"),
                                                 (None, None)])
def test_comm_init_or_fallback_is_symmetric_when_one_rank_fails(fail_rank, fail_call):
    """whichever rank fails, and in whichever phase, BOTH ranks end in the fallback group with
    comm_error set -- nobody waits in a broadcast or a collective for a peer that has given up (this test would hang) --
    and a clean handshake leaves both with the RCCL communicator and no fallback."""
    import json
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_handshake_worker, args=(2, os.path.join(tmp, "init"), tmp, fail_rank, fail_call), nprocs=2, join=True)
        res = [json.load(open(os.path.join(tmp, "r%d.json" % r))) for r in range(2)]
    if fail_call is None:
        assert all(not r["fallback"] and r["ready"] and r["error"] is None for r in res)
        assert res[0]["calls"] == ["dr_comm_available", "dr_comm_unique_id", "dr_comm_init"]
        assert res[1]["calls"] == ["dr_comm_available", "dr_comm_init"]
        return
    assert all(r["fallback"] and not r["ready"] and r["error"] for r in res), res
    assert all(r["sum"] == 3.0 for r in res)
    assert "fake error" in res[fail_rank]["error"] and "another rank" in res[1 - fail_rank]["error"]
    if fail_call == "dr_comm_init":  # the rank whose init succeeded gives its communicator back before falling back
        assert res[1 - fail_rank]["calls"][-1] == "dr_comm_destroy"
        assert "dr_comm_destroy" not in res[fail_rank]["calls"]
    else:  # nobody reached ncclCommInitRank
        assert all("dr_comm_init" not in r["calls"] for r in res)


class _FakeDeviceScene:
    """What dist.share_pilot touches of core._DeviceScene: the kernel pair and the state layout, with their setters."""

    def __init__(self, kernels, layout):
        self.kernels, self.layout, self.sets = tuple(kernels), layout, []

    def trace_kernels(self, kernels=None):
        if kernels is not None:
            self.kernels = tuple(kernels)
            self.sets.append(("kernels", tuple(kernels)))
        return self.kernels

    def state_layout(self, layout=None):
        if layout is not None:
            self.layout = layout
            self.sets.append(("layout", layout))
        return self.layout, 0.5


def _pilot_worker(rank, world, init_file, out_dir, src_picks):
    sys.path.insert(0, ROOT)
    import json
    import torch.distributed as dist
    from dartray_amd import dist as drdist
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    # every rank's own pilot would have said something else; rank 0's is what counts
    dev = _FakeDeviceScene(src_picks[:2], src_picks[2]) if rank == 0 else _FakeDeviceScene((0, 0), 0)
    picks = drdist.share_pilot(dev)
    allp = drdist.gather_picks(dev)
    json.dump({"picks": picks, "kernels": dev.kernels, "layout": dev.layout, "sets": dev.sets, "all": allp}, open(os.path.join(out_dir, "r%d.json" % rank), "w"))
    dist.destroy_process_group()


@pytest.mark.parametrize("src_picks", [(5, 3, 4), (2, 2, 64), (0, 0, 0)])
def test_ranks_take_rank_zeros_pilot_picks(src_picks):
    """dist.share_pilot: rank 0's kernel pair and state layout reach every rank over the gloo control
    plane and are stored there; (0, 0, 0) -- rank 0 ran no pilot -- stores nothing."""
    import json
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_pilot_worker, args=(2, os.path.join(tmp, "init"), tmp, src_picks), nprocs=2, join=True)
        r0, r1 = (json.load(open(os.path.join(tmp, "r%d.json" % k))) for k in (0, 1))
    assert tuple(r0["picks"]) == tuple(r1["picks"]) == tuple(src_picks)
    assert r0["sets"] == []                                            # the source rank keeps what it measured
    if src_picks == (0, 0, 0):
        assert r1["sets"] == [] and tuple(r1["kernels"]) == (0, 0)
    else:
        assert [tuple(x) if isinstance(x, list) else x for _, x in r1["sets"]] == [tuple(src_picks[:2]), src_picks[2]]
        assert tuple(r1["kernels"]) == tuple(src_picks[:2]) and r1["layout"] == src_picks[2]
    assert r0["all"] == r1["all"] and len(r0["all"]) == 2 and len({tuple(p) for p in r0["all"]}) == 1
