"""The multi-GPU half of the C ABI on hardware: dr_comm_init / dr_film_reduce / dr_comm_allreduce_f64 (librccl bound
at run time, ncclReduce over xGMI).  World 1: the same code path is the identity.  World 2: two processes, one per
GPU, each rendering its round-robin tile share into a full-frame device film, merged by ONE dr_film_reduce -- equal to
the single-GPU film bit for bit (skipped on a 1-GPU box)."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_world_one_film_reduce_is_the_identity_through_rccl(gpu):
    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
        "from dartray_amd import _abi, dist as drdist\n"
        "torch.cuda.set_device(0)\n"
        "rank, world, local = drdist.init_process_group(device_comm=True)\n"
        "lib = _abi.lib()\n"
        "assert (rank, world) == (0, 1) and lib.dr_comm_world() == 1 and lib.dr_comm_rank() == 0\n"
        "g = torch.Generator(device='cuda'); g.manual_seed(3)\n"
        "film = torch.rand((257, 129, 4), generator=g, device='cuda', dtype=torch.float32)\n"
        "ref = film.clone()\n"
        "drdist.reduce_film(film, 0)\n"
        "torch.cuda.synchronize()\n"
        "assert torch.equal(film, ref)\n"
        "assert drdist.max_over_ranks(1.25) == 1.25\n"
        "assert lib.dr_film_reduce(film.data_ptr(), film.numel() // 4, 1, None) == -1  # root out of range\n"
        "assert lib.dr_comm_init(0, 1, None, 128) == -1  # a communicator already exists\n"
        "drdist.comm_destroy()\n"
        "assert lib.dr_comm_world() == 0\n"
        "print('OK')\n" % ROOT)
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=500)
    assert res.returncode == 0 and "OK" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]


def test_render_sharded_without_a_communicator_is_dr_render(gpu):
    """dr_render_sharded: no communicator (or a world of 1) => this rank is the root and the merge is skipped."""
    from dartray_amd import scenes
    prims, mk = scenes.config("C2", xres=48, yres=40, spp=8, blob=(24, 12))
    scene = scenes.make_scene(prims)
    a = mk().render(scene)
    b = mk().render_sharded(scene)
    assert np.array_equal(a.film, b.film) and np.array_equal(a.rgb, b.rgb)
    r = mk()
    d, keep = r.describe()
    import ctypes as C
    assert gpu.lib().dr_render_sharded(scene._device().handle, C.byref(d), 1, None, None) == -1  # root out of range
    assert b"root" in gpu.lib().dr_last_error()


_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import torch
from dartray_amd import _abi, scenes, dist as drdist
local = drdist.local_device(int(os.environ["LOCAL_RANK"]))
torch.cuda.set_device(local)
rank, world, _ = drdist.init_process_group()
prims, mk = scenes.config("C2", xres=160, yres=128, spp=16, blob=(60, 30))
r = drdist.shard(mk(), rank, world)
scene = scenes.make_scene(prims)
dev = scene._device()
H, W = r.camera.film.height, r.camera.film.width
film = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
# rank 0 renders first (DARTRAY_PILOT=force: its render runs the pilot), its picks travel to rank 1, which then runs no pilot
if rank == 0:
    r.render_device(scene, film.data_ptr(), stream)
    torch.cuda.synchronize()
    assert dev.last_render_info()["pilot_batches"] >= 3
picks = drdist.share_pilot(dev)
if rank != 0:
    r.render_device(scene, film.data_ptr(), stream)
    torch.cuda.synchronize()
    assert dev.last_render_info()["pilot_batches"] == 0, dev.last_render_info()
assert picks[0] in (2, 3, 5) and picks[1] in (2, 3, 6, 7) and picks[2] in (4, 64), picks
assert dev.trace_kernels() == picks[:2] and dev.state_layout()[0] == picks[2]
allp = drdist.gather_picks(dev)
assert len(allp) == world and len(set(allp)) == 1, allp     # both ranks run the same kernels in the same layout
info = dev.last_render_info()
assert (info["closest_kernel"], info["any_hit_kernel"], info["state_layout"]) == picks, (info, picks)
drdist.reduce_film(film, 0, stream)
t = drdist.max_over_ranks(float(rank + 1))
torch.cuda.synchronize()
assert t == float(world), t
if rank == 0:
    np.save(sys.argv[1], film.cpu().numpy())
if not drdist.rehearsal():
    out = r.render_sharded(scene, 0)   # the same through the one-call entry point of the C ABI
    assert (out is not None) == (rank == 0)
    if rank == 0:
        assert np.array_equal(out.film, film.cpu().numpy())
drdist.barrier()
drdist.comm_destroy()
torch.distributed.destroy_process_group()
"""


@pytest.mark.parametrize("rehearsal", [False, True])
def test_two_ranks_tile_shards_merged_by_dr_film_reduce(gpu, rehearsal):
    """rehearsal=True (any box): the two ranks SHARE a GPU and gloo sums the film through host memory (DARTRAY_COMM_REHEARSAL,
    dartray_amd/dist.py) -- everything of the world-2 path but the RCCL calls runs on hardware: the tile split, each rank's device
    render into a full-frame film, the reduce's call sites, the max over ranks."""
    import torch
    if not rehearsal and torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (the driver's multi-GPU tier); the world-2 logic is covered on CPU by tests/test_dist_cpu.py")
    from dartray_amd import scenes
    with tempfile.TemporaryDirectory() as tmp:
        script = os.path.join(tmp, "worker.py")
        out = os.path.join(tmp, "film.npy")
        open(script, "w").write(_WORKER % {"root": ROOT})
        env = dict(os.environ, DARTRAY_PILOT="force")  # (a small scene: the pilot has to be asked for)
        env.pop("DARTRAY_TRACE_IMPL", None)
        env.pop("DARTRAY_STATE_LAYOUT", None)
        if rehearsal:
            env["DARTRAY_COMM_REHEARSAL"] = "1"
        res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                              "--master-addr", "127.0.0.1", "--master-port", "29532" if rehearsal else "29531", script, out],
                             capture_output=True, text=True, timeout=900, env=env)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
        merged = np.load(out)
    prims, mk = scenes.config("C2", xres=160, yres=128, spp=16, blob=(60, 30))
    single = mk().render(scenes.make_scene(prims)).film
    assert np.array_equal(merged, single)   # (and inside the worker: both ranks report the same kernel pair and state layout)
