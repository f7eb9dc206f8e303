"""bench.py's launch contract, without a GPU: `python bench.py --gpus N` starts one rank per GPU itself (the driver's
own command line: torch.distributed.run on 127.0.0.1) unless it already runs under torchrun, and refuses a world size
that differs from --gpus.  The N-rank HEADLINE is BASELINE configs[2] (C3: the C2 scene at 4096 x 4096, 1024 spp, its 32 x 32
tiles over the N ranks -- round 5; it used to be the N = 1 workload per GPU, which now follows as extra_configs[0]).
Mirrors the fan-out of lib/dartray_web/render_manager.dart:100-141 (N workers, one sub-image each)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402  (imports nothing heavy at module level)


def test_single_gpu_run_is_not_a_launcher():
    assert bench.check_world(1, 0) is None


def test_gpus_n_outside_torchrun_launches_the_ranks():
    assert bench.check_world(8, 0) == "launch"
    cmd = bench.rank_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"] and cmd[-7] == os.path.join(ROOT, "bench.py")


def test_a_rank_of_the_right_world_runs():
    assert bench.check_world(4, 4) is None


@pytest.mark.parametrize("gpus,world", [(4, 2), (1, 2), (2, 8)])
def test_world_size_that_differs_from_gpus_is_refused(gpus, world):
    with pytest.raises(SystemExit) as e:
        bench.check_world(gpus, world)
    assert "launch one rank per GPU" in str(e.value)


def test_refusal_through_the_command_line():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "launch one rank per GPU" in r.stderr


def test_weak_scaling_keeps_the_per_gpu_pixel_count():
    for n in (2, 4, 8):
        side = bench.weak_resolution(1024, n)
        assert side % 32 == 0
        assert abs(side * side / (n * 1024 * 1024) - 1.0) < 0.03
    assert bench.weak_resolution(1024, 4) == 2048


def test_the_n_rank_headline_is_baseline_configs_2():
    """`--gpus N` with N > 1 and nothing else: the headline Run is C3 = (4096, 1024 spp, tiles over the ranks, total work
    fixed), whatever --steps / --warmup the driver passes; the N = 1 workload per GPU follows as a short extra.  N = 1 keeps
    configs[1] with the short C4 / C5 runs."""
    for n in (2, 4, 8):
        pl = bench.plan(n)
        assert pl["headline"] == ("C3", 4096, 1024, "c3")
        assert pl["extras"] == [("C2", bench.weak_resolution(1024, n), 256, "weak", 2, 1)]
    assert bench.plan(8, no_extra=True)["extras"] == []
    assert bench.plan(8, scaling="weak") == {"headline": ("C2", bench.weak_resolution(1024, 8), 256, "weak"), "extras": []}
    assert bench.plan(2, scaling="strong-c2")["headline"] == ("C2", 1024, 256, "strong-c2")
    pl = bench.plan(1)
    assert pl["headline"] == ("C2", 1024, 256, "weak") and [e[0] for e in pl["extras"]] == ["C4", "C5"]
    assert bench.plan(1, config="C5") == {"headline": ("C5", 2048, 512, "weak"), "extras": []}
    # the argument path of the driver's N = 2 command line ends in that plan
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "pl = plan(world, args.scaling, args.config, args.res, args.spp, args.no_extra)" in src
    assert 'cfg, res, spp, mode = pl["headline"]' in src and "Run(cfg, res, spp, rank, world, mode, args)" in src


def test_the_n_rank_line_names_its_single_gpu_reference():
    ref = bench.one_gpu_reference("weak", "C2")
    finals = sorted(n for n in os.listdir(os.path.join(ROOT, "profiles")) if n.endswith("_bench_final.json"))
    assert ref["source"] == "profiles/" + finals[-1] and ref["value"] > 100 and ref["unit"] == "Msamples/s"
    ref = bench.one_gpu_reference("c3", "C3")
    c3 = sorted(n for n in os.listdir(os.path.join(ROOT, "profiles")) if n.endswith("_c3_one_gpu.json"))
    assert "4096x4096" in ref["workload"] and ref["source"] == "profiles/" + c3[-1] and 500 < ref["value"] < 1000  # (the committed whole-C3 line)
