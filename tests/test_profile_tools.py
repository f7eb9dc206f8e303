"""The tools that turn rocprofv3 --pmc output into the physical_* figures of the bench line (tools/pmc_summary.py,
tools/make_traffic.py): every figure on the line must follow from the committed PMC text.
CPU only: parses files under profiles/."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_traffic  # noqa: E402
import pmc_summary  # noqa: E402


def test_kernel_names_keep_their_namespace_parentheses():
    n = pmc_summary.kernel_name
    assert n("void k_trace3<0>(DScene, BatchState, unsigned int const*, unsigned int*)") == "k_trace3<0>"
    assert n("void sp4::k_shade_path<true, false, true>(DScene, RenderParams, BatchState, StageQueues, int)") == "sp4::k_shade_path<true, false, true>"
    # round 3 cut at the FIRST parenthesis: these came out empty and their counters were filed under the kernel above
    assert n("(anonymous namespace)::kb_bounds((anonymous namespace)::Items, unsigned int)") == "(anonymous namespace)::kb_bounds"
    assert n("void (anonymous namespace)::kb_small((anonymous namespace)::Items, HIP_vector_type<unsigned int, 3u> const*, int) [clone .kd]") == \
        "(anonymous namespace)::kb_small"
    assert n("__amd_rocclr_copyBuffer") == "__amd_rocclr_copyBuffer"
    assert n("void k<(int)3>(float*)") == "k<(int)3>"
    assert n("") == "<unnamed>"


def test_round3_c4_text_a_headerless_block_is_kept_apart():
    """profiles/r03_pmc_c4_rdreq.txt carries the 315 builder dispatches as a block without a name, right behind
    k_shade_path: its 2.3e6 requests per dispatch must not replace k_shade_path's 5.84e7."""
    d = make_traffic.parse_pmc(os.path.join(ROOT, "profiles", "r03_pmc_c4_rdreq.txt"))
    k = d["k_shade_path<false, false, true>"]
    assert k["dispatches"] == 14
    assert k["TCC_EA0_RDREQ"] == pytest.approx(5.84e7, rel=0.01)
    assert d["<unnamed>"]["dispatches"] == 315
    assert d["<unnamed>"]["TCC_EA0_RDREQ"] == pytest.approx(2.30e6, rel=0.01)


def test_counter_lines_without_a_header_are_refused(tmp_path):
    p = tmp_path / "pmc.txt"
    p.write_text("   TCC_EA0_RDREQ   10   per-dispatch 5\n")
    with pytest.raises(ValueError, match="before any kernel header"):
        make_traffic.parse_pmc(str(p))
    p.write_text("k_a dispatches=2\n   X 1 per-dispatch 0.5\nk_a dispatches=3\n")
    with pytest.raises(ValueError, match="listed twice"):
        make_traffic.parse_pmc(str(p))


@pytest.mark.parametrize("cfg", ["c2", "c4", "c5"])
def test_committed_traffic_files_follow_from_the_committed_pmc_text(cfg):
    """The newest committed traffic file of each config is what make_traffic.py builds from the committed sources."""
    rnd = next(r for r in ("r06", "r05", "r04", "r03") if os.path.exists(os.path.join(ROOT, "profiles", "%s_%s_traffic.json" % (r, cfg))))
    built = make_traffic.build(os.path.join(ROOT, "profiles"), cfg, rnd)
    committed = json.load(open(os.path.join(ROOT, "profiles", "%s_%s_traffic.json" % (rnd, cfg))))
    assert set(built["kernels"]) == set(committed["kernels"])
    for name, e in built["kernels"].items():
        for key in ("hbm_bytes_per_launch", "hbm_side_GBps", "avg_launch_ms_kernel_trace"):
            assert committed["kernels"][name][key] == pytest.approx(e[key], rel=1e-9), (name, key)
    assert committed["traffic_over_algorithmic"] == pytest.approx(built["traffic_over_algorithmic"], rel=1e-9)
    if rnd >= "r06":  # the stamp that ties the file to a library: the committed buildinfo of the profiled .so
        assert committed["library_sources"] == built["library_sources"] and set(built["library_sources"]) == set(make_traffic.KERNEL_SOURCES)


def test_c4_shade_physical_rate_is_the_recomputed_one():
    """5.84e7 x 128 B + 4.91e6 KB = 12.5 GB per launch / 3.69 ms = 3.39 TB/s (the line said 112 GB/s)."""
    rnd = next(r for r in ("r06", "r05", "r04", "r03") if os.path.exists(os.path.join(ROOT, "profiles", "%s_c4_traffic.json" % r)))
    d = json.load(open(os.path.join(ROOT, "profiles", "%s_c4_traffic.json" % rnd)))
    k = next(v for n, v in d["kernels"].items() if n.split("::")[-1].startswith("k_shade_path"))
    assert 2500.0 < k["hbm_side_GBps"] < 4500.0
    assert 0.5 < k["valu_lane_utilisation"] <= 1.0
