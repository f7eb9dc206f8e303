"""The ONE JSON line bench.py prints must stay something the driver can parse: numbers only, under 8 000 bytes, the last
line of stdout, with the contract's keys (round 5's 26 KB line came back `parsed: null`).  The
line is a pure function (bench.compact_line) of the full result, which goes to a sidecar file the line names; here it is
built from committed full results of both shapes -- the N = 1 line and the N-rank line -- without a GPU."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402

FULL_N1 = os.path.join(ROOT, "profiles", "r05_bench_final.json")              # 26 KB: the line the driver could not parse
FULL_N2 = os.path.join(ROOT, "profiles", "r05_rehearsal_two_ranks_one_gpu.json")  # the --gpus 2 shape (rehearsal)

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")
ROOFLINE = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "saturated", "traffic_over_algorithmic", "frac_physical_of_copy",
            "frac_of_binding_ceiling", "avg_launch_ms", "launches", "alg_bytes_per_launch")


def strings_in(o):
    if isinstance(o, str):
        yield o
    elif isinstance(o, dict):
        for v in o.values():
            yield from strings_in(v)
    elif isinstance(o, list):
        for v in o:
            yield from strings_in(v)


@pytest.mark.parametrize("path", [FULL_N1, FULL_N2])
def test_line_is_small_and_carries_the_contract(path):
    full = bench.load_detail(path)
    full.setdefault("scaling", "weak")  # (the round-5 N = 1 result predates the key)
    line = bench.compact_line(full, "gpurun_out/bench_detail_c2_n1.json")
    text = json.dumps(line)
    assert len(text) < bench.LINE_LIMIT == 8000
    assert len(text) < 4000  # the usual size: room for eight ranks' worth of keys
    for k in CONTRACT:
        assert k in line, k
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"] and line["n_gpus"] == full["n_gpus"]
    assert isinstance(line["config"]["workload"], str) and "model" not in line["config"]
    for k in ROOFLINE:
        assert k in line["roofline"], k
    assert line["roofline"]["frac"] == pytest.approx(line["roofline"]["achieved"] / line["roofline"]["peak"], abs=1e-3)
    assert line["detail"] == "gpurun_out/bench_detail_c2_n1.json"
    # numbers, not paragraphs: no string on the line is longer than the workload name
    assert max(len(s) for s in strings_in(line)) <= max(160, len(line["config"]["workload"]))
    assert all(len(s) <= 130 for s in strings_in({k: v for k, v in line.items() if k != "config"}))


def test_n1_line_has_cpu_baseline_and_extras():
    line = bench.compact_line(bench.load_detail(FULL_N1))
    cb = line["cpu_baseline"]
    assert cb["cores"] == 1 and cb["kind"] == "port" and cb["value"] > 0 and cb["unit"] == "Msamples/s" and cb["sample"]
    assert [e["workload"] for e in line["extra_configs"]] == ["C4", "C5"]
    for e in line["extra_configs"]:
        assert set(e) >= {"workload", "value", "ms_per_step", "steps", "roofline_frac", "frac_physical_of_copy"}
    assert line["roofline"]["frac_of_binding_ceiling"] == 0.556 and line["roofline"]["kernel"] == "k_trace<0>"


def test_n_rank_line_names_the_collective_without_prose():
    full = bench.load_detail(FULL_N2)
    line = bench.compact_line(full)
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["collective"] == "gloo-rehearsal" and line["rehearsal"] is True
    assert line["reduce_ms"] == full["reduce_ms"] and set(line["per_rank_step_ms"]) == {"min", "max"}
    assert line["extra_configs"][0]["workload"] == "C2" and line["extra_configs"][0]["scaling"] == "weak"


def test_the_limit_is_enforced_not_hoped_for():
    full = bench.load_detail(FULL_N1)
    full["extra_configs"] = full["extra_configs"] * 40  # a run with eighty extras
    line = bench.compact_line(full)
    assert len(json.dumps(line)) < bench.LINE_LIMIT and "roofline" in line and "cpu_baseline" in line
    full["config"]["workload"] = "x" * 9000
    with pytest.raises(AssertionError, match="limit"):
        bench.compact_line(full)


def test_the_line_is_the_last_thing_on_stdout(tmp_path, capsys, monkeypatch):
    monkeypatch.setenv("DARTRAY_BENCH_DETAIL_DIR", str(tmp_path))
    full = bench.load_detail(FULL_N1)
    print("library chatter before the line")
    bench.emit(full)
    out = capsys.readouterr().out.splitlines()
    line = json.loads(out[-1])
    assert line["value"] == full["value"] and len(out[-1]) < bench.LINE_LIMIT
    side = json.load(open(os.path.join(str(tmp_path), os.path.basename(line["detail"]))))
    assert side == full and "ceilings" in side["roofline"]  # everything the line dropped is in the sidecar


def test_from_detail_through_the_command_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--from-detail", FULL_N1], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    last = r.stdout.strip().splitlines()[-1]
    assert len(last) < 8000 and json.loads(last)["metric"].startswith("Msamples/sec")


# ---- profile-derived figures belong to the library that was profiled ----
def _stats():
    """DrRenderStats of three C2-like steps (the round-5 profile's own proportions)."""
    n = 3
    return {"camera_samples": n * 268960000, "film_samples": n * 268435456, "closest_rays": n * 1094000000, "any_rays": n * 734000000,
            "closest_nodes": n * 5.2e10, "any_nodes": n * 2.25e10, "closest_tris": n * 2.2e9, "any_tris": n * 1.1e9,
            "closest_launches": n * 7, "any_launches": n * 6, "closest_ms": n * 209.5, "any_ms": n * 96.8, "shade_ms": n * 96.1, "gen_ms": n * 9.2,
            "film_ms": n * 3.4, "trace_ms": n * 306.0, "total_ms": n * 416.0, "shade_items": n * 1.3e9, "shade_vertices": n * 1.0e9,
            "shade_cont": n * 0.8e9, "shade_mis": n * 0.25e9, "shade_shadow": n * 0.73e9}


def _profile(stamp):
    prof = json.load(open(os.path.join(ROOT, "profiles", "r05_c2_traffic.json")))
    prof["library_sources"] = stamp
    prof["pmc_renders"] = 2.0
    return prof


def test_profile_figures_only_for_the_library_that_was_profiled(tmp_path):
    hashes = {"dr_trace.hip": "a" * 64, "dr_kernels.hip": "b" * 64}
    pk = {"rays": 3 * 268960000, "nodes": 3 * 4.5e9, "tris": 3 * 3.0e8, "launches": 3, "ms": 3 * 17.0}
    st = _stats()
    # fresh: stamped with the running library's hashes
    roof, shade, gen, _, cam, _ = bench.roofline_objects(st, 3 * 0.416, 5600.0, 3, "k_trace<0>", "C2", 37.0, pk=pk, profile=("r05_c2_traffic.json", _profile(hashes), True))
    assert roof["stale_profile"] is None and roof["traffic"] > 1e11 and 0.3 < roof["traffic_over_algorithmic"] < 1.0
    assert 0.5 < roof["frac_physical_of_copy"] < 1.0 and 0.3 < roof["lane_utilisation"] < 0.7
    for obj in (roof, shade, gen):
        c = obj["ceilings"]
        assert c and c["binding"] in c["ceilings_ms"] and 0.0 < c["frac_of_binding_ceiling"] <= 1.0, obj   # every kernel is UNDER its ceilings
        assert all(v <= c["measured_ms"] * 1.0001 for v in c["ceilings_ms"].values())
    assert "valu_issue_ms" in shade["ceilings"]["ceilings_ms"] and "memory_lines_ms" in shade["ceilings"]["ceilings_ms"]
    assert "lines_beyond_l2_ms" in roof["ceilings"]["ceilings_ms"] and "l1_divergent_loads_ms" in roof["ceilings"]["ceilings_ms"]
    line = bench.compact_line({"value": 1.0, "config": {"workload": "C2: x"}, "roofline": roof, "roofline_shade": shade, "roofline_gen": gen, "roofline_camera": cam})
    assert line["roofline"]["frac_of_binding_ceiling"] == roof["ceilings"]["frac_of_binding_ceiling"] and line["roofline"]["stale_profile"] is None
    assert line["roofline_others"]["shade"]["frac_of_binding_ceiling"] == shade["ceilings"]["frac_of_binding_ceiling"]
    # stale: the same file, another library -- the algorithmic figures stay, everything read off the profile is null and the line says why
    roof2, shade2, gen2, _, cam2, _ = bench.roofline_objects(st, 3 * 0.416, 5600.0, 3, "k_trace<0>", "C2", 37.0, pk=pk, profile=("r05_c2_traffic.json", _profile({"dr_trace.hip": "c" * 64}), False))
    assert roof2["achieved"] == roof["achieved"] and roof2["frac"] == roof["frac"]
    assert roof2["traffic"] is None and "ceilings" not in roof2 and "frac_physical_of_copy" not in roof2 and "r05_c2_traffic.json" in roof2["stale_profile"]
    assert "ceilings" not in shade2 and "ceilings" not in gen2 and shade2["stale_profile"] and cam2["stale_profile"]
    line2 = bench.compact_line({"value": 1.0, "config": {"workload": "C2: x"}, "roofline": roof2, "roofline_shade": shade2})
    assert line2["roofline"]["traffic"] is None and line2["roofline"]["frac_physical_of_copy"] is None and line2["roofline"]["frac_of_binding_ceiling"] is None
    assert "another library" in line2["roofline"]["stale_profile"]
    # another traversal kernel than the profiled one: not this kernel's profile either
    roof3 = bench.roofline_objects(st, 3 * 0.416, 5600.0, 3, "k_trace3c", "C2", 37.0, pk=pk, profile=("r05_c2_traffic.json", _profile(hashes), True))[0]
    assert roof3["traffic"] is None and "another traversal kernel" in roof3["stale_profile"]
    # load_profile: the stamp decides
    d = tmp_path / "profiles"
    d.mkdir()
    json.dump(_profile(hashes), open(d / "r06_c2_traffic.json", "w"))
    assert bench.load_profile("C2", hashes, str(d))[2] is True
    assert bench.load_profile("C2", dict(hashes, **{"dr_trace.hip": "d" * 64}), str(d))[2] is False
    json.dump({k: v for k, v in _profile(hashes).items() if k != "library_sources"}, open(d / "r06_c2_traffic.json", "w"))
    assert bench.load_profile("C2", hashes, str(d))[2] is False       # rounds 1-5 recorded no hashes: never fresh


def test_the_committed_profiles_of_this_round_belong_to_the_committed_sources():
    """Once round-6 traffic files are committed they must carry the hashes of the kernel sources in the tree (a kernel change after the
    passes makes the driver's line print "stale_profile": retake the passes or do not commit the change)."""
    import hashlib
    files = [n for n in os.listdir(os.path.join(ROOT, "profiles")) if n.startswith("r06_") and n.endswith("_traffic.json")]
    for n in files:
        stamp = json.load(open(os.path.join(ROOT, "profiles", n))).get("library_sources")
        assert stamp, n
        for src, digest in stamp.items():
            have = hashlib.sha256(open(os.path.join(ROOT, "dartray_amd", "csrc", src), "rb").read()).hexdigest()
            assert have == digest, "%s was profiled with another %s" % (n, src)
