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
