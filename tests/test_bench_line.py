"""The LAST stdout line of bench.py is what the driver parses: it must stay short and carry the contract's keys.
(Round 4's line had grown to 22.6 KB; the driver kept a tail that began in the middle of it and recorded `parsed: null`.)
The recorded full records under tests/golden/ are bench.py's own earlier output (data, produced on a GPU box)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (argument parsing and the line builders import nothing heavy)

GOLDEN = os.path.join(ROOT, "tests", "golden")
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def _line(name):
    details = json.load(open(os.path.join(GOLDEN, name)))
    final = bench.compact_line(details)
    return details, final, json.dumps(final, separators=(",", ":"))


def test_final_line_of_the_default_run_is_short_and_complete():
    details, final, text = _line("bench_details_n1.json")
    assert len(json.dumps(details)) > 20000          # (the record that broke the driver's parser)
    assert len(text) < bench.FINAL_LINE_LIMIT
    assert "\n" not in text and "NaN" not in text and "Infinity" not in text
    back = json.loads(text)
    for key in CONTRACT:
        assert key in back, key
    assert back["value"] == pytest.approx(details["value"], rel=1e-6)
    assert back["ms_per_step"] == pytest.approx(details["ms_per_step"], rel=1e-6)
    assert back["metric"] == details["metric"] and back["unit"] == "knots/s" and back["dtype"] == "f64"
    assert set(back["config"]) >= {"workload", "horizon", "batch", "n", "m", "nnz", "jac_order", "ranks", "exchange"}
    assert "model" not in back["config"]
    # where the process ran (bench.py pins itself to the card's NUMA node): passed through, cut to a short string
    pinned = bench.compact_line(dict(details, config=dict(details["config"], numa="process pinned to the 128 allowed CPUs of NUMA node 1, the card's")))
    assert pinned["config"]["numa"].startswith("process pinned to the 128 allowed CPUs of NUMA node 1") and len(pinned["config"]["numa"]) <= 60
    roof = back["roofline"]
    assert set(roof) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "algorithmic_bytes_per_knot"}
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s"
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], rel=1e-3)
    assert roof["achieved"] == pytest.approx(details["roofline"]["achieved"], rel=1e-4)
    cb = back["cpu_baseline"]
    assert set(cb) >= {"value", "unit", "cores", "kind", "sample", "all_cores", "casadi", "gpu_over_cpu"}
    assert cb["kind"] == "port" and cb["cores"] == 1 and len(cb["gpu_over_cpu"]) <= 6
    assert set(cb["all_cores"]) == {"value", "cores", "nproc"}
    # the side measurements are rows of numbers
    for tag, row in back["throughput"].items():
        assert isinstance(row, list) and len(row) == 4, tag
        assert row[1] == pytest.approx(details["throughput"][tag].get("knots_per_s", details["throughput"][tag].get("poses_per_s")), rel=1e-3)
    assert len(back["host_visible"]) <= 8 and all(isinstance(v, float) for v in back["host_visible"].values())
    assert back["host_visible"]["all_us"] == pytest.approx(1e3 * details["host_visible"]["all"]["ms_per_call"], abs=0.01)


def test_final_line_of_a_multi_rank_run_is_short_and_complete():
    details, final, text = _line("bench_details_n4_rehearsal.json")
    assert len(text) < bench.FINAL_LINE_LIMIT
    back = json.loads(text)
    for key in CONTRACT:
        assert key in back, key
    assert back["n_gpus"] == 4 and back["config"]["ranks"] == 4 and back["config"]["exchange"] == details["config"]["exchange"]
    assert "all_gather" in back["exchanges"] and back["exchanges"]["all_gather"][2] == details["all_gather"]["bytes_sent_per_rank_per_step"]
    assert "cpu_baseline" not in back      # (rank 0 at N = 1 only)
    assert "REHEARSAL" in back["config"]["backend"]
    for key in ("config4_strong", "config5"):
        assert isinstance(back[key], dict) and back[key]


def test_an_overgrown_record_still_yields_a_short_line():
    details = json.load(open(os.path.join(GOLDEN, "bench_details_n1.json")))
    for i in range(400):     # a future builder adds legs: the side rows are dropped before the contract's keys are
        details["throughput"]["extra_leg_with_a_long_name_%03d" % i] = dict(details["throughput"]["periodic_N100_B64"])
    final = bench.compact_line(details)
    text = json.dumps(final, separators=(",", ":"))
    assert len(text) < bench.FINAL_LINE_LIMIT
    assert "roofline" in final and "cpu_baseline" in final and "value" in final
