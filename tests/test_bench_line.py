"""bench.py's LAST stdout line must survive the driver's 8 KB stdout tail (round 4's 20 KB line did not).
CPU-only: builds the compact line from a recorded full result and from a worst-case one."""
import io
import json
import os
import sys
from contextlib import redirect_stderr, redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _bench():
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        import bench
    finally:
        sys.argv = argv
    return bench


def _recorded():
    with open(os.path.join(ROOT, "profiles", "r04_bench_default.json")) as f:
        return json.load(f)


def _worst_case():
    """every string as long as bench.py could make it, 20 secondary blocks, every optional block present"""
    out = _recorded()
    long_text = "x" * 2000
    out["config"]["plan"] = long_text
    out["config"]["parallelism"] = long_text
    out["config"]["exchange"] = long_text
    out["cpu_baseline"]["sample"] = long_text
    out["cpu_baseline_mt"]["sample"] = long_text
    out["roofline"]["traffic_source"] = long_text
    out["strong_scaling"] = {"scaling": "strong", "value": 1.23456789e11, "unit": "rows/s", "steps": 5, "warmup": 2,
                             "ms_per_step": 4.56789, "rows_total": 600037902, "rows_per_gpu": 75004737}
    out["result_check"] = {"what": long_text, "ok": True}
    blocks = list(out["secondary"].items())
    for i in range(20):
        key, blk = blocks[i % len(blocks)]
        out["secondary"]["%s_%s_%d" % (key, "y" * 40, i)] = blk
    out["secondary"]["broken"] = {"error": "RuntimeError: " + long_text}
    return out


def test_compact_line_of_a_recorded_run_keeps_the_contract():
    bench = _bench()
    text = bench.compact_line(_recorded())
    assert len(text) < bench.LINE_LIMIT
    line = json.loads(text)
    for k in CONTRACT_KEYS:
        assert k in line, k
    full = _recorded()
    assert abs(line["value"] - full["value"]) / full["value"] < 1e-6
    assert line["roofline"]["kernel"] == full["roofline"]["kernel"]
    assert abs(line["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-3
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["unit"] == "GB/s"
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] == 1
    assert set(line["secondary"]) == set(full["secondary"])
    for blk in line["secondary"].values():
        assert set(blk) >= {"value", "ms_per_step", "kernel", "frac", "traffic_ratio", "cpu_value"}


def test_worst_case_line_fits_the_drivers_tail():
    bench = _bench()
    out = _worst_case()
    assert len(json.dumps(out)) > 40000
    text = bench.compact_line(out)
    assert len(text) < 6000 and len(text) < bench.LINE_LIMIT
    line = json.loads(text)
    for k in CONTRACT_KEYS:
        assert k in line, k


def test_last_8000_bytes_of_stdout_parse(tmp_path):
    bench = _bench()
    out = _worst_case()
    so, se = io.StringIO(), io.StringIO()
    so.write("RCCL version banner and other noise\n" * 300)
    with redirect_stdout(so), redirect_stderr(se):
        bench.emit(out, str(tmp_path / "detail.json"))
    tail = so.getvalue()[-8000:]
    last = tail.rstrip("\n").split("\n")[-1]
    line = json.loads(last)
    assert line["metric"].startswith("rows/s")
    assert line["detail"] == str(tmp_path / "detail.json")
    # the full object went to the file and to stderr, not to stdout
    with open(tmp_path / "detail.json") as f:
        assert json.load(f)["roofline"]["traffic_source"] == "x" * 2000
    assert se.getvalue().startswith("BENCH_DETAIL ")
    assert "BENCH_DETAIL" not in so.getvalue()
