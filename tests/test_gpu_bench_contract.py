"""The driver's contract for bench.py: one JSON line with the agreed keys, `roofline` and (N = 1) `cpu_baseline`."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--frames", "256"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py must print exactly one line"
    b = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in b, k
    assert b["unit"] == "frames/s" and b["n_gpus"] == 1 and b["steps"] == 3 and b["warmup"] == 1
    assert b["higher_is_better"] is True and b["scaling"] == "weak" and b["vs_baseline"] is None and b["dtype"] == "u16"
    assert "workload" in b["config"] and "model" not in b["config"]
    assert abs(b["value"] - 256 / (b["ms_per_step"] * 1e-3)) / b["value"] < 0.02
    rf = b["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0 < rf["frac"] < 1
    # both sides of the step carry a roofline block (the dominant one under `roofline`), and the legs the headline
    # data hides are reported: index-free decode of the 4096x4096 frames, the header-dense noisy stack
    other = [k for k in ("roofline_encode", "roofline_decode") if k in b]
    assert len(other) == 1 and b[other[0]]["avg_launch_ms"] <= rf["avg_launch_ms"]
    assert "decode_fps" in b["config3_4096x4096_int32"] and "roofline_encode" in b["config3_4096x4096_int32"], b["config3_4096x4096_int32"]
    assert b["noisy_u16"].get("roundtrip_exact") is True and b["noisy_u16"]["decode_fps"] > 0, b["noisy_u16"]
    assert "byte-identical" in b["oracle_check"]
    cb = b["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0
