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


def _bench(extra_env, *flags):
    env = dict(os.environ, **extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=900,
                       cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_distributed_branch_runs_on_one_rank():
    """The code an N > 1 run takes -- init_process_group("nccl"), the single-call C-ABI sharded encode (trpx_encode_sharded: the rank's encode + the size gather on its own RCCL communicator
    and side stream, trpx_decode_sharded behind the timed region), the barrier + MAX-over-ranks timing, the global-offset assertions behind the timed region -- rehearsed with one
    rank in a fresh process (TRPX_BENCH_FORCE_DIST=1), next to the plain run of the same workload: the gather must be the
    C-ABI one and may not cost the step more than 10 % (frames are independent, Terse.hpp:502-505: nothing but the sizes
    is exchanged)."""
    flags = ("--steps", "20", "--warmup", "3", "--headline-only", "--no-cpu-baseline")
    plain = _bench({}, *flags)
    forced = _bench({"TRPX_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533"}, *flags)
    assert plain["config"]["size_gather"] is None
    assert "trpx_encode_sharded" in forced["config"]["size_gather"], forced["config"]
    assert forced["n_gpus"] == 1 and forced["scaling"] == "weak"
    assert "byte-identical" in forced["oracle_check"]
    assert forced["value"] > 0.9 * plain["value"], (forced["value"], plain["value"])


@pytest.mark.gpu
def test_bench_reports_the_workloads_the_headline_hides():
    """poisson3 / mid-size / odd-size legs with their index-free and indexed decode times, and the box's measured ceilings."""
    b = _bench({}, "--steps", "3", "--warmup", "1", "--frames", "200", "--no-cpu-baseline")
    for k in ("noisy_u16", "poisson3_u16", "midsize_u16", "midsize_poisson3_u16", "oddsize_u16"):
        leg = b[k]
        assert leg.get("roundtrip_exact") is True, (k, leg)
        for kk in ("encode_ms", "decode_ms", "decode_with_index_ms", "decode_frac_of_hbm_peak", "decode_with_index_frac_of_hbm_peak", "kernel_ms"):
            assert kk in leg, (k, kk)
        assert 0 < leg["decode_frac_of_hbm_peak"] < 1
    pm = b["peak_measured"]
    assert 500 < pm["write_misaligned_GBps"] <= 1.05 * pm["write_GBps"], pm          # (the decoder's store shape on frames that start inside a line)
    assert min(pm["read_GBps"], pm["write_GBps"], pm["copy_GBps"]) > 1000, pm      # (200 frames fit the Infinity Cache: no upper bound here)
    rf = b["roofline"]
    assert rf["peak"] == 8000.0 and rf["peak_measured"] > 0 and abs(rf["frac_of_measured"] - rf["achieved"] / rf["peak_measured"]) < 1e-9
