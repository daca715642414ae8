"""CPU: the bench's CPU legs and bookkeeping helpers (the GPU legs run on the MI355X box)."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_baseline_shape():
    cb = bench.cpu_baseline(0.4)
    assert set(cb) >= {"value", "unit", "cores", "kind", "sample"}
    assert cb["kind"] == "port" and cb["unit"] == "env-steps/s" and cb["cores"] >= 1 and cb["value"] > 1e5
    assert bench.python_port_rate() > 1e3


def test_constants_match_the_roofline_model():
    assert bench.ALGO_BYTES_PER_STEP == 16 + 1 + 16 + 4 + 1
    assert bench.HBM_PEAK_GBS == 8000.0


def test_committed_traffic_profile_is_consistent():
    t = bench.load_traffic()
    assert t and t["kernel"] == "step_kernel"
    algorithmic = bench.ALGO_BYTES_PER_STEP * (1 << 20)
    assert algorithmic < t["bytes_per_launch"] < 1.5 * algorithmic
    assert abs(t["crosscheck_dram_32b_bytes_per_launch"] - t["bytes_per_launch"]) < 0.02 * t["bytes_per_launch"]


def test_committed_bench_lines_carry_the_contract_fields():
    prof = os.path.join(ROOT, "profiles")
    lines = sorted(f for f in os.listdir(prof) if f.endswith("_bench.json"))
    assert lines
    d = json.loads(open(os.path.join(prof, lines[-1])).read())
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "u8" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert d["value"] > 1e8          # the north-star target
