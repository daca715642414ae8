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


def test_cpu_baseline_team_is_calibrated_and_capped_at_the_physical_cores():
    """Round 5's CPU figure halved between two driver runs because the OpenMP team was picked from a two-step sample and
    could be an oversubscribed one: the team is now calibrated on >= 0.3 s per candidate, capped at the physical cores, and
    the line says what was tried and on which CPU."""
    info = bench.host_cpu_info()
    assert 1 <= info["physical_cores"] <= info["visible_cpus"]
    cb = bench.cpu_baseline(0.3)
    tried = {int(k): v for k, v in cb["threads_tried"].items()}
    assert 1 in tried and max(tried) <= info["physical_cores"] and all(v > 1e5 for v in tried.values())
    assert cb["cores"] == max(tried, key=tried.get) and cb["physical_cores"] == info["physical_cores"]
    assert abs(cb["per_core_value"] - cb["value"] / cb["cores"]) < 1e-6 * cb["value"] and "cpu_model" in cb


def test_constants_match_the_roofline_model():
    assert bench.ALGO_BYTES_PER_STEP == 16 + 1 + 16 + 4 + 1
    assert bench.HBM_PEAK_GBS == 8000.0


def test_committed_traffic_profile_is_consistent():
    t = bench.load_traffic()
    assert t and t["kernel"] == "step_kernel" and t["profile"] and len(t["csrc_sha16"]) == 16
    algorithmic = bench.ALGO_BYTES_PER_STEP * (1 << 20)
    assert algorithmic < t["bytes_per_launch"] < 1.15 * algorithmic          # round 2: x1.08 (round 1: x1.28)
    assert abs(t["crosscheck_dram_32b_bytes_per_launch"] - t["bytes_per_launch"]) < 0.02 * t["bytes_per_launch"]


def test_traffic_is_reported_only_for_the_profiled_sources(monkeypatch):
    """roofline.traffic comes from a committed profile: it is printed for the profiled batch size while the
    kernel sources hash to what was profiled, and is null otherwise (a stale profile never goes unnoticed)."""
    t = bench.load_traffic()
    monkeypatch.setattr(bench, "csrc_hash", lambda: t["csrc_sha16"])
    val, prov = bench.traffic_for_current_sources(1 << 20)
    assert val == t["bytes_per_launch"] and prov["profile"] == t["profile"]
    assert bench.traffic_for_current_sources(1 << 16)[0] is None
    monkeypatch.setattr(bench, "csrc_hash", lambda: "0" * 16)
    val, prov = bench.traffic_for_current_sources(1 << 20)
    assert val is None and prov["profiled_csrc"] == t["csrc_sha16"] and prov["current_csrc"] == "0" * 16


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


def test_bench_script_runs_its_main():
    """bench.py must execute main() when run as a script (the driver calls `python bench.py ...`): --help prints the
    contract's flags, and without a GPU the run ends with the "needs a ROCm GPU" message instead of silence."""
    import subprocess
    import sys
    bench_py = os.path.join(ROOT, "bench.py")
    out = subprocess.run([sys.executable, bench_py, "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "--gpus" in out.stdout and "--steps" in out.stdout and "--warmup" in out.stdout
    import torch
    if not torch.cuda.is_available():
        out = subprocess.run([sys.executable, bench_py, "--steps", "2", "--warmup", "1"], capture_output=True, text=True,
                             timeout=300)
        assert out.returncode != 0 and "needs a ROCm GPU" in (out.stderr + out.stdout)


# ---------------------------------------------------------------- the self-launcher (`python bench.py --gpus N`, N > 1)
_RANK_SCRIPT = r"""
import json, os, sys, time
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["LOCAL_WORLD_SIZE"] == os.environ["WORLD_SIZE"]
mode = sys.argv[1]
print(f"banner from rank {rank}")
if mode == "ok":
    if rank == 0:
        print(json.dumps({"n_gpus": world, "args": sys.argv[2:]}))
        print("trailing noise from rank 0")          # e.g. a library banner flushed at exit
elif mode == "fail1":
    if rank == 1:
        sys.exit(7)
    time.sleep(60)                                   # "stuck in a collective": the launcher must stop it
elif mode == "nojson":
    pass
"""


def _run_launcher(tmp_path, mode, n=2, grace=1.0):
    import subprocess
    import sys
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    driver = (f"import sys; sys.path.insert(0, {ROOT!r}); import bench; "
              f"sys.exit(bench.launch_ranks({n}, [sys.executable, {str(script)!r}, {mode!r}, '--steps', '20'], grace={grace}))")
    return subprocess.run([sys.executable, "-c", driver], capture_output=True, text=True, timeout=120)


def test_launcher_relays_rank0_json_as_the_last_stdout_line(tmp_path):
    out = _run_launcher(tmp_path, "ok", n=3)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    d = json.loads(lines[-1])                                    # the bench line is LAST, whatever rank 0 printed after it
    assert d == {"n_gpus": 3, "args": ["--steps", "20"]}
    assert "banner from rank 0" in out.stdout and "trailing noise from rank 0" in lines[:-1]
    assert "banner from rank 1" in out.stderr and "banner from rank 2" in out.stderr      # other ranks: stderr
    assert "banner from rank 1" not in out.stdout


def test_launcher_propagates_the_worst_exit_code_and_stops_stuck_ranks(tmp_path):
    import time
    t0 = time.time()
    out = _run_launcher(tmp_path, "fail1", n=2, grace=1.0)
    assert out.returncode == 143 and time.time() - t0 < 30       # rank 1: 7; rank 0 terminated (SIGTERM = 128 + 15)
    assert "rank 1 exited with 7" in out.stderr and "rank 0 exited with -15" in out.stderr
    out = _run_launcher(tmp_path, "nojson")
    assert out.returncode == 1 and "printed no JSON line" in out.stderr


def test_bench_gpus_n_without_enough_gpus_fails_with_one_clear_line():
    """`python bench.py --gpus 8` on a box with fewer GPUs: no launcher needed, no hang, one line, rc != 0."""
    import subprocess
    import sys
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "G2048_BENCH_SAME_DEVICE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 7), "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 2 and out.stdout == ""
    assert f"--gpus {have + 7} needs {have + 7} visible GPUs and this box has {have}" in out.stderr
