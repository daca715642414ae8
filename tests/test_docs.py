"""The documents cite their evidence by file name: every `rNN_*` name DESIGN.md, INTEGRATION.md, README.md and
profiles/README.md mention must exist under profiles/ (or profiles/archive/, where the intermediate states live), and
DESIGN.md stays within the size it was cut to."""
import fnmatch
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _profile_names():
    names = set(os.listdir(os.path.join(ROOT, "profiles")))
    names |= set(os.listdir(os.path.join(ROOT, "profiles", "archive")))
    return names


def test_cited_profile_files_exist():
    have = _profile_names()
    missing = []
    for doc in ("DESIGN.md", "INTEGRATION.md", "README.md", os.path.join("profiles", "README.md")):
        text = open(os.path.join(ROOT, doc), encoding="utf-8").read()
        for m in re.finditer(r"`(?:profiles/)?((?:r0[1-9]|traffic)_[A-Za-z0-9_*.…]+)`", text):
            name = m.group(1).rstrip(".")
            if "…" in name:
                continue
            pattern = name if ("*" in name or "." in name) else name + "*"
            if not any(fnmatch.fnmatch(f, pattern) for f in have):
                missing.append((doc, name))
    assert not missing, missing


def test_design_md_is_the_short_current_state_document():
    size = os.path.getsize(os.path.join(ROOT, "DESIGN.md"))
    assert size <= 35 * 1024, f"DESIGN.md is {size} bytes: move history to DESIGN_HISTORY.md"


def test_per_size_profiles_exist_and_reproduce_the_three_roofline_fractions():
    """The headline fraction (2^20 boards, Infinity-Cache resident), the HBM-streaming one (2^24) and the latency-chain one
    (65 536 boards, BASELINE configs[1]) each have a kernel trace of their own under profiles/ (tools/gpu_profile.sh takes
    one per size since round 6): a reader can recompute all three from tracked files."""
    import csv
    import json
    prof = os.path.join(ROOT, "profiles")
    tag = json.load(open(os.path.join(prof, "traffic_latest.json")))["profile"]

    def avg_us(size, kernel):
        path = os.path.join(prof, f"{tag}_kernel_stats_{size}.csv")
        assert os.path.exists(path), path
        for row in csv.DictReader(open(path)):
            if kernel in row["Name"]:
                return float(row["AverageNs"]) / 1e3, int(row["Calls"])
        raise AssertionError(f"{kernel} not in {path}")

    frac = lambda boards, us: 38 * boards / (us * 1e-6) / 8e12   # noqa: E731  (38 B per env-step, 8 TB/s)
    us20, calls20 = avg_us("2p20", "step_kernel<1, true, true, false>")
    us24, _ = avg_us("2p24", "step_kernel<1, true, true, false>")
    us16, _ = avg_us("2p16", "step_graph_kernel<1, true>")
    assert calls20 > 10000 and 0.45 < frac(1 << 20, us20) < 0.65
    assert 0.6 < frac(1 << 24, us24) < 0.8                      # every launch streams HBM
    assert 0.05 < frac(1 << 16, us16) < 0.2                     # 256 workgroups: a latency chain (serialised by the profiler)
    for size in ("2p16", "2p24"):
        assert os.path.exists(os.path.join(prof, f"{tag}_summary_{size}.txt"))   # FETCH_SIZE / WRITE_SIZE passes of that size
