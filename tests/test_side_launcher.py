"""The launch thread of a two-chain rollout (gym-2048_amd/csrc/g2048_side_launcher.h) under ThreadSanitizer.

The header is HIP-free, so the protocol g2048_api.hip runs on it -- post / wait under the side chain's lock, nudges from
threads that do not hold it, the thread's sleep / wake transitions, stop -- can be executed here, on the CPU, with every
memory access watched: tests/side_launcher/side_launcher_test.cpp, 10^5 rounds from each of two threads."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "side_launcher", "side_launcher_test.cpp")
HEADER = os.path.join(os.path.dirname(HERE), "gym-2048_amd", "csrc", "g2048_side_launcher.h")


def build(tmp_path, sanitizer):
    exe = str(tmp_path / f"side_launcher_{sanitizer}")
    cmd = ["g++", "-O1", "-g", "-std=c++17", f"-fsanitize={sanitizer}", "-o", exe, SRC, "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and ("cannot find" in r.stderr or "unrecognized" in r.stderr):
        pytest.skip(f"this toolchain has no -fsanitize={sanitizer} runtime: {r.stderr.strip().splitlines()[-1]}")
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_side_launcher_is_clean_under_thread_sanitizer(tmp_path):
    exe = build(tmp_path, "thread")
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    # spin window of 20 us: the posters' random pauses cross it thousands of times (sleep -> wake -> job)
    r = subprocess.run([exe, "100000", "2", "20"], capture_output=True, text=True, timeout=600, env=env)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0, (r.stdout, r.stderr[-2000:])
    assert r.stdout.startswith("ok: 200000 jobs from 2 threads"), r.stdout
    slept = int(r.stdout.split("slept")[1].split()[0])
    assert slept > 100, "the run never crossed the launcher's sleep / wake transition"
    # three posters and the product's default window
    r = subprocess.run([exe, "20000", "3", "200"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr, (r.stdout, r.stderr[-2000:])


def test_the_header_is_hip_free_and_the_product_uses_it():
    text = open(HEADER).read()
    assert "#include <hip" not in text and "hipStream" not in text and "hipError" not in text
    api = open(os.path.join(os.path.dirname(HEADER), "g2048_api.hip")).read()
    assert '#include "g2048_side_launcher.h"' in api and "struct SideLauncher" not in api
