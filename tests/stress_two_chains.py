#!/usr/bin/env python
"""Stress of the two-chain rollout's host machinery (test infrastructure): random rollout lengths, random pauses that
cross the side launcher's spin window (awake / about to sleep / asleep / nudged), several engines of one device taking
turns on the shared side chain from several threads, each on its own NON-DEFAULT stream, engines created and destroyed
along the way -- every rollout compared with a one-chain twin engine.  A lost wake-up or a ticket that never comes
shows as a hang (run it under `timeout`), anything else as a mismatch.  (It found that hipMemset only enqueues its fill:
an engine used on a non-blocking stream right after its creation could start before its state was cleared, about once
in 10^5 rollouts -- g2048_create now waits for the clear.)
    python tests/stress_two_chains.py [seconds=120] [seed=0] [threads=3]"""
import os
import sys
import threading
import time

if __name__ == "__main__":
    os.environ.setdefault("G2048_SIDE_SPIN_US", "300")  # a short spin window: the sleep / wake transitions come up often

sys.path.insert(0, ".")
import numpy as np
import torch

import __graft_entry__ as ge

ge.build()
from gym2048_amd.batched import Batched2048


def run(seconds=120.0, seed=0, n_threads=3):
    deadline = time.time() + seconds
    counts = {"rollouts": 0, "two": 0, "one": 0, "engines": 0}
    lock = threading.Lock()
    failures = []

    def worker(wid):
        try:
            rs = np.random.default_rng(seed * 100 + wid)
            stream = torch.cuda.Stream()
            while time.time() < deadline and not failures:
                n = int(rs.choice([512, 768, 1000, 4096, 65536, 1 << 18, 1 << 20], p=[.2, .15, .15, .2, .15, .1, .05]))
                s = int(rs.integers(0, 1 << 30))
                with torch.cuda.stream(stream):
                    two, one = Batched2048(n, seed=s, chains=2), Batched2048(n, seed=s, chains=1)
                    two.reset()
                    one.reset()
                    with lock:
                        counts["engines"] += 1
                    for _ in range(int(rs.integers(3, 40))):
                        k = int(rs.integers(1, 90))
                        with lock:   # the knob is process-wide and g2048_set_chains is what reads it (once, not per rollout)
                            if rs.random() < 0.7:
                                os.environ["G2048_TWO_CHAIN_MIN_STEPS"] = "2"
                            else:
                                os.environ.pop("G2048_TWO_CHAIN_MIN_STEPS", None)
                            two.set_chains(2)
                        r2 = torch.zeros((k, n), dtype=torch.float32, device=two.device)
                        r1 = torch.zeros((k, n), dtype=torch.float32, device=one.device)
                        d2 = torch.zeros((k, n), dtype=torch.uint8, device=two.device)
                        d1 = torch.zeros((k, n), dtype=torch.uint8, device=one.device)
                        extra2, extra1 = {}, {}
                        if n <= 65536 and rs.random() < 0.3:      # sometimes every optional output rides along
                            for ex in (extra2, extra1):
                                ex["illegal"] = torch.zeros((k, n), dtype=torch.uint8, device=two.device)
                                ex["highest"] = torch.zeros((k, n), dtype=torch.uint8, device=two.device)
                                ex["terminal_boards"] = torch.zeros((k, n, 16), dtype=torch.uint8, device=two.device)
                                ex["obs"] = torch.zeros((k, n, 16, 4, 4), dtype=torch.uint8, device=two.device) if n <= 4096 else None
                        two.rollout(k, reward=r2, terminated=d2, **{a: b for a, b in extra2.items() if b is not None})
                        used = two.chains_used
                        one.rollout(k, reward=r1, terminated=d1, **{a: b for a, b in extra1.items() if b is not None})
                        same = all(torch.equal(extra2[a], extra1[a]) for a in extra2 if extra2[a] is not None)
                        if not (same and torch.equal(r2, r1) and torch.equal(d2, d1)):
                            failures.append(f"worker {wid}: n={n} seed={s} k={k}: outputs differ (chains_used={used})")
                            break
                        with lock:
                            counts["rollouts"] += 1
                            counts["two" if used == 2 else "one"] += 1
                        pause = rs.choice([0.0, 0.0, 1e-4, 2.5e-4, 3.5e-4, 1e-3, 5e-3])
                        if pause:
                            time.sleep(float(pause))
                    if not torch.equal(two.records(), one.records()):
                        failures.append(f"worker {wid}: n={n} seed={s}: final records differ")
                    if two.episode_stats() != one.episode_stats():
                        failures.append(f"worker {wid}: n={n} seed={s}: statistics differ")
                    two.close()
                    one.close()
        except Exception as exc:  # a worker that dies must fail the run, not shorten it
            failures.append(f"worker {wid}: {type(exc).__name__}: {exc}")

    threads = [threading.Thread(target=worker, args=(w,)) for w in range(n_threads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    os.environ.pop("G2048_TWO_CHAIN_MIN_STEPS", None)
    if failures:
        return "stress FAILED:\n" + "\n".join(failures)
    return (f"stress ok: {counts['rollouts']} rollouts ({counts['two']} as two chains, {counts['one']} as one) on "
            f"{counts['engines']} engine pairs from {n_threads} threads in {seconds:.0f} s, every output equal to the "
            f"one-chain twin")


if __name__ == "__main__":
    line = run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0,
               int(sys.argv[3]) if len(sys.argv) > 3 else 3)
    print(line)
    sys.exit(0 if line.startswith("stress ok") else 1)
