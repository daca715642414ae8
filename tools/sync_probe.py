#!/usr/bin/env python
"""Where do the microseconds between the K-launch train and the host's clock go (tool, GPU box)?
K = 20 step launches at 2^20 boards, timed three ways: HIP events around the train, wall clock to
torch.cuda.synchronize(), wall clock to a host spin on event.query() followed by synchronize()."""
import sys
import time

sys.path.insert(0, ".")
import torch

import __graft_entry__ as ge

ge.build_hip()
from gym2048_amd.batched import Batched2048

B, K = 1 << 20, int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
eng = Batched2048(B, device=0, seed=42)
eng.reset()
eng.rollout_random(64)
actions = eng.random_actions(K)
reward = torch.zeros((K, B), dtype=torch.float32, device=dev)
term = torch.zeros((K, B), dtype=torch.uint8, device=dev)
plan = eng.prepare_rollout(actions, reward=reward, terminated=term)
for _ in range(20):
    plan.run()
torch.cuda.synchronize()


def once(mode):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ev0.record()
    t0 = time.perf_counter()
    plan.run()
    t_enq = time.perf_counter()
    ev1.record()
    if mode == "spin":
        while not ev1.query():
            pass
    elif mode == "stream":
        torch.cuda.current_stream().synchronize()
    elif mode == "event":
        ev1.synchronize()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    return (t_enq - t0) * 1e6, (t1 - t0) * 1e6, ev0.elapsed_time(ev1) * 1e3


for mode in ("sync", "spin", "stream", "event", "sync", "stream", "event"):
    rows = sorted(once(mode) for _ in range(30))
    med = rows[len(rows) // 2]
    enq = sorted(r[0] for r in rows)[15]
    wall = sorted(r[1] for r in rows)[15]
    evt = sorted(r[2] for r in rows)[15]
    print(f"{mode}: K={K} enqueue {enq:7.1f} us, wall to sync {wall:7.1f} us ({wall / K:.2f}/step), events {evt:7.1f} us ({evt / K:.2f}/step)")
