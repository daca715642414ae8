#!/usr/bin/env python
"""What strict actions cost a launch at 2^20 boards (tool, GPU box): the reward + terminated rollout of bench.py with
g2048_set_strict_actions off (the STD specialisation of the step kernel) and on (the general kernel + one compare per lane)."""
import sys

sys.path.insert(0, ".")
import torch

from gym2048_amd.batched import Batched2048

n, k = 1 << 20, 200
e = Batched2048(n, seed=42, last_records=False)
e.reset()
e.rollout_random(64)
acts = e.random_actions(k)
rew = torch.zeros((k, n), dtype=torch.float32, device=e.device)
term = torch.zeros((k, n), dtype=torch.uint8, device=e.device)
for rep in range(3):
    for strict in (False, True):
        e.set_strict_actions(strict)
        plan = e.prepare_rollout(acts, reward=rew, terminated=term)
        plan.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        plan.run()
        e1.record()
        torch.cuda.synchronize()
        print(f"strict actions {'on ' if strict else 'off'}: {e0.elapsed_time(e1) * 1e3 / k:.2f} us per launch at 2^20 boards")
