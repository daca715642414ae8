#!/usr/bin/env python
"""numpy-RNG mode: env-steps/s and (under rocprofv3 --kernel-trace --stats) the kernels behind it (tool, GPU box)."""
import sys
import time

sys.path.insert(0, ".")
import torch

from gym2048_amd.batched import Batched2048

n, k = 1 << 20, 64
e = Batched2048(n, seed=42, rng="numpy")
e.reset()
acts = e.random_actions(k)
rew = torch.zeros((k, n), dtype=torch.float32, device=e.device)
term = torch.zeros((k, n), dtype=torch.uint8, device=e.device)
for _ in range(2):
    e.rollout(acts, reward=rew, terminated=term)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4):
    e.rollout(acts, reward=rew, terminated=term)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"numpy-RNG mode, 2^20 boards: {4 * k * n / dt:.3e} env-steps/s, {dt / (4 * k) * 1e6:.1f} us per step")
