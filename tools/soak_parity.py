#!/usr/bin/env python
"""One-off soak: GPU vs C oracle at full batch with a biased policy that builds large tiles.
    python tools/soak_parity.py [log2_boards] [steps]"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

import __graft_entry__ as ge

ge.build()
from gym2048_amd.batched import Batched2048
from oracle import OracleBatch

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
NOISE = float(sys.argv[3]) if len(sys.argv) > 3 else 0.02
n = 1 << lg
eng = Batched2048(n, seed=2024)
ora = OracleBatch(n, 2024, threads=0)
eng.set_illegal_move_reward(-1.0)
ora.illegal_move_reward = -1.0
eng.reset()
ora.reset()
assert np.array_equal(eng.get_boards().reshape(n, 16), ora.boards)
rng = np.random.default_rng(1)
t0 = time.time()
for s in range(steps):
    # legality-aware corner policy (first legal of left, down, right, up -- found with trial moves on
    # the device), 2 % uniformly random: long episodes, big tiles, full-board endings
    choice = torch.full((n,), 3, dtype=torch.uint8, device=eng.device)
    found = torch.zeros(n, dtype=torch.bool, device=eng.device)
    for d in (3, 2, 1, 0):
        _, legal = eng.move(torch.full((n,), d, dtype=torch.uint8, device=eng.device), trial=True)
        take = legal.bool() & ~found
        choice = torch.where(take, torch.full_like(choice, d), choice)
        found |= take
    noise = torch.as_tensor(rng.random(n) < NOISE).to(eng.device)
    rand = torch.as_tensor(rng.integers(0, 4, n).astype(np.uint8)).to(eng.device)
    acts_t = torch.where(noise, rand, choice)
    acts = acts_t.cpu().numpy()
    eng.step(acts_t)
    ora.step(acts)
    if s % 25 == 24 or s == steps - 1:
        assert np.array_equal(eng.get_boards().reshape(n, 16), ora.boards), s
        assert np.array_equal(eng.get_scores(), ora.score), s
        assert np.array_equal(eng.reward.cpu().numpy(), ora.reward), s
        assert np.array_equal(eng.terminated.cpu().numpy(), ora.terminated), s
        assert np.array_equal(eng.highest.cpu().numpy(), ora.highest), s
st = eng.episode_stats()
assert st["episodes"] == int(ora.ep_count.sum())
print(f"soak ok: 2^{lg} boards x {steps} steps bit-exact vs oracle in {time.time() - t0:.0f} s; episodes {st['episodes']}, "
      f"max tile 2^{st['max_exp']}, max score {st['max_score']}, illegal ends {st['illegal_ends']}")
