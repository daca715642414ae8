#!/usr/bin/env python
"""Soak script (test infrastructure; lives under tests/ because it drives the oracle): the HIP engine against the C oracle for thousands of steps of a legality-aware corner
policy with a little noise -- long games, tiles of 2^11 and beyond, deficits that carry across many bits.
    python tests/soak_parity.py [log2_boards=18] [steps=3000] [noise=0.01] [rng=philox|numpy]
rng=numpy: the same in numpy-RNG mode (the reference's own PCG64 per board; the record keeps the score by its deficit there
too since round 6), generator states compared as well, and every 100 steps a 24-step FUSED rollout of random actions
(g2048_rollout_fused in that mode: lanes drifting in time) checked step by step."""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

import __graft_entry__ as ge

ge.build()
from gym2048_amd.batched import Batched2048
from oracle import OracleBatch

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 18
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
noise = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
numpy_mode = len(sys.argv) > 4 and sys.argv[4] == "numpy"
n = 1 << lg
eng = Batched2048(n, seed=2025, illegal_move_reward=-1.0, rng="numpy" if numpy_mode else "philox")
ora = OracleBatch(n, 2025, threads=0)
ora.illegal_move_reward = -1.0
if numpy_mode:
    ora.seed_numpy(2025)
    ora.step, ora.reset = ora.step_numpy, ora.reset_numpy
eng.reset()
ora.reset()
obs_u8 = torch.zeros((n, 16, 4, 4), dtype=torch.uint8, device=eng.device)     # the step writes its observation itself,
obs_f16 = torch.zeros((n, 16, 4, 4), dtype=torch.float16, device=eng.device)  # alternating between two dtypes
gen = torch.Generator(device=eng.device)
gen.manual_seed(1)
t0 = time.time()
for s in range(steps):
    mask = eng.legal_actions()                                   # bit d = move d is legal
    choice = torch.full((n,), 3, dtype=torch.uint8, device=eng.device)
    for d in (0, 1, 2, 3):                                       # highest legal direction wins: 3, 2, 1, 0
        choice = torch.where((mask >> d) & 1 == 1, torch.full_like(choice, d), choice)
    rnd = torch.rand(n, device=eng.device, generator=gen) < noise
    rand_a = torch.randint(0, 4, (n,), device=eng.device, generator=gen, dtype=torch.uint8)
    acts = torch.where(rnd, rand_a, choice)
    obs = obs_u8 if s % 2 == 0 else obs_f16
    eng.step(acts, obs=obs)
    ora.step(acts.cpu().numpy())
    if numpy_mode and s % 100 == 50:                             # a fused chunk in the middle of long games
        kf = 24
        fa = torch.randint(0, 4, (kf, n), device=eng.device, generator=gen, dtype=torch.uint8)
        fr = torch.zeros((kf, n), dtype=torch.float32, device=eng.device)
        ft = torch.zeros((kf, n), dtype=torch.uint8, device=eng.device)
        eng.rollout(fa, reward=fr, terminated=ft, fused=True)
        fa_h, fr_h, ft_h = fa.cpu().numpy(), fr.cpu().numpy(), ft.cpu().numpy()
        for j in range(kf):
            ora.step(fa_h[j])
            assert np.array_equal(fr_h[j], ora.reward) and np.array_equal(ft_h[j], ora.terminated), (s, j)
        eng.step(acts, obs=obs)                                  # (so that eng.reward / obs below belong to a per-step call)
        ora.step(acts.cpu().numpy())
    assert np.array_equal(eng.reward.cpu().numpy(), ora.reward), s
    assert np.array_equal(eng.terminated.cpu().numpy(), ora.terminated), s
    if s % 100 == 99 or s == steps - 1:
        assert np.array_equal(eng.get_boards().reshape(n, 16), ora.boards), s
        assert np.array_equal(eng.get_scores(), ora.score), s
        if numpy_mode:
            assert np.array_equal(eng.get_numpy_rng().T, ora.rng), s
        assert np.array_equal(eng.get_last_scores(), ora.last_score), s
        assert np.array_equal(eng.highest.cpu().numpy(), ora.highest), s
        assert eng.episode_stats()["return_sum"] == ora.finished_return_sum == ora.return_sum, s     # exact, all episodes
        assert np.array_equal(obs.cpu().numpy(), ora.onehot().astype(obs.cpu().numpy().dtype)), s   # fused observation
st = eng.episode_stats()
assert st["episodes"] == int(ora.ep_count.sum())
print(f"soak ok ({'numpy-RNG mode' if numpy_mode else 'spawn stream'}): 2^{lg} boards x {steps} steps bit-exact vs oracle in {time.time() - t0:.0f} s; episodes {st['episodes']}, "
      f"return sum {st['return_sum']} (mean episode score {st['mean_episode_score']:.1f}), "
      f"max tile 2^{st['max_exp']}, best last score {st['last_score_max']}, illegal ends {st['illegal_ends']}, "
      f"highest-tile histogram {[c for c in st['highest_hist'] if c]}")
