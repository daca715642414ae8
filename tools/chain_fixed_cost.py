"""Measurement tool: fixed cost of a two-chain rollout (fork / join / hand-off) -- HIP-event time of rollout(K) for a few K,
one chain vs two chains (G2048_TWO_CHAIN_MIN_STEPS=2 forces two chains for every K >= 2), and the linear fit."""
import os, sys
sys.path.insert(0, ".")
os.environ.setdefault("G2048_TWO_CHAIN_MIN_STEPS", "2")
import numpy as np
import torch
import __graft_entry__ as ge
ge.build_hip()
from gym2048_amd.batched import Batched2048

B = 1 << 20
for chains in (1, 2):
    eng = Batched2048(B, seed=42, last_records=False, chains=chains)
    eng.reset(); eng.rollout_random(64)
    ks = [2, 4, 8, 16, 32, 64, 128]
    kmax = max(ks)
    acts = eng.random_actions(kmax)
    rew = torch.zeros((kmax, B), dtype=torch.float32, device=eng.device)
    term = torch.zeros((kmax, B), dtype=torch.uint8, device=eng.device)
    warm = eng.prepare_rollout(acts, reward=rew, terminated=term)
    for _ in range(30):
        warm.run()
    torch.cuda.synchronize()
    res = []
    for k in ks:
        plan = eng.prepare_rollout(acts[:k], reward=rew[:k], terminated=term[:k])
        ts = []
        for rep in range(25):
            warm.run()                       # keep the device in its sustained state
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); plan.run(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        res.append(np.median(ts))
    slope, icpt = np.polyfit(ks[2:], res[2:], 1)
    print(f"chains {chains}: " + "  ".join(f"K={k}: {t:.1f}" for k, t in zip(ks, res)) + f"   fit over K >= 8: {slope:.3f} us/step + {icpt:.1f} us")
    eng.close()
