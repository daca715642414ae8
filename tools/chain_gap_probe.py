"""Measurement tool: a forced two-chain 20-step rollout (G2048_TWO_CHAIN_MIN_STEPS=2) after a warm 128-step rollout, a
stream synchronisation and X microseconds of host idle -- against the same with one chain.  HIP-event time."""
import os, sys, time
sys.path.insert(0, ".")
os.environ["G2048_TWO_CHAIN_MIN_STEPS"] = "2"
import numpy as np
import torch
import __graft_entry__ as ge
ge.build_hip()
from gym2048_amd.batched import Batched2048

B, K = 1 << 20, 20
for chains in (1, 2):
    eng = Batched2048(B, seed=42, last_records=False, chains=chains)
    eng.reset(); eng.rollout_random(64)
    acts = eng.random_actions(128)
    rew = torch.zeros((128, B), dtype=torch.float32, device=eng.device)
    term = torch.zeros((128, B), dtype=torch.uint8, device=eng.device)
    warm = eng.prepare_rollout(acts, reward=rew, terminated=term)
    plan = eng.prepare_rollout(acts[:K], reward=rew[:K], terminated=term[:K])
    for _ in range(20):
        warm.run()
    torch.cuda.synchronize()
    out = []
    for gap in (0, 200, 5000, 50000, 300000):
        ts, used = [], set()
        for rep in range(9):
            warm.run(); torch.cuda.synchronize()
            t = time.perf_counter()
            while (time.perf_counter() - t) * 1e6 < gap:
                pass
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); plan.run(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3); used.add(eng.chains_used)
        out.append(f"gap {gap} us: {np.median(ts):.1f} ({np.median(ts) / K:.2f}/step, chains used {sorted(used)})")
    print(f"engine chains {chains}: " + "; ".join(out))
    eng.close()
