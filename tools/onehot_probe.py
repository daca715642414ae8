import sys, time
sys.path.insert(0, ".")
import torch
import __graft_entry__ as ge
ge.build_hip()
from gym2048_amd.batched import Batched2048
n = 1 << 20
eng = Batched2048(n, seed=1); eng.reset(); eng.rollout_random(64)
for dt in (torch.uint8, torch.float16, torch.float32):
    obs = torch.zeros((n, 16, 4, 4), dtype=dt, device=eng.device)
    for _ in range(5): eng.observe_onehot(out=obs)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): eng.observe_onehot(out=obs)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(dt, round(us, 1), "us", round(obs.numel() * obs.element_size() / us / 1e6, 2), "TB/s written")
