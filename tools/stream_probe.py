"""Probe (tool): how the 2^24-board streaming launch time depends on what the GPU did just before."""
import sys, time
sys.path.insert(0, ".")
import torch
import __graft_entry__ as ge
ge.build_hip()
from gym2048_amd.batched import Batched2048
dev = torch.device("cuda", 0)
nb, kb = 1 << 24, 24
big = Batched2048(nb, seed=42)
big.reset()
small = Batched2048(1 << 20, seed=1)
small.reset()
ab = big.random_actions(kb)
rb = torch.zeros((kb, nb), dtype=torch.float32, device=dev)
tb = torch.zeros((kb, nb), dtype=torch.uint8, device=dev)


def runs(label, n=6):
    out = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        big.rollout(ab, reward=rb, terminated=tb)
        e1.record()
        torch.cuda.synchronize()
        out.append(round(e0.elapsed_time(e1) * 1e3 / kb, 1))
    print(f"{label:40s}", out, "clock", big.clock, flush=True)


runs("cold start")
runs("again")
t = time.time()
while time.time() - t < 0.5:
    small.rollout_random(256)
    torch.cuda.synchronize()
runs("after 0.5 s of fused compute (2^20)")
runs("again")
time.sleep(1.0)
runs("after 1 s idle")
big.rollout_random(64)
runs("after 64 fused steps on the big engine")
runs("again")
