import sys, time
sys.path.insert(0, ".")
import torch
from gym2048_amd.batched import Batched2048
for lg in (20, 16, 24):
    e = Batched2048(1 << lg, seed=3); e.reset(); e.rollout_random(80)
    torch.cuda.synchronize()
    for _ in range(3): e.episode_stats_device()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(20): d = e.episode_stats_device()
    ev1.record(); torch.cuda.synchronize()
    print(f"2^{lg}: episode_stats_device {ev0.elapsed_time(ev1) * 1e3 / 20:.1f} us per call", e.episode_stats()["episodes"])
    buf = torch.empty(176, dtype=torch.uint8, device=e.device)   # sizeof(g2048_stats)
    for _ in range(3): e.episode_stats_device(out=buf, returns_only=True)
    ev0.record()
    for _ in range(20): e.episode_stats_device(out=buf, returns_only=True)
    ev1.record(); torch.cuda.synchronize()
    print(f"2^{lg}: returns-only summary  {ev0.elapsed_time(ev1) * 1e3 / 20:.1f} us per call")
    e.close()
