"""Tool: where the fixed cost of bench.py's N > 1 timed region comes from, on a one-GPU box.
A one-rank RCCL process group; the 20-launch train of bench.py is timed (HIP events + wall clock) with the pieces of
the once-per-rollout exchange switched on one at a time.  Run: python tools/dist_probe.py [reps]"""
import os
import sys
import time

sys.path.insert(0, ".")
import torch
import torch.distributed as dist

from gym2048_amd.batched import Batched2048
from gym2048_amd.sharding import allgather_stats, allgather_returns, weak_shard, SummaryExchange

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29611")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
B, K = 1 << 20, 20
eng = Batched2048(B, seed=42)
eng.reset()
eng.rollout_random(64)
actions = eng.random_actions(K)
reward = torch.zeros((K, B), dtype=torch.float32, device=dev)
term = torch.zeros((K, B), dtype=torch.uint8, device=dev)
plan = eng.prepare_rollout(actions, reward=reward, terminated=term)
stats_buf = torch.empty(176, dtype=torch.uint8, device=dev)   # sizeof(g2048_stats)
ret_buf = torch.empty(B, dtype=torch.int32, device=dev)
shard = weak_shard(B, 0, 1)
for _ in range(30):
    plan.run()
torch.cuda.synchronize()


def region(opening, tail):
    """opening(): the bracket before the train; tail(): what is enqueued behind the K launches."""
    res = []
    for _ in range(reps):
        plan.run()                       # warm-up steps
        opening()
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        t0 = time.perf_counter()
        plan.run()
        e1.record()
        tail()
        e2.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e6
        res.append((wall, e0.elapsed_time(e1) * 1e3, e1.elapsed_time(e2) * 1e3))
    res.sort()
    return res[len(res) // 2]


def show(name, r):
    print(f"{name:78s} wall {r[0]:7.1f} us | launch train {r[1]:7.1f} | tail {r[2]:6.1f} | host rest {r[0] - r[1] - r[2]:6.1f}")


nothing = lambda: None  # noqa: E731
show("no process group: sync | 20 launches | sync", region(nothing, nothing))
show("no process group: + returns-only summary kernels", region(nothing, lambda: eng.episode_stats_device(out=stats_buf, returns_only=True)))
show("no process group: + full statistics kernels", region(nothing, lambda: eng.episode_stats_device(out=stats_buf)))
show("no process group: + last-returns export kernel (int32[B])", region(nothing, lambda: eng.last_scores(out=ret_buf)))
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
dist.barrier()
show("one-rank RCCL group, no barrier: 20 launches", region(nothing, nothing))
show("one-rank RCCL group: barrier | 20 launches", region(dist.barrier, nothing))
show("one-rank RCCL group: barrier | 20 launches + summary kernels", region(dist.barrier, lambda: eng.episode_stats_device(out=stats_buf, returns_only=True)))
show("one-rank RCCL group: barrier | 20 launches + summary kernels + all-gather(176 B)",
     region(dist.barrier, lambda: allgather_stats(eng.episode_stats_device(out=stats_buf, returns_only=True))))
ex = SummaryExchange(eng)                  # the library's own communicator (one rank): summary launch + in-place ncclAllGather
ex.gather()
torch.cuda.synchronize()
show("one-rank RCCL group: barrier | 20 launches + g2048_allgather_summary (direct, launch stream)", region(dist.barrier, ex.gather))
show("one-rank RCCL group: barrier | 20 launches + all-gather(176 B) of a stale buffer", region(dist.barrier, lambda: allgather_stats(stats_buf)))
show("one-rank RCCL group: barrier | 20 launches + export + all-gather(int32[B])",
     region(dist.barrier, lambda: allgather_returns(eng.last_scores(out=ret_buf), shard)))
ex.close()
dist.destroy_process_group()
