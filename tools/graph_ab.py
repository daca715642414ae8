"""Measurement tool: one k-step rollout over fixed [k][n] buffers launched kernel by kernel (stream) against the same
rollout replayed from the engine's cached hipGraph (g2048_rollout_prepare), interleaved in one process, HIP events and
wall clock, medians.  Usage: python tools/graph_ab.py [log2_boards ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["G2048_GRAPH_MAX_BOARDS"] = str(1 << 30)
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

ge.build_hip()
from gym2048_amd.batched import Batched2048  # noqa: E402

for lg in [int(x) for x in sys.argv[1:]] or [16, 18, 19, 20, 22]:
    n = 1 << lg
    for k in (20, 200):
        if n * k * 6 > 8e9:
            continue
        os.environ["G2048_ROLLOUT_GRAPH"] = "1"
        g = Batched2048(n, seed=42, last_records=False)
        os.environ["G2048_ROLLOUT_GRAPH"] = "0"
        s = Batched2048(n, seed=42, last_records=False)
        engines = {"graph": g, "stream": s}
        plans, times = {}, {"graph": ([], []), "stream": ([], [])}
        for name, e in engines.items():
            e.reset()
            e.rollout_random(64)
            a = e.random_actions(k)
            r = torch.zeros((k, n), dtype=torch.float32, device=e.device)
            t = torch.zeros((k, n), dtype=torch.uint8, device=e.device)
            plans[name] = e.prepare_rollout(a, reward=r, terminated=t)
            if name == "graph":
                plans[name].prepare_graph()
        for rnd in range(-3, 25):
            for name in (("graph", "stream") if rnd % 2 else ("stream", "graph")):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                w0 = time.perf_counter()
                e0.record()
                plans[name].run()
                e1.record()
                torch.cuda.synchronize()
                w = (time.perf_counter() - w0) * 1e6 / k
                if rnd >= 0:
                    times[name][0].append(e0.elapsed_time(e1) * 1e3 / k)
                    times[name][1].append(w)
        assert g.graph_replays >= 25 and s.graph_replays == 0
        line = f"2^{lg} boards, k = {k:3d}:"
        for name in ("stream", "graph"):
            ev, wl = sorted(times[name][0]), sorted(times[name][1])
            line += f"  {name}: events median {ev[len(ev) // 2]:7.3f} us/step (min {ev[0]:7.3f}), wall median {wl[len(wl) // 2]:7.3f}"
        print(line, flush=True)
        g.close()
        s.close()
