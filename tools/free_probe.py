"""Does destroying an engine wait for its work in flight on a NON-DEFAULT stream?  (g2048_destroy relies on hipFree
synchronising the device.)  A long fused rollout is enqueued on a fresh non-blocking stream and the engine is closed
at once: close() must take about as long as the rollout."""
import sys
import time

import torch

sys.path.insert(0, ".")
from gym2048_amd.batched import Batched2048

for chains in (1, 2):
    eng = Batched2048(1 << 20, seed=1, chains=chains)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        eng.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.rollout_random(20000)
        t1 = time.perf_counter()
        eng.close()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
    print(f"chains={chains}: enqueue {1e3 * (t1 - t0):.2f} ms, close() {1e3 * (t2 - t1):.2f} ms, synchronize after {1e3 * (t3 - t2):.2f} ms")
