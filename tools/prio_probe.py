"""What happens when the caller's stream has the side stream's priority (G2048_CHAIN_ANY_PRIORITY=1 lifts the rule that
gives such a caller one chain)?  Many high-priority streams are created and kept busy so that the runtime has to share
hardware queues among them; a two-chain rollout is then issued from each in turn and timed."""
import sys
import time

import torch

sys.path.insert(0, ".")
from gym2048_amd.batched import Batched2048

n, k = 1 << 20, 70
nstreams = int(sys.argv[1]) if len(sys.argv) > 1 else 1
highs = [torch.cuda.Stream(priority=-1) for _ in range(nstreams)]
junk = [torch.zeros(1 << 18, device="cuda") for _ in highs]
ref = Batched2048(n, seed=5, chains=1)
ref.reset()
want = torch.zeros((k, n), dtype=torch.float32, device=ref.device)
ref.rollout(k, reward=want)
torch.cuda.synchronize()
for i, high in enumerate(highs):
    eng = Batched2048(n, seed=5, chains=2)
    got = torch.zeros((k, n), dtype=torch.float32, device=eng.device)
    with torch.cuda.stream(high):
        eng.reset()
        torch.cuda.synchronize()
        for h, j in zip(highs, junk):
            with torch.cuda.stream(h):
                j.add_(1.0).cumsum_(0)
        t0 = time.perf_counter()
        eng.rollout(k, reward=got)
        torch.cuda.synchronize()
    print(i, "chains_used", eng.chains_used, "took %.1f us" % ((time.perf_counter() - t0) * 1e6), "equal", bool(torch.equal(got, want)), flush=True)
    eng.close()
