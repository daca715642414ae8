#!/usr/bin/env python
"""Stress of LocalShards on the one GPU (test infrastructure): 2-4 two-chain shard engines on cuda:0, each with its own
launch thread and its own non-default stream, all sharing the device's side chain; random rollouts with pauses, every
reward and the final boards compared with ONE one-chain engine holding all the boards (global-index spawn stream: the
games do not depend on the split).
    python tests/stress_local_shards.py [seconds=60]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import __graft_entry__ as ge
ge.build()
from gym2048_amd import LocalShards
from gym2048_amd.batched import Batched2048
import os
os.environ["G2048_TWO_CHAIN_MIN_STEPS"] = "2"
rs = np.random.default_rng(3)
t_end = time.time() + float(sys.argv[1]) if len(sys.argv) > 1 else time.time() + 60
cases = rolls = 0
while time.time() < t_end:
    g = int(rs.integers(2, 5)); n = int(rs.choice([512, 1024, 4096, 1 << 16])); seed = int(rs.integers(0, 1 << 30))
    streams = [torch.cuda.Stream(device=0) for _ in range(g)]
    sh = LocalShards(n, devices=[0] * g, seed=seed, streams=streams, chains=2)
    ref = Batched2048(n * g, seed=seed, chains=1)
    sh.reset(); ref.reset()
    for _ in range(int(rs.integers(2, 12))):
        k = int(rs.integers(1, 60))
        acts = torch.as_tensor(rs.integers(0, 4, (k, n * g)).astype(np.uint8)).cuda()
        rew = [torch.zeros((k, n), dtype=torch.float32, device="cuda:0") for _ in range(g)]
        torch.cuda.synchronize()
        sh.rollout([acts[:, r * n:(r + 1) * n].contiguous() for r in range(g)], reward=rew)
        rr = torch.zeros((k, n * g), dtype=torch.float32, device="cuda:0")
        ref.rollout(acts, reward=rr)
        sh.synchronize(); torch.cuda.synchronize()
        got = torch.cat(rew, dim=1)
        assert torch.equal(got, rr), (g, n, seed, k)
        rolls += 1
        if rs.random() < 0.3: time.sleep(float(rs.choice([1e-4, 1e-3, 4e-3])))
    b = np.concatenate([e.get_boards().reshape(n, 16) for e in sh.engines])
    assert np.array_equal(b, ref.get_boards().reshape(n * g, 16))
    sh.close(); ref.close(); cases += 1
print(f"local shards stress ok: {cases} shard sets, {rolls} rollouts")
