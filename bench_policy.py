#!/usr/bin/env python
"""bench_policy.py -- BASELINE configs[4]: the env driven by a ppo_train.py-shaped policy on the
same GPU, zero host copies.

    python bench_policy.py [--boards 1048576] [--steps 20] [--chunk 131072] [--dtype float16]

Loop per env-step (everything stays in HBM, everything on torch's current HIP stream):
  1. policy forward in chunks on the observation tensor: the trunk of ``ppo_train.py:36-62`` (conv3x3 16->64, BN,
     ReLU, 4 x the residual block of ``model.py:10-25``) + the action head SB3 adds on the flattened 1024 features
     (``net_arch=[]``, ``ppo_train.py:131-133``) -> greedy ``argmax`` (int64), random-init weights;
  2. ``Batched2048.step(actions, obs=obs)``: ONE launch reads the int64 action tensor through its data_ptr, plays
     the step and writes the next (N,16,4,4) observation into the torch tensor (game2048_env.py:100).
The first observation comes from ``observe_onehot`` after the reset.

This is a consumer-side measurement (the policy is PyTorch-ROCm/MIOpen, out of scope of this repo): the line leads
with what the ENV costs in that loop (`env_only`: the step-with-observation launches alone, from HIP events around them)
and then gives the whole loop and the env's share of it.  MIOpen chooses its convolution kernels the first time it sees a
shape (a "find" that benchmarks candidates, a naive fallback kernel of ~100 ms among them): that happens in an explicit
policy warm-up over every chunk shape BEFORE anything is timed, and `policy_forward_ms_per_step_runs` lists every timed
step so that a find that leaked into one would show.  `tools/gpu_profile.sh <tag> --policy` runs this file twice -- once
to fill MIOpen's user find-db, once under `rocprofv3 --kernel-trace` -- and fails if any `naive_conv*` kernel appears in
the second process.  Prints one JSON line.  Not the driver's benchmark (that is bench.py).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build_policy(torch, filters=64, blocks=4):
    nn = torch.nn

    class Residual(nn.Module):                     # shape of model.py:10-25
        def __init__(self, f):
            super().__init__()
            self.c1 = nn.Conv2d(f, f, 3, padding=1, bias=False)
            self.b1 = nn.BatchNorm2d(f)
            self.c2 = nn.Conv2d(f, f, 3, padding=1, bias=False)
            self.b2 = nn.BatchNorm2d(f)

        def forward(self, x):
            y = torch.relu(self.b1(self.c1(x)))
            return torch.relu(self.b2(self.c2(y)) + x)

    return nn.Sequential(                           # shape of ppo_train.py:53-59 + SB3's action_net
        nn.Conv2d(16, filters, 3, padding=1, bias=False), nn.BatchNorm2d(filters), nn.ReLU(inplace=True),
        *[Residual(filters) for _ in range(blocks)], nn.Flatten(), nn.Linear(filters * 16, 4))


def run(boards=1 << 20, steps=20, warmup=3, chunk=1 << 17, dtype="float16") -> dict:
    """The measurement; returns the JSON-able result (bench.py embeds it as extras.policy_loop)."""
    args = argparse.Namespace(boards=boards, steps=steps, warmup=warmup, chunk=chunk, dtype=dtype)
    import torch
    if not torch.cuda.is_available():
        sys.exit("needs a ROCm GPU")
    import __graft_entry__ as ge
    ge.build_hip()
    from gym2048_amd.batched import Batched2048

    dev = torch.device("cuda", torch.cuda.current_device())
    dt = getattr(torch, args.dtype)
    torch.manual_seed(0)
    policy = build_policy(torch).to(dev).to(dt).eval().to(memory_format=torch.channels_last)
    n = args.boards
    eng = Batched2048(n, device=dev.index, seed=42)
    eng.reset()
    obs = torch.empty((n, 16, 4, 4), dtype=dt, device=dev)
    actions = torch.empty(n, dtype=torch.int64, device=dev)

    eng.observe_onehot(out=obs)                    # observation of the reset; every later one comes from step()
    # MIOpen's kernel selection for every shape the loop will use (the full chunk and, when n is not a multiple of it, the
    # ragged last one), outside of anything that is timed or counted as a warm-up step
    t_find = time.perf_counter()
    with torch.no_grad():
        for lo in sorted({0, (n - 1) // args.chunk * args.chunk}):
            for _ in range(2):
                policy(obs[lo:lo + args.chunk].contiguous(memory_format=torch.channels_last)).argmax(dim=1)
    torch.cuda.synchronize()
    find_s = time.perf_counter() - t_find

    def one_step(timers=None):
        t0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), \
            torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0[0].record()
        t0[1].record()
        with torch.no_grad():
            for lo in range(0, n, args.chunk):
                logits = policy(obs[lo:lo + args.chunk].contiguous(memory_format=torch.channels_last))
                actions[lo:lo + args.chunk] = logits.argmax(dim=1)
        t0[2].record()
        eng.step(actions, want_info=False, obs=obs)
        t0[3].record()
        if timers is not None:
            timers.append(t0)

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    timers = []
    t_start = time.perf_counter()
    for _ in range(args.steps):
        one_step(timers)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t_start
    onehot_ms = sum(t[0].elapsed_time(t[1]) for t in timers) / args.steps
    policy_runs = [t[1].elapsed_time(t[2]) for t in timers]
    policy_ms = sum(policy_runs) / args.steps
    step_ms = sum(t[2].elapsed_time(t[3]) for t in timers) / args.steps
    stats = eng.episode_stats()
    eng.close()
    obs_bytes = 38 + 256 * torch.empty(0, dtype=dt).element_size()      # the step's 38 B + the one-hot it writes
    return ({
        "metric": "env-steps/sec with a ppo_train.py-shaped policy in the loop (BASELINE configs[4])",
        # first what the ENV costs in this loop: the int64-action step that writes its own fp16 / fp32 observation
        "env_only": {"steps_per_s": n / (step_ms * 1e-3), "ms_per_step": step_ms, "launches_per_env_step": 1,
                     "algorithmic_bytes_per_env_step": obs_bytes,
                     "frac_of_hbm_peak": obs_bytes * n / (step_ms * 1e-3) / 8e12},
        "value": n * args.steps / wall, "unit": "env-steps/s", "boards": n, "steps": args.steps,
        "policy_dtype": args.dtype, "policy_chunk": args.chunk,
        "ms_per_step": {"policy_forward_argmax": policy_ms, "env_step_with_observation": step_ms,
                        "wall": wall * 1e3 / args.steps},
        "policy_forward_ms_per_step_runs": policy_runs,          # every timed step: a MIOpen find in one of them would show
        "policy_find_warmup_s": find_s,                           # kernel selection for every chunk shape, before anything timed
        "launches_per_env_step": 1,
        "env_fraction_of_loop": (onehot_ms + step_ms) / (onehot_ms + policy_ms + step_ms),
        "host_copies": 0, "episodes_finished": int(stats["episodes"]),
        "mean_last_episode_score": stats["mean_last_score"], "max_tile": 1 << int(stats["max_exp"])})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--boards", type=int, default=1 << 20)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--chunk", type=int, default=1 << 17, help="policy forward batch")
    ap.add_argument("--dtype", default="float16", choices=["float16", "float32"])
    a = ap.parse_args()
    print(json.dumps(run(a.boards, a.steps, a.warmup, a.chunk, a.dtype)))


if __name__ == "__main__":
    main()
