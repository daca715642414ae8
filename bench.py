#!/usr/bin/env python
"""bench.py -- env-steps/s of the batched 2048 step path on N MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--boards B]

A "step" is one pass of the hot path over the whole batch: ONE launch of step_kernel per rank that
advances every board by one action (move + score + spawn + done + auto-reset), reading the action
from, and writing reward/terminated to, [K][B] rollout buffers that are resident in HBM before the
timed region starts.  Workload = BASELINE configs[2]: B = 2^20 boards per GPU, synthetic uniform
random policy (actions pre-generated on the device by g2048_fill_random_actions), seed 42.

N > 1 (launched by torch.distributed.run, one rank per GPU): rank r owns global boards
[r*B, (r+1)*B) -- no data-path collective; one RCCL all-gather of the per-board episodic returns at
the end of the rollout, inside the timed region.  scaling = "weak".

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     -- algorithmic bytes (38 B/env-step) / HIP-event time per launch vs the 8 TB/s HBM peak
  cpu_baseline -- the C oracle (oracle/, "port") timed on this box's host cores on a bounded sample
  extras       -- fused K-step rollout kernel, a 2^24-board run that really streams HBM, Python port
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

ALGO_BYTES_PER_STEP = 38          # board in 16 + action 1 + board out 16 + reward 4 + terminated 1
HBM_PEAK_GBS = 8000.0             # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SEED = 42


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--boards", type=int, default=1 << 20, help="boards per GPU")
    ap.add_argument("--no-extras", action="store_true", help="skip fused / streaming / cpu legs")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    return ap.parse_args()


def cpu_baseline(target_seconds: float) -> dict:
    """The C oracle (oracle/libg2048_oracle.so, OpenMP over boards) on a bounded sample of the same
    workload: B_cpu boards, synthetic random policy, auto-reset, seed 42."""
    from oracle import OracleBatch
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    n = 1 << 18
    # calibrate the OpenMP team size: a cgroup CPU quota can make "all visible cores" far slower
    best_threads, best_rate = 1, 0.0
    for threads in sorted({1, 8, 32, 64, 128, avail}):
        if threads > avail:
            continue
        ob = OracleBatch(n, SEED, threads=threads)
        ob.reset()
        ob.step(None)
        t0 = time.perf_counter()
        ob.step(None)
        ob.step(None)
        rate = 2 * n / (time.perf_counter() - t0)
        if rate > best_rate:
            best_threads, best_rate = threads, rate
    ob = OracleBatch(n, SEED, threads=best_threads)
    ob.reset()
    steps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < target_seconds:      # bounded by wall time, not by a step count
        ob.step(None)
        steps += 1
    dt = time.perf_counter() - t0
    one = OracleBatch(1 << 16, SEED, threads=1)
    one.reset()
    one.step(None)
    t1 = time.perf_counter()
    for _ in range(16):
        one.step(None)
    dt1 = time.perf_counter() - t1
    return {"value": n * steps / dt, "unit": "env-steps/s", "cores": best_threads, "kind": "port",
            "sample": f"C oracle (oracle/g2048_oracle.c, OpenMP x{best_threads} of {avail} visible cores), {n} boards x "
                      f"{steps} steps, random policy, auto-reset, seed {SEED}; {dt:.1f} s",
            "single_core_value": (1 << 16) * 16 / dt1}


def python_port_rate() -> float:
    """oracle.cpu_ref.RefEnv, BASELINE configs[0] shape: one env, seed 42, random actions, 1 core."""
    from oracle.cpu_ref import RefEnv, random_action
    n = 20000
    acts = [random_action(SEED, t + 1, 0) for t in range(n)]
    env = RefEnv(SEED, 0)
    env.reset()
    t0 = time.perf_counter()
    for a in acts:
        if env.step(a)[1]:
            env.reset()
    return n / (time.perf_counter() - t0)


def load_traffic():
    """HBM bytes per launch from the committed PMC profile (profiles/traffic_latest.json), or None."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...`")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a ROCm GPU; the product has no CPU path")
    # G2048_BENCH_BACKEND=gloo + G2048_BENCH_SAME_DEVICE=1: smoke-test the N > 1 code path on a
    # one-GPU box (all ranks share cuda:0, collectives over gloo through host copies).
    backend = os.environ.get("G2048_BENCH_BACKEND", "nccl")
    if os.environ.get("G2048_BENCH_SAME_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build_hip()
    if world > 1:
        dist.barrier()
    from gym2048_amd.batched import Batched2048
    from gym2048_amd.sharding import weak_shard, allgather_returns

    B, K, W = args.boards, args.steps, args.warmup
    shard = weak_shard(B, rank, world)
    eng = Batched2048(B, device=local_rank, seed=SEED, board_offset=shard.offset)
    eng.reset()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def gather_returns():
        local = eng.last_scores()                  # int32[B] returns, from the terminal records (one kernel)
        if backend != "nccl" and world > 1:        # gloo smoke test: collectives on host copies
            return allgather_returns(local.cpu(), shard)
        return allgather_returns(local, shard)

    # ---- warm-up: W untimed steps (own small buffers), then inputs for the timed K steps
    if W > 0:
        wa = eng.random_actions(W)
        eng.rollout(wa)
        del wa
    actions = eng.random_actions(K)                      # [K][B] u8, resident in HBM
    reward = torch.empty((K, B), dtype=torch.float32, device=dev)
    terminated = torch.empty((K, B), dtype=torch.uint8, device=dev)
    reward.zero_()
    terminated.zero_()
    if world > 1:                                        # warm the collective once (communicator setup)
        gather_returns()

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record()
    eng.rollout(actions, reward=reward, terminated=terminated)   # EXACTLY K step launches
    ev1.record()
    gathered = gather_returns()                                  # once per rollout (N > 1: RCCL)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_region_ms = ev0.elapsed_time(ev1)

    tmax = torch.tensor([elapsed, kernel_region_ms], dtype=torch.float64,
                        device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed, kernel_region_ms = float(tmax[0]), float(tmax[1])
    assert gathered.numel() == B * world

    # sanity inside the bench: the rollout really happened (episodes finished, rewards written)
    stats = eng.episode_stats()
    total_steps = K * B * world
    value = total_steps / elapsed
    launch_us = kernel_region_ms * 1e3 / K
    achieved = ALGO_BYTES_PER_STEP * B / (launch_us * 1e-6) / 1e9
    traffic = load_traffic()

    out = {
        "metric": ("env-steps/sec at batch=2^20 per MI355X" if B == (1 << 20) else f"env-steps/sec at batch={B} per MI355X"), "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": elapsed * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"batch={B} envs per GPU, random-policy rollout, int8 boards (BASELINE configs[2])",
                   "boards_per_gpu": B, "global_boards": B * world, "seed": SEED,
                   "path": "one step_kernel launch per env-step (g2048_rollout), actions/reward/terminated in "
                           "[K][B] HBM rollout buffers, auto-reset fused",
                   "collective": "none per step; one all-gather of episodic returns per rollout" if world > 1
                   else "none"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic.get("bytes_per_launch") if (traffic and B == (1 << 20)) else None,
                     "kernel": "g2048::step_kernel<1>", "algorithmic_bytes_per_env_step": ALGO_BYTES_PER_STEP,
                     "launch_us": launch_us},
        "episodes_finished": int(stats["episodes"]), "mean_last_episode_score": stats["mean_last_score"],
    }

    if rank == 0 and world == 1 and not args.no_extras:
        extras = {}
        # (a) fused K-step rollout kernel: boards stay in registers; NOT HBM-bound, 38 B model n/a
        eng.rollout_random(8)
        torch.cuda.synchronize()
        kf = 256
        t1 = time.perf_counter()
        eng.rollout_random(kf)
        torch.cuda.synchronize()
        extras["fused_rollout_steps_per_s"] = kf * B / (time.perf_counter() - t1)
        # (a2) the SAME rollout as the timed region (same actions, same [K][B] outputs) as ONE fused launch
        try:
            kf2 = min(K, 256)
            t1 = time.perf_counter()
            eng.rollout(actions[:kf2], reward=reward[:kf2], terminated=terminated[:kf2], fused=True)
            torch.cuda.synchronize()
            extras["fused_rollout_with_io_steps_per_s"] = kf2 * B / (time.perf_counter() - t1)
        except Exception as exc:  # pragma: no cover
            extras["fused_rollout_with_io_steps_per_s"] = f"error: {exc}"
        # (b) a batch that does not fit L2 + Infinity Cache: 2^24 boards (256 MiB of boards)
        del reward, terminated, actions
        try:
            nb, kb = 1 << 24, 40
            big = Batched2048(nb, device=local_rank, seed=SEED)
            big.reset()
            ab = big.random_actions(kb)
            rb = torch.empty((kb, nb), dtype=torch.float32, device=dev)
            tb = torch.empty((kb, nb), dtype=torch.uint8, device=dev)
            big.rollout(ab[:8], reward=rb[:8], terminated=tb[:8])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            big.rollout(ab, reward=rb, terminated=tb)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / kb
            extras["streaming_2p24"] = {"boards": nb, "steps": kb, "launch_us": us,
                                        "steps_per_s": nb / (us * 1e-6),
                                        "achieved_GBs": ALGO_BYTES_PER_STEP * nb / (us * 1e-6) / 1e9}
            big.close()
            del ab, rb, tb
        except Exception as exc:  # pragma: no cover - memory pressure on a shared box
            extras["streaming_2p24"] = {"error": str(exc)}
        # (b2) numpy-compatible RNG mode (the reference's own PCG64 per board, seeded on the device)
        try:
            nn, kn = 1 << 20, 20
            npe = Batched2048(nn, device=local_rank, seed=SEED, rng="numpy")
            npe.reset()
            an = npe.random_actions(kn)
            npe.rollout(an[:5])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            npe.rollout(an)
            torch.cuda.synchronize()
            extras["numpy_rng_mode_steps_per_s"] = kn * nn / (time.perf_counter() - t1)
            npe.close()
        except Exception as exc:  # pragma: no cover
            extras["numpy_rng_mode_steps_per_s"] = f"error: {exc}"
        out["extras"] = extras
        # (c) CPU legs (rank 0, N = 1 only per the contract; cheap enough to always show at N = 1)
        if world == 1:
            ge.build()
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
            extras["python_port_steps_per_s_1core"] = python_port_rate()

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
