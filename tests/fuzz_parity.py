#!/usr/bin/env python
"""Differential fuzz (test infrastructure; lives under tests/ because it drives the oracle): random batch sizes,
configurations and CALL MIXES of the C ABI against the C oracle, every output compared after every call.
Each case draws: n in 1..6000 (whole blocks, ragged, single boards), seed, board_offset (incl. near 2^32), illegal
reward, max_tile, auto_reset; then 12-40 calls chosen among
  step (all optional outputs, random action dtype, fused observation of a random dtype or none)
  rollout of k steps over [k, n] buffers (with / without observation and boards_out)
  fused rollout of k steps (pipelined action loads: every tail length occurs)
  rollout_random(k)
  step_host (host-resident I/O)
  masked reset, set_boards, set_scores, state save / restore into a second engine
and after every call the boards, scores, last returns, episode counts and the exact return sum (both statistics flavours)
    python tests/fuzz_parity.py [seconds=120] [seed=0]          (G2048_FUZZ_STREAMS=1: every case on its own non-default stream)"""
import ctypes as C
import os
import sys
import time

os.environ.setdefault("G2048_TWO_CHAIN_MIN_STEPS", "2")   # two-chain engines split EVERY rollout of >= 2 steps (the
                                                          # product only does so for long or back-to-back rollouts)

sys.path.insert(0, ".")
import numpy as np
import torch

import __graft_entry__ as ge

ge.build()
from gym2048_amd import _lib
from gym2048_amd.batched import Batched2048, parse_stats
from oracle import OracleBatch

rs = np.random.default_rng(0)
DT = {0: torch.uint8, 1: torch.float16, 2: torch.float32}
ADT = [torch.uint8, torch.int32, torch.int64]
counts = {}


def bump(name, by=1):
    counts[name] = counts.get(name, 0) + by


def check_state(eng, ora, where, always_auto_reset=False):
    n = eng.n_envs
    assert np.array_equal(eng.get_boards().reshape(n, 16), ora.boards), where
    assert np.array_equal(eng.get_scores(), ora.score), where
    if eng.last_records_enabled:
        assert np.array_equal(eng.get_last_scores(), ora.last_score), where
    st = eng.episode_stats()
    assert st["episodes"] == int(ora.ep_count.sum()), where
    assert st["return_sum"] == ora.return_sum, (where, st["return_sum"], ora.return_sum, ora.finished_return_sum)
    if always_auto_reset:                       # then it IS the sum of the final scores of all finished episodes
        assert st["return_sum"] == ora.finished_return_sum, where
    ro = parse_stats(eng.episode_stats_device(returns_only=True))
    assert (ro["episodes"], ro["illegal_ends"], ro["return_sum"]) == (st["episodes"], st["illegal_ends"], st["return_sum"]), where


def one_case(case):
    n = int(rs.choice([1, 2, 63, 64, 65, 255, 256, 257, 511, 512, 1000, 4096, int(rs.integers(1, 6000))]))
    seed = int(rs.integers(0, 2**63))
    offset = int(rs.choice([0, 12345, (1 << 32) - n, int(rs.integers(0, (1 << 32) - n))]))
    irw = float(rs.choice([0.0, -1.0, 2.5]))
    max_tile = rs.choice([None, 8, 32, 2048])
    max_tile = None if max_tile is None else int(max_tile)
    auto_reset = bool(rs.random() < 0.8)
    numpy_mode = bool(rs.random() < 0.2)       # the reference's own PCG64 per board (separate kernels)
    if numpy_mode:
        n, seed, offset = min(n, 1500), seed % (1 << 40), offset % (1 << 20)
    keep_last = bool(rs.random() < 0.75)       # per-board terminal records on / off (g2048_set_last_records)
    chains = int(rs.choice([1, 2]))            # rollouts as one chain or as two half-batch chains (g2048_set_chains)
    eng = Batched2048(n, seed=seed, board_offset=offset, illegal_move_reward=irw, max_tile=max_tile,
                      rng="numpy" if numpy_mode else "philox", last_records=keep_last, chains=chains)
    ora = OracleBatch(n, seed, offset)
    ora.illegal_move_reward = irw
    ora.max_exp = 0 if max_tile is None else max_tile.bit_length() - 1
    if numpy_mode:
        ora.seed_numpy(seed)
        ora.step = ora.step_numpy               # same signature
    eng.reset()
    ora.reset_numpy() if numpy_mode else ora.reset()
    dev = eng.device
    bump("numpy_mode_cases" if numpy_mode else "philox_cases")
    tag = f"case {case}: n={n} seed={seed} offset={offset} irw={irw} max_tile={max_tile} auto_reset={auto_reset} numpy={numpy_mode} chains={chains}"
    replay_bufs = replay_plan = None
    for call in range(int(rs.integers(12, 40))):
        kind = str(rs.choice(["step", "rollout", "fused", "random", "host", "mask_reset", "set_boards", "state", "set_scores", "replay"],
                             p=[0.26, 0.16, 0.13, 0.08, 0.1, 0.06, 0.04, 0.06, 0.03, 0.08]))
        if numpy_mode and kind in ("mask_reset", "replay"):   # unmaskable oracle reset / spawn-stream-only graph replays
            kind = "step"                                      # (fused and random rollouts exist in numpy-RNG mode since round 6)
        where = f"{tag} call {call} {kind}"
        bump(kind)
        if kind == "step":
            adt = ADT[int(rs.integers(0, 3))]
            a = rs.integers(0, 4, n)
            od = int(rs.integers(-1, 3))
            obs = None if od < 0 else torch.zeros((n, 16, 4, 4), dtype=DT[od], device=dev)
            eng.step(torch.as_tensor(a).to(adt), auto_reset=auto_reset, obs=obs)
            ora.step(a.astype(np.uint8), auto_reset=auto_reset)
            assert np.array_equal(eng.reward.cpu().numpy(), ora.reward), where
            assert np.array_equal(eng.terminated.cpu().numpy(), ora.terminated), where
            assert np.array_equal(eng.illegal.cpu().numpy(), ora.illegal), where
            assert np.array_equal(eng.highest.cpu().numpy(), ora.highest), where
            done = ora.terminated.astype(bool)
            assert np.array_equal(eng.terminal_boards.cpu().numpy()[done], ora.terminal_boards[done]), where
            if obs is not None:
                got = obs.cpu().numpy()
                assert np.array_equal(got, ora.onehot().astype(got.dtype)), where
        elif kind in ("rollout", "fused"):
            k = int(rs.integers(1, 24))                 # (>= 12: two chains when on)
            acts = (eng.random_actions(k) if rs.random() < 0.5 and not numpy_mode
                    else torch.as_tensor(rs.integers(0, 4, (k, n)).astype(np.uint8)).to(dev))
            if kind == "fused":
                acts = acts.to(ADT[int(rs.integers(0, 3))])
            rew = torch.zeros((k, n), dtype=torch.float32, device=dev)
            term = torch.zeros((k, n), dtype=torch.uint8, device=dev)
            obs = None
            bout = None
            if kind == "rollout" and rs.random() < 0.5:
                obs = torch.zeros((k, n, 16, 4, 4), dtype=DT[int(rs.integers(0, 3))], device=dev)
            if kind == "rollout" and rs.random() < 0.5:
                bout = torch.zeros((k, n, 16), dtype=torch.uint8, device=dev)
            if bout is None:
                eng.rollout(acts, reward=rew, terminated=term, auto_reset=auto_reset, fused=(kind == "fused"), obs=obs)
            else:
                io = eng._io(acts, rew, term, None, None, None, obs)
                io.boards_out = bout.data_ptr()
                _lib.check(eng._lib.g2048_rollout(eng._h, k, C.byref(io), n, int(auto_reset), eng._stream()))
            a = (acts.cpu().numpy() & 3).astype(np.uint8)
            for j in range(k):
                ora.step(a[j], auto_reset=auto_reset)
                assert np.array_equal(rew[j].cpu().numpy(), ora.reward), (where, j)
                assert np.array_equal(term[j].cpu().numpy(), ora.terminated), (where, j)
                if obs is not None:
                    got = obs[j].cpu().numpy()
                    assert np.array_equal(got, ora.onehot().astype(got.dtype)), (where, j)
                if bout is not None:
                    assert np.array_equal(bout[j].cpu().numpy(), ora.boards), (where, j)
        elif kind == "replay":
            # the SAME [k, n] buffers rollout after rollout (what a trainer with fixed rollout buffers does): from the second
            # one in a row -- or the first, when the plan was prepared -- g2048_rollout replays its cached hipGraph
            if replay_bufs is None:
                kg = int(rs.integers(8, 20))
                replay_bufs = (torch.zeros((kg, n), dtype=ADT[int(rs.integers(0, 3))], device=dev),
                               torch.zeros((kg, n), dtype=torch.float32, device=dev), torch.zeros((kg, n), dtype=torch.uint8, device=dev))
                replay_plan = eng.prepare_rollout(replay_bufs[0], reward=replay_bufs[1], terminated=replay_bufs[2], auto_reset=auto_reset)
                if rs.random() < 0.5:
                    replay_plan.prepare_graph()
            acts, rew, term = replay_bufs
            for _ in range(int(rs.integers(1, 4))):
                a = rs.integers(0, 4, acts.shape).astype(np.uint8)
                acts.copy_(torch.as_tensor(a).to(dev))
                before = eng.graph_replays
                replay_plan.run()
                bump("graph_replays", eng.graph_replays - before)
                for j in range(acts.shape[0]):
                    ora.step(a[j], auto_reset=auto_reset)
                    assert np.array_equal(rew[j].cpu().numpy(), ora.reward), (where, j)
                    assert np.array_equal(term[j].cpu().numpy(), ora.terminated), (where, j)
        elif kind == "random":
            if not auto_reset:
                continue
            k = int(rs.integers(1, 30))
            eng.rollout_random(k)
            for _ in range(k):
                ora.step(None)
        elif kind == "host":
            io = eng.host_io()
            a = rs.integers(0, 4, n)
            io["actions"][:] = a
            out = eng.step_host(auto_reset)
            ora.step(a.astype(np.uint8), auto_reset=auto_reset)
            assert np.array_equal(out["reward"], ora.reward) and np.array_equal(out["terminated"], ora.terminated), where
            assert np.array_equal(out["boards"].reshape(n, 16), ora.boards), where
            assert np.array_equal(out["illegal"], ora.illegal) and np.array_equal(out["highest"], ora.highest), where
        elif kind == "mask_reset":
            mask = (rs.random(n) < 0.3).astype(np.uint8)
            eng.reset(mask=mask)
            ora.reset(mask=mask)
        elif kind == "set_boards":
            b = (rs.integers(0, 12, (n, 16)) * (rs.random((n, 16)) < 0.6)).astype(np.uint8)
            eng.set_boards(b)
            ora.boards[:] = b
        elif kind == "set_scores":
            sc = rs.integers(0, 1 << 20, n).astype(np.int32)
            eng.set_scores(sc if rs.random() < 0.5 else torch.as_tensor(sc).to(dev))
            ora.set_scores(sc)
        else:  # state save / restore into a fresh engine that then replaces the original
            blob = eng.state_dict()
            other = Batched2048(n, seed=1, board_offset=0, rng="numpy" if numpy_mode else "philox", chains=chains)
            other.load_state_dict(blob)
            eng.close()
            eng = other
            replay_bufs = replay_plan = None                             # (a plan belongs to the engine it was made for)
            assert eng.last_records_enabled == keep_last, where          # travels with the state blob
            eng.set_illegal_move_reward(irw)
            eng.set_max_tile(max_tile)
        check_state(eng, ora, where, always_auto_reset=auto_reset)
    eng.close()


def run(budget: float = 120.0, seed: int = 0) -> str:
    """Fuzz for ``budget`` seconds; returns the summary line (raises AssertionError on the first mismatch)."""
    global rs
    rs = np.random.default_rng(seed)
    counts.clear()
    t0 = time.time()
    case = 0
    # G2048_FUZZ_STREAMS=1: every case runs on a fresh NON-DEFAULT (non-blocking) torch stream instead of the default
    # stream -- nothing in the library may rely on the ordering the null stream gives for free
    own_streams = os.environ.get("G2048_FUZZ_STREAMS") == "1"
    while time.time() - t0 < budget:
        if own_streams:
            with torch.cuda.stream(torch.cuda.Stream()):
                one_case(case)
        else:
            one_case(case)
        case += 1
    return (f"fuzz ok: {case} cases in {time.time() - t0:.0f} s, calls {dict(sorted(counts.items()))}; every output of "
            f"every call bit-exact vs the oracle")


if __name__ == "__main__":
    print(run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0))
