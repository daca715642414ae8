"""Batched counterpart of the reference's evaluation loop (``/root/reference/train.py:127-229``:
``choose_action``, ``evaluate_episode``, ``evaluate_model``, ``report_evaluation_results``).

The reference plays ``episodes`` games one after the other -- ``env.reset(seed=456 + i)``, epsilon-greedy
actions from Python's ``random`` seeded with ``123 + i``, stop at the first ``terminated`` or after 2 001
moves -- and tabulates reward / highest tile / moves / illegal moves per game.  Here the games are the boards of
ONE ``Batched2048`` in ``rng="numpy"`` mode (board i is seeded like the reference's i-th ``reset``) stepped in
lockstep without auto-reset; the policy sees the whole batch of observations at once.  With the same policy the
result dict and the CSV equal the reference's, row for row (``tests/golden/eval_table.npz``).
"""
from __future__ import annotations

import csv
import random
from typing import Callable, Optional

import numpy as np

MAX_MOVES = 2000          # train.py:159 (`if moves_taken > 2000: break`, i.e. at most 2 001 moves)


def _to_numpy(x):
    if isinstance(x, np.ndarray):
        return x
    return x.detach().cpu().numpy()


def evaluate_model(policy: Callable, episodes: int, epsilon: float = 0.0, *, engine=None, device: int = 0,
                   env_seed: int = 456, agent_seed: Optional[int] = 123, illegal_move_reward: float = -1.0,
                   obs_dtype=None, verbose: bool = False) -> dict:
    """``train.py:168-213`` for all episodes at once.

    ``policy(obs)``: ``obs`` is the batch of one-hot observations ``(episodes, 16, 4, 4)`` (a device tensor of
    ``obs_dtype``, default float32, from ``Batched2048.observe_onehot``); it returns per-action scores
    ``(episodes, 4)`` -- the batched form of ``predict`` (``train.py:84-97``) -- or the chosen actions
    ``(episodes,)``.  ``epsilon`` / ``agent_seed`` reproduce ``choose_action`` (``train.py:100-117``): episode i
    draws from ``random.Random(agent_seed + i)`` exactly as the reference's re-seeded global generator does
    (``agent_seed=None``: unseeded generators, as ``train.py:143-144``).

    Returns the reference's dict: ``'Average score'``, ``'Max score'``, ``'Highest tile'``, ``'Episodes'`` (a list
    of ``{'total_reward', 'highest', 'moves', 'illegal_moves'}``).
    """
    n = int(episodes)
    if n <= 0:
        raise ValueError("episodes must be positive")
    own = engine is None
    if own:
        import torch
        from .batched import Batched2048
        engine = Batched2048(n, device=device, seed=env_seed, rng="numpy")       # board i <- reset(seed=env_seed + i)
        if obs_dtype is None:
            obs_dtype = torch.float32
    engine.set_illegal_move_reward(illegal_move_reward)                            # train.py:184
    engine.reset()
    rngs = [random.Random(None if agent_seed is None else agent_seed + i) for i in range(n)]   # train.py:141-144

    total = np.zeros(n, np.float64)
    moves = np.zeros(n, np.int64)
    illegals = np.zeros(n, np.int64)
    highest = np.zeros(n, np.int64)
    active = np.ones(n, bool)
    while active.any():
        obs = engine.observe_onehot(obs_dtype) if obs_dtype is not None else engine.observe_onehot()
        out = _to_numpy(policy(obs))
        greedy = out.argmax(axis=1) if out.ndim == 2 else out                      # train.py:115
        actions = np.asarray(greedy, dtype=np.int64).copy()
        for i in np.flatnonzero(active):                                           # train.py:114-117, per episode
            if not (rngs[i].uniform(0, 1) > epsilon):
                actions[i] = rngs[i].randint(0, 3)
        step = engine.step_numpy(actions.astype(np.uint8), auto_reset=False)       # train.py:152
        rew, term = np.asarray(step["reward"], np.float64), np.asarray(step["terminated"], bool)
        total[active] += rew[active]                                               # :154
        illegals[active] += np.asarray(step["illegal"], bool)[active]              # :155-156
        moves[active] += 1                                                         # :157
        highest[active] = np.asarray(step["highest"], np.int64)[active]            # :165 (the last step's info)
        active &= ~(moves > MAX_MOVES)                                             # :158-159
        active &= ~term                                                            # :161-162
    if own:
        engine.close()
    scores = [{"total_reward": float(total[i]), "highest": int(1 << highest[i]) if highest[i] else 0,
               "moves": int(moves[i]), "illegal_moves": int(illegals[i])} for i in range(n)]
    if verbose:
        for i, s in enumerate(scores):                                             # train.py:191-194
            print(f"Episode {i}, epsilon {epsilon}, highest {s['highest']}, reward {s['total_reward']:.1f}, "
                  f"moves {s['moves']}, illegals {s['illegal_moves']}")
    return {"Average score": sum(s["total_reward"] for s in scores) / n,          # train.py:203-212
            "Max score": max(s["total_reward"] for s in scores),
            "Highest tile": max(s["highest"] for s in scores),
            "Episodes": scores}


def report_evaluation_results(results: dict, label: str = "eval", path: Optional[str] = None) -> str:
    """``train.py:216-229``: ``scores_{label}.csv`` with the reference's columns; returns the file name."""
    name = path or f"scores_{label}.csv"
    with open(name, "w") as f:
        writer = csv.DictWriter(f, fieldnames=["total_reward", "highest", "moves", "illegal_moves"], lineterminator="\n")
        writer.writeheader()
        for s in results["Episodes"]:
            writer.writerow(s)
    return name
