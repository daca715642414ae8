"""Transitions recorded from batched rollouts, in the reference's on-disk format.

The data format either side of the step path: ``training_data.py`` of the reference stores
``(board, action, reward, next_board, done)`` rows and exchanges them as a 35-column CSV
(``training_data.py:188-248``; header ``:212-225``; row format ``'%d,' * 17 + '%f,' + '%d,' * 16 + '%i'``
``:244``).  ``pretrain_bc.py`` / ``train.py`` read those files.  This module lets GPU rollouts produce
them, and runs the reference's ``augment()`` (``:282-299``) as one device kernel.

Boards cross this interface as int tile values (the reference's convention); on the device they
are uint8 exponents.
"""
from __future__ import annotations

import io

import numpy as np

from .batched import exp_to_values, values_to_exp


def unstack(stacked, layers=15):
    """gather_training_data.py:71-75: (16,4,4) one-hot -> (4,4) tile values."""
    representation = 2 ** (np.arange(layers, dtype=int) + 1)
    return np.sum(np.asarray(stacked)[1:] * representation[:, np.newaxis, np.newaxis], axis=0)


def csv_header(add_returns=False):
    """training_data.py:212-225."""
    cols = [f"{m}-{n}" for m in range(1, 5) for n in range(1, 5)] + ["action", "reward"]
    cols += [f"next {m}-{n}" for m in range(1, 5) for n in range(1, 5)] + ["done"]
    if add_returns:
        cols.append("return")
    return cols


class Transitions:
    """(board, action, reward, next_board, done) rows; arrays shaped as ``training_data`` keeps them."""

    def __init__(self, x=None, action=None, reward=None, next_x=None, done=None):
        self.x = np.zeros((0, 4, 4), int) if x is None else np.asarray(x, dtype=int).reshape(-1, 4, 4)
        n = len(self.x)
        self.action = np.zeros((0, 1), int) if action is None else np.asarray(action, dtype=int).reshape(n, 1)
        self.reward = np.zeros((0, 1), float) if reward is None else np.asarray(reward, dtype=float).reshape(n, 1)
        self.next_x = np.zeros((0, 4, 4), int) if next_x is None else np.asarray(next_x, dtype=int).reshape(n, 4, 4)
        self.done = np.zeros((0, 1), bool) if done is None else np.asarray(done, dtype=bool).reshape(n, 1)

    def size(self):
        return len(self.x)

    # ------------------------------------------------------------------ recording from the engine
    @classmethod
    def record(cls, engine, actions, n_steps=None):
        """Play ``actions`` (``[k, n]`` tensor/array, or ``None`` with ``n_steps`` for the synthetic
        policy) on a ``Batched2048`` and record every transition, env-major (env 0's k steps first,
        i.e. "game order" as ``get_discounted_return`` expects, training_data.py:104-124).

        ``next_board`` of a step that ends an episode is the terminal board (what the reference's
        recorder stores, gather_training_data.py:191-196), not the auto-reset board."""
        import torch
        k = n_steps if actions is None else len(actions)
        n = engine.n_envs
        xs = torch.empty((k, n, 16), dtype=torch.uint8, device=engine.device)
        nxt = torch.empty_like(xs)
        acts = torch.empty((k, n), dtype=torch.uint8, device=engine.device)
        rew = torch.empty((k, n), dtype=torch.float32, device=engine.device)
        done = torch.empty((k, n), dtype=torch.uint8, device=engine.device)
        if actions is None:
            acts.copy_(engine.random_actions(k))
        else:
            acts.copy_(torch.as_tensor(np.asarray(actions) if not isinstance(actions, torch.Tensor) else actions))
        for j in range(k):
            xs[j].copy_(engine.boards().reshape(n, 16))
            engine.step(acts[j], auto_reset=True, want_info=True)
            rew[j].copy_(engine.reward)
            done[j].copy_(engine.terminated)
            nxt[j].copy_(torch.where(engine.terminated.bool().unsqueeze(1), engine.terminal_boards,
                                     engine.boards().reshape(n, 16)))
        order = lambda t: t.transpose(0, 1).contiguous().cpu().numpy()  # noqa: E731  env-major
        return cls(exp_to_values(order(xs)).reshape(-1, 4, 4), order(acts).reshape(-1),
                   order(rew).reshape(-1), exp_to_values(order(nxt)).reshape(-1, 4, 4), order(done).reshape(-1))

    # ------------------------------------------------------------------ returns
    def discounted_return(self, gamma=0.9):
        """training_data.py:104-124 (rows in game order; ``done`` ends an episode)."""
        out = np.zeros(self.size())
        previous = None
        for i in range(self.size() - 1, -1, -1):
            smoothed = self.reward[i, 0]
            if self.done[i, 0]:
                previous = None
            if previous:
                smoothed += gamma * previous
            out[i] = smoothed
            previous = smoothed
        return out.reshape(-1, 1)

    # ------------------------------------------------------------------ CSV, the reference's format
    def export_csv(self, filename, add_returns=False):
        """training_data.py:227-248, byte-identical output."""
        n = self.size()
        cols = [self.x.reshape(n, 16), self.action, self.reward, self.next_x.reshape(n, 16), self.done]
        fmt = "%d," * 17 + "%f," + "%d," * 16 + "%i"
        if add_returns:
            cols.append(self.discounted_return())
            fmt += ",%f"
        data = np.concatenate([c.astype(float) for c in cols], axis=1)
        np.savetxt(filename, data, comments="", fmt=fmt, header=",".join(csv_header(add_returns)))

    def to_csv_text(self, add_returns=False) -> str:
        buf = io.StringIO()
        self.export_csv(buf, add_returns)
        return buf.getvalue()

    @classmethod
    def import_csv(cls, filename):
        """training_data.py:188-210."""
        raw = np.loadtxt(filename, delimiter=",", skiprows=1, ndmin=2)
        return cls(raw[:, :16].astype(int), raw[:, 16].astype(int), raw[:, 17], raw[:, 18:34].astype(int),
                   raw[:, 34].astype(bool))

    # ------------------------------------------------------------------ augmentation on the device
    def augment(self, device=0):
        """training_data.py:282-299: + hflip, then + 3 rotations of both, 8x the rows, reference order.
        Runs as one launch of ``augment_kernel`` (g2048_augment)."""
        import ctypes as C

        import torch

        from . import _lib
        lib = _lib.load()
        n = self.size()
        dev = torch.device("cuda", device)
        b = torch.as_tensor(values_to_exp(self.x).reshape(n, 16)).to(dev)
        nb = torch.as_tensor(values_to_exp(self.next_x).reshape(n, 16)).to(dev)
        a = torch.as_tensor(self.action.reshape(n).astype(np.uint8)).to(dev)
        bo = torch.empty((8 * n, 16), dtype=torch.uint8, device=dev)
        no = torch.empty((8 * n, 16), dtype=torch.uint8, device=dev)
        ao = torch.empty(8 * n, dtype=torch.uint8, device=dev)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.g2048_augment(b.data_ptr(), nb.data_ptr(), a.data_ptr(), n, bo.data_ptr(), no.data_ptr(),
                                     ao.data_ptr(), stream))
        return Transitions(exp_to_values(bo.cpu().numpy()).reshape(-1, 4, 4), ao.cpu().numpy(),
                           np.tile(self.reward.reshape(-1), 8), exp_to_values(no.cpu().numpy()).reshape(-1, 4, 4),
                           np.tile(self.done.reshape(-1), 8))

    # ------------------------------------------------------------------ rewards from (board, action)
    def recompute_rewards(self, device=0, illegal_move_reward=0.0):
        """add_rewards_to_training_data.py:55-68 for every row at once: the reward ``env.step(action)`` gives from
        ``env.set_board(board)`` -- the merge score of the move, or ``illegal_move_reward`` where the move is
        illegal (game2048_env.py:85-95).  One ``g2048_move(trial)`` launch over all rows; returns a new
        ``Transitions`` with the rewards filled in (boards, actions, next boards, done flags unchanged)."""
        import torch

        from .batched import Batched2048
        n = self.size()
        eng = Batched2048(n, device=device)
        try:
            eng.set_boards(values_to_exp(self.x).reshape(n, 16))
            score, legal = eng.move(torch.as_tensor(self.action.reshape(n).astype(np.uint8)), trial=True)
            reward = np.where(legal.cpu().numpy().astype(bool), score.cpu().numpy().astype(float), float(illegal_move_reward))
        finally:
            eng.close()
        return Transitions(self.x, self.action, reward, self.next_x, self.done)

    # ------------------------------------------------------------------ canonicalisation on the device
    def canonicalize(self, device=0):
        """Symmetric-board canonicalisation (the inverse view of ``augment``): every row is
        replaced by the lexicographically smallest (row-major board) of its eight symmetries -- the ones
        training_data.py:257-299 generates -- with the action remapped and ``next_x`` turned the same way.
        One launch of ``canonicalize_kernel`` (g2048_canonicalize).  Returns ``(Transitions, symmetry)`` where
        ``symmetry[i] = 2 * clockwise_quarter_turns + hflip`` is the variant that was applied."""
        b, nb, a, sym = canonicalize_device(values_to_exp(self.x).reshape(-1, 16), self.action.reshape(-1),
                                            values_to_exp(self.next_x).reshape(-1, 16), device)  # noqa: E501
        return (Transitions(exp_to_values(b).reshape(-1, 4, 4), a, self.reward.reshape(-1),
                            exp_to_values(nb).reshape(-1, 4, 4), self.done.reshape(-1)), sym)


def canonicalize_device(boards_exp, actions=None, next_boards_exp=None, device=0):
    """uint8 exponent boards ``[n, 16]`` (+ optional actions ``[n]`` and next boards) -> canonical form, on
    ``cuda:device`` through ``g2048_canonicalize``.  Returns host arrays ``(boards, next_boards | None,
    actions | None, symmetry)``."""
    import ctypes as C

    import torch

    from . import _lib
    lib = _lib.load()
    dev = torch.device("cuda", device)
    b = torch.as_tensor(np.ascontiguousarray(boards_exp, dtype=np.uint8).reshape(-1, 16)).to(dev)
    n = b.shape[0]
    nb = None if next_boards_exp is None else torch.as_tensor(
        np.ascontiguousarray(next_boards_exp, dtype=np.uint8).reshape(n, 16)).to(dev)
    a = None if actions is None else torch.as_tensor(np.ascontiguousarray(actions).reshape(n).astype(np.uint8)).to(dev)
    sym = torch.empty(n, dtype=torch.uint8, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(lib.g2048_canonicalize(b.data_ptr(), None if nb is None else nb.data_ptr(),
                                      None if a is None else a.data_ptr(), n, sym.data_ptr(), stream))
    return (b.cpu().numpy(), None if nb is None else nb.cpu().numpy(), None if a is None else a.cpu().numpy(),
            sym.cpu().numpy())
