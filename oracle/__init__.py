"""CPU ORACLE package -- TEST INFRASTRUCTURE ONLY (see g2048_oracle.h / cpu_ref.py).

``load()`` returns a ctypes handle on ``libg2048_oracle.so`` (built by ``make -C oracle`` or by
``__graft_entry__.build()``), with argtypes set.  ``OracleBatch`` is a numpy-backed driver with
the same state/outputs as the device engine, used by the parity tests as the checker.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# G2048_ORACLE_LIB: an alternative build of the same sources (the sanitizer build of `make -C oracle asan`)
_SO = os.environ.get("G2048_ORACLE_LIB") or os.path.join(_HERE, "libg2048_oracle.so")
_lib = None


class _Batch(C.Structure):
    _fields_ = [
        ("boards", C.c_void_p), ("score", C.c_void_p), ("ep_start", C.c_void_p),
        ("actions", C.c_void_p),
        ("reward", C.c_void_p), ("terminated", C.c_void_p), ("illegal", C.c_void_p),
        ("highest", C.c_void_p), ("terminal_boards", C.c_void_p), ("last_score", C.c_void_p),
        ("last_len", C.c_void_p), ("ep_count", C.c_void_p),
    ]


def build(force: bool = False) -> str:
    if os.environ.get("G2048_ORACLE_LIB"):
        return _SO                                   # a pre-built alternative library: never rebuilt here
    src = [os.path.join(_HERE, f) for f in ("g2048_oracle.c", "g2048_oracle.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libg2048_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    lib = C.CDLL(_SO)
    u32, u64, i64p = C.c_uint32, C.c_uint64, C.POINTER(C.c_int64)
    lib.g2048o_philox4x32_10.argtypes = [C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    lib.g2048o_spawn_word.argtypes = [u64, u64, u32, u32]
    lib.g2048o_spawn_word.restype = u32
    lib.g2048o_spawn_value.argtypes = [u32, u32]
    lib.g2048o_spawn_value.restype = C.c_int64
    lib.g2048o_random_action.argtypes = [u64, u64, u32]
    lib.g2048o_random_action.restype = C.c_uint8
    lib.g2048o_shift.argtypes = [i64p, i64p]
    lib.g2048o_shift.restype = C.c_int64
    lib.g2048o_move.argtypes = [i64p, C.c_int, C.c_int, i64p]
    lib.g2048o_move.restype = C.c_int
    lib.g2048o_highest.argtypes = [i64p]
    lib.g2048o_highest.restype = C.c_int64
    lib.g2048o_isend.argtypes = [i64p, C.c_int64]
    lib.g2048o_isend.restype = C.c_int
    lib.g2048o_add_tile.argtypes = [i64p, u32]
    lib.g2048o_add_tile.restype = C.c_int
    lib.g2048o_stack.argtypes = [i64p, i64p]
    lib.g2048o_reset_batch.argtypes = [C.POINTER(_Batch), u64, u64, u64, u64, u32, C.c_int]
    lib.g2048o_step_batch.argtypes = [C.POINTER(_Batch), u64, u64, u64, u64, C.c_float, C.c_int, C.c_int, C.c_int]
    lib.g2048o_onehot_batch.argtypes = [C.c_void_p, u64, C.c_void_p]
    lib.g2048o_pcg64_next64.argtypes = [C.c_void_p]
    lib.g2048o_pcg64_next64.restype = u64
    lib.g2048o_pcg64_next32.argtypes = [C.c_void_p]
    lib.g2048o_pcg64_next32.restype = u32
    lib.g2048o_pcg64_interval.argtypes = [C.c_void_p, u32]
    lib.g2048o_pcg64_interval.restype = u32
    lib.g2048o_add_tile_numpy.argtypes = [i64p, C.c_void_p]
    lib.g2048o_add_tile_numpy.restype = C.c_int
    lib.g2048o_reset_batch_numpy.argtypes = [C.POINTER(_Batch), C.c_void_p, u64, u64, C.c_int]
    lib.g2048o_step_batch_numpy.argtypes = [C.POINTER(_Batch), C.c_void_p, u64, u64, u64, u64, C.c_float, C.c_int,
                                            C.c_int, C.c_int]
    _lib = lib
    return lib


def _ptr(a):
    return None if a is None else a.ctypes.data


class OracleBatch:
    """N boards stepped by the C oracle, in the device data layout (the parity checker)."""

    def __init__(self, n: int, seed: int = 0, board_offset: int = 0, threads: int = 1):
        self.lib = load()
        self.n, self.seed, self.board_offset, self.threads = n, seed, board_offset, threads
        self.t = 0
        self.fresh = True
        self.illegal_move_reward = 0.0
        self.max_exp = 0
        self.boards = np.zeros((n, 16), np.uint8)
        self.score = np.zeros(n, np.int32)
        self.ep_start = np.zeros(n, np.uint32)
        self.reward = np.zeros(n, np.float32)
        self.terminated = np.zeros(n, np.uint8)
        self.illegal = np.zeros(n, np.uint8)
        self.highest = np.zeros(n, np.uint8)
        self.terminal_boards = np.zeros((n, 16), np.uint8)
        self.last_score = np.zeros(n, np.int32)
        self.last_len = np.zeros(n, np.int32)
        self.ep_count = np.zeros(n, np.uint32)
        # ---- return accounting: the checker's side of g2048_stats.return_sum, kept two independent ways
        # (a) finished_return_sum: the final score of every episode, added at the step that ends it (what SB3's Monitor
        #     sums as info["episode"]["r"], ppo_train.py:123, minus illegal-move rewards)
        # (b) the conservation form the device uses: every merge score ever made is either on a live board or belongs to
        #     a finished episode -> return_sum = gain_total - sum(score of boards whose episode is still running);
        #     `pending` = episode ended but the board was not reset (auto_reset off): no longer "running"
        # With auto_reset always on the two agree; (b) also defines the edge cases (masked resets abandon running
        # episodes, a board stepped again after its episode ended continues that episode).
        self.finished_return_sum = 0
        self.gain_total = 0
        self.pending = np.zeros(n, bool)

    @property
    def return_sum(self) -> int:
        return self.gain_total - int(self.score[~self.pending].astype(np.int64).sum())

    def _clear_return_accounting(self):
        """g2048_seed: the statistics restart; scores already on the boards are not gains of the new books."""
        self.pending[:] = False
        self.gain_total = int(self.score.astype(np.int64).sum())
        self.finished_return_sum = 0

    def _account_reset(self, sel):
        """Boards ``sel`` are about to be reset: a running episode is abandoned (its score leaves the books), an ended
        one keeps its place among the finished."""
        self.gain_total -= int(self.score[sel & ~self.pending].astype(np.int64).sum())
        self.pending[sel] = False

    def _account_step(self, auto_reset: bool):
        done = self.terminated.astype(bool)
        legal = self.illegal == 0
        self.gain_total += int(self.reward[legal].astype(np.int64).sum())          # :86 (an illegal move scores nothing)
        self.finished_return_sum += int(self.last_score[done].astype(np.int64).sum())
        self.pending[:] = False if auto_reset else done      # every board was stepped: older marks are gone

    def set_scores(self, scores):
        """g2048_set_scores: the score of a running episode moves; a board whose episode has ended keeps its place."""
        scores = np.asarray(scores, dtype=np.int32)
        live = ~self.pending
        self.gain_total += int(scores[live].astype(np.int64).sum()) - int(self.score[live].astype(np.int64).sum())
        self.score[:] = scores

    def _batch(self, actions=None):
        return _Batch(_ptr(self.boards), _ptr(self.score), _ptr(self.ep_start), _ptr(actions),
                      _ptr(self.reward), _ptr(self.terminated), _ptr(self.illegal), _ptr(self.highest),
                      _ptr(self.terminal_boards), _ptr(self.last_score), _ptr(self.last_len),
                      _ptr(self.ep_count))

    def seed_(self, seed: int):
        self.seed, self.t, self.fresh = seed, 0, True
        self._clear_return_accounting()

    def reset(self, first_slot: int = 0, new_transaction=None, mask=None):
        """``mask``: uint8[n], only boards with mask != 0 are reset (g2048_reset's mask)."""
        if new_transaction is None:
            new_transaction = not self.fresh
        if new_transaction:
            self.t += 1
        self.fresh = False
        sel = np.ones(self.n, bool) if mask is None else np.asarray(mask) != 0
        self._account_reset(sel)
        keep = (self.boards.copy(), self.score.copy(), self.ep_start.copy()) if mask is not None else None
        b = self._batch()
        self.lib.g2048o_reset_batch(C.byref(b), self.n, self.seed, self.t, self.board_offset, first_slot,
                                    self.threads)
        if keep is not None:
            self.boards[~sel], self.score[~sel], self.ep_start[~sel] = keep[0][~sel], keep[1][~sel], keep[2][~sel]

    def step(self, actions=None, auto_reset: bool = True):
        """actions: uint8[n] or None for the synthetic random policy."""
        if actions is not None:
            actions = np.ascontiguousarray(actions, dtype=np.uint8)
            assert actions.shape == (self.n,)
        self.t += 1
        self.fresh = False
        b = self._batch(actions)
        self.lib.g2048o_step_batch(C.byref(b), self.n, self.seed, self.t, self.board_offset,
                                   self.illegal_move_reward, self.max_exp, int(auto_reset), self.threads)
        self._account_step(auto_reset)

    # -- numpy-compatible RNG mode: rng[i] = (state_lo, state_hi, inc_lo, inc_hi, buf) of board i
    def seed_numpy(self, seed: int):
        """Board i gets numpy's PCG64(SeedSequence(seed + board_offset + i)) -- what gymnasium gives
        env i of a vector env seeded with ``seed`` (SB3 seeds env i with seed + i)."""
        self.rng = pcg64_states_from_seeds(seed + self.board_offset + np.arange(self.n))
        self.seed, self.t, self.fresh = seed, 0, True
        self._clear_return_accounting()

    def reset_numpy(self):
        b = self._batch()
        self.fresh = False
        self._account_reset(np.ones(self.n, bool))
        self.lib.g2048o_reset_batch_numpy(C.byref(b), self.rng.ctypes.data, self.n, self.t, self.threads)

    def step_numpy(self, actions=None, auto_reset: bool = True):
        if actions is not None:
            actions = np.ascontiguousarray(actions, dtype=np.uint8)
        self.t += 1
        b = self._batch(actions)
        self.lib.g2048o_step_batch_numpy(C.byref(b), self.rng.ctypes.data, self.n, self.seed, self.t,
                                         self.board_offset, self.illegal_move_reward, self.max_exp,
                                         int(auto_reset), self.threads)
        self._account_step(auto_reset)

    def onehot(self):
        out = np.zeros((self.n, 16, 4, 4), np.uint8)
        self.lib.g2048o_onehot_batch(_ptr(self.boards), self.n, _ptr(out))
        return out


def pcg64_states_from_seeds(seeds) -> np.ndarray:
    """uint64 [n, 5] = (state_lo, state_hi, inc_lo, inc_hi, buf) of numpy's PCG64(SeedSequence(seed)),
    the generator gymnasium builds for ``reset(seed=seed)``.  The SeedSequence hashing is numpy's own."""
    m64 = (1 << 64) - 1
    out = np.zeros((len(seeds), 5), np.uint64)
    for i, s in enumerate(seeds):
        st = np.random.PCG64(np.random.SeedSequence(int(s))).state
        state, inc = st["state"]["state"], st["state"]["inc"]
        out[i] = (state & m64, state >> 64, inc & m64, inc >> 64, st["uinteger"] | (st["has_uint32"] << 32))
    return out
