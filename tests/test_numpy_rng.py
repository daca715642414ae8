"""CPU: the numpy-compatible RNG restatement (oracle/numpy_rng.py and the C oracle) against numpy itself."""
import numpy as np
import pytest

from oracle.numpy_rng import NumpyPCG64, TWO_THRESHOLD_53, add_tile_numpy


@pytest.mark.parametrize("seed", [0, 1, 42, 2 ** 31 - 1, 123456789012345])
def test_stream_matches_numpy_generator(seed):
    gen = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
    mine = NumpyPCG64.from_seed(seed)
    rng = np.random.default_rng(seed + 1)
    for _ in range(300):
        kind = rng.integers(0, 3)
        if kind == 0:
            assert gen.random() == mine.random()
        elif kind == 1:
            n = int(rng.integers(2, 20))
            a, b = list(range(n)), list(range(n))
            gen.shuffle(a)
            mine.shuffle(b)
            assert a == b
        else:
            a = [(r, c) for r in range(4) for c in range(4)]      # the reference's _all_positions
            b = list(a)
            gen.shuffle(a)
            mine.shuffle(b)
            assert a == b
    st = gen.bit_generator.state
    assert (st["state"]["state"], st["state"]["inc"], st["has_uint32"], st["uinteger"]) == \
           (mine.state, mine.inc, mine.has_uint32, mine.uinteger)


def test_threshold_is_exact():
    for k in (TWO_THRESHOLD_53 - 2, TWO_THRESHOLD_53 - 1, TWO_THRESHOLD_53, TWO_THRESHOLD_53 + 1):
        assert (k * (1.0 / 9007199254740992.0) < 0.9) == (k < TWO_THRESHOLD_53)


def test_add_tile_consumes_like_the_reference_would():
    gen = np.random.Generator(np.random.PCG64(np.random.SeedSequence(7)))
    mine = NumpyPCG64.from_seed(7)
    M = [0] * 16
    for _ in range(12):
        # the reference's add_tile, written out (game2048_env.py:166-176)
        val = 2 if gen.random() < 0.9 else 4
        pos = [(r, c) for r in range(4) for c in range(4)]
        gen.shuffle(pos)
        want = list(M)
        for r, c in pos:
            if want[r * 4 + c] == 0:
                want[r * 4 + c] = val
                break
        add_tile_numpy(M, mine)
        assert M == want


# ------------------------------------------------------------------ C oracle and device header

import ctypes as C  # noqa: E402

from conftest import load_golden  # noqa: E402
from oracle import OracleBatch, pcg64_states_from_seeds  # noqa: E402

NUMPY_TRAJ = ["traj_numpy_seed42", "traj_numpy_seed7_irw"]


def test_c_oracle_pcg64_matches_numpy(oracle_lib):
    for seed in (0, 42, 99991):
        st = pcg64_states_from_seeds([seed])
        gen = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
        mine = NumpyPCG64.from_seed(seed)
        for k in range(200):
            if k % 3 == 0:
                assert oracle_lib.g2048o_pcg64_next64(st.ctypes.data) == mine.next64()
            elif k % 3 == 1:
                assert oracle_lib.g2048o_pcg64_next32(st.ctypes.data) == mine.next32()
            else:
                mx = 1 + k % 15
                assert oracle_lib.g2048o_pcg64_interval(st.ctypes.data, mx) == mine.interval(mx)
        # and the Python restatement is numpy (checked above), so close the triangle on the raw stream
        g2 = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
        st2 = pcg64_states_from_seeds([seed])
        for _ in range(20):
            assert oracle_lib.g2048o_pcg64_next64(st2.ctypes.data) == int(g2.bit_generator.random_raw())


class NumpyModeBatch(OracleBatch):
    """OracleBatch driven in numpy-RNG mode; `impl` = oracle lib or the host-compiled device header."""

    def __init__(self, n, seed, impl=None, prefix="g2048o"):
        super().__init__(n, seed)
        self.impl, self.prefix = impl or self.lib, prefix
        self.seed_numpy(seed)

    def reset(self):
        b = self._batch()
        getattr(self.impl, self.prefix + "_reset_batch_numpy")(C.byref(b), C.c_void_p(self.rng.ctypes.data),
                                                              C.c_uint64(self.n), C.c_uint64(self.t), 1)

    def step(self, actions=None, auto_reset=True):
        if actions is not None:
            actions = np.ascontiguousarray(actions, dtype=np.uint8)
        self.t += 1
        b = self._batch(actions)
        getattr(self.impl, self.prefix + "_step_batch_numpy")(
            C.byref(b), C.c_void_p(self.rng.ctypes.data), C.c_uint64(self.n), C.c_uint64(self.seed),
            C.c_uint64(self.t), C.c_uint64(self.board_offset), C.c_float(self.illegal_move_reward),
            self.max_exp, int(auto_reset), 1)


def replay_numpy(d, make):
    seed, _, n, steps, _, _ = (int(x) for x in d["meta"])
    b = make(n, seed)
    b.illegal_move_reward = float(d["illegal_move_reward"][0])
    b.reset()
    assert np.array_equal(b.boards, d["initial_boards"])
    for s in range(steps):
        b.step(d["actions"][:, s])
        for f in ("reward", "terminated", "illegal", "highest", "boards", "score"):
            assert np.array_equal(getattr(b, f), d[f][:, s]), (f, s)
        done = d["terminated"][:, s].astype(bool)
        assert np.array_equal(b.terminal_boards[done], d["terminal_boards"][done, s])
    return b


@pytest.mark.parametrize("name", NUMPY_TRAJ)
def test_c_oracle_replays_reference_with_its_own_rng(name):
    replay_numpy(load_golden(name), lambda n, seed: NumpyModeBatch(n, seed))


@pytest.mark.parametrize("name", NUMPY_TRAJ)
def test_device_header_replays_reference_with_its_own_rng(host_check, name):
    replay_numpy(load_golden(name), lambda n, seed: NumpyModeBatch(n, seed, host_check, "hostcheck"))


def test_device_header_pcg64_and_rollout_vs_oracle(host_check, oracle_lib):
    host_check.hostcheck_pcg64_next64.restype = C.c_uint64
    a, b = pcg64_states_from_seeds([5]), pcg64_states_from_seeds([5])
    for k in range(300):
        if k % 2:
            assert host_check.hostcheck_pcg64_next64(C.c_void_p(a.ctypes.data)) == \
                   oracle_lib.g2048o_pcg64_next64(b.ctypes.data)
        else:
            mx = 1 + k % 15
            assert host_check.hostcheck_pcg64_interval(C.c_void_p(a.ctypes.data), mx) == \
                   oracle_lib.g2048o_pcg64_interval(b.ctypes.data, mx)
    assert np.array_equal(a, b)
    rng = np.random.default_rng(0)
    for _ in range(200):
        board = (rng.integers(0, 3, 16) * rng.integers(0, 2, 16)).astype(np.uint8)
        want = sum(1 << i for i in range(16) if board[i] == 0)
        assert host_check.hostcheck_empty_mask16(board.ctypes.data_as(C.POINTER(C.c_uint8))) == want
    x, y = NumpyModeBatch(1024, 3), NumpyModeBatch(1024, 3, host_check, "hostcheck")
    x.reset()
    y.reset()
    for s in range(64):
        x.step(None)
        y.step(None)
        for f in ("boards", "score", "reward", "terminated", "illegal", "highest", "rng"):
            assert np.array_equal(getattr(x, f), getattr(y, f)), (f, s)


def test_device_seeding_matches_numpy_seedsequence(host_check):
    """pcg64_from_seed (the device header's SeedSequence + PCG64 seeding) == numpy, incl. seeds >= 2^32."""
    seeds = list(range(0, 300)) + [2 ** 31 - 1, 2 ** 32 - 1, 2 ** 32, 2 ** 32 + 5, 2 ** 40 + 12345, 2 ** 63 + 99,
                                   2 ** 64 - 1] + [int(x) for x in np.random.default_rng(1).integers(0, 2 ** 62, 200)]
    want = pcg64_states_from_seeds(seeds)
    got = np.zeros(5, np.uint64)
    for s, w in zip(seeds, want):
        host_check.hostcheck_pcg64_from_seed(C.c_uint64(s), C.c_void_p(got.ctypes.data))
        assert np.array_equal(got, w), s
