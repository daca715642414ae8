"""CPU: the byte-parallel board arithmetic of gym-2048_amd/csrc/g2048_device.h -- the exact header
the gfx950 kernels are compiled from -- built for the host by tests/host_check and compared with the
golden vectors and the oracle.  Catches logic errors here, where there is no GPU; the GPU tests then
prove the same on the device."""
import ctypes as C
import itertools

import numpy as np
import pytest

from conftest import TRAJECTORIES, load_golden, replay_trajectory
from oracle import OracleBatch, _Batch

U8P = C.POINTER(C.c_uint8)


class HostCheckBatch(OracleBatch):
    """OracleBatch-shaped driver whose reset/step run the device header's code."""

    def __init__(self, hc, n, seed=0, board_offset=0):
        super().__init__(n, seed, board_offset)
        self.hc = hc
        for f in (hc.hostcheck_reset_batch, hc.hostcheck_step_batch):
            f.restype = None
        hc.hostcheck_reset_batch.argtypes = self.lib.g2048o_reset_batch.argtypes
        hc.hostcheck_step_batch.argtypes = self.lib.g2048o_step_batch.argtypes

    def reset(self, first_slot=0, new_transaction=None):
        if new_transaction is None:
            new_transaction = not self.fresh
        if new_transaction:
            self.t += 1
        self.fresh = False
        b = self._batch()
        self.hc.hostcheck_reset_batch(C.byref(b), self.n, self.seed, self.t, self.board_offset, first_slot, 1)

    def step(self, actions=None, auto_reset=True):
        if actions is not None:
            actions = np.ascontiguousarray(actions, dtype=np.uint8)
        self.t += 1
        self.fresh = False
        b = self._batch(actions)
        self.hc.hostcheck_step_batch(C.byref(b), self.n, self.seed, self.t, self.board_offset,
                                     self.illegal_move_reward, self.max_exp, int(auto_reset), 1)


def test_philox(host_check):
    out = (C.c_uint32 * 4)()
    host_check.hostcheck_philox((C.c_uint32 * 4)(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344),
                                (C.c_uint32 * 2)(0xa4093822, 0x299f31d0), out)
    assert tuple(out) == (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)


def test_shift4_exhaustive(host_check):
    g = load_golden("shift_exhaustive")
    rng = np.random.default_rng(1)
    junk_all = rng.integers(0, 18, size=(18 ** 4, 12)).astype(np.uint8)
    for i, exps in enumerate(itertools.product(range(18), repeat=4)):
        out = (C.c_uint8 * 4)()
        sc = host_check.hostcheck_shift((C.c_uint8 * 4)(*exps), out, i % 4, junk_all[i].ctypes.data_as(U8P))
        assert list(out) == list(g["out"][i]) and sc == g["score"][i], exps


def test_move_query_tables(host_check):
    m = load_golden("move_table")
    for i in range(len(m["boards"])):
        b = np.ascontiguousarray(m["boards"][i])
        p = b.ctypes.data_as(U8P)
        assert host_check.hostcheck_highest(p) == m["highest"][i]
        assert host_check.hostcheck_count_empty(p) == int((b == 0).sum())
        for d in range(4):
            out = np.zeros(16, np.uint8)
            sc = C.c_uint32()
            legal = host_check.hostcheck_move(p, d, out.ctypes.data_as(U8P), C.byref(sc))
            assert legal == m["legal"][i, d]
            assert np.array_equal(out, m["new"][i, d]) and sc.value == m["score"][i, d]
    t = load_golden("isend_table")
    full = [(b, e) for b, e in zip(t["boards"], t["isend"]) if (b != 0).all()]
    for b, _ in full:
        b = np.ascontiguousarray(b)
        v = b.reshape(4, 4)
        want = bool((v[:, :-1] == v[:, 1:]).any() or (v[:-1] == v[1:]).any())
        assert bool(host_check.hostcheck_has_equal_neighbours(b.ctypes.data_as(U8P))) == want


def test_add_tile_matches_oracle(host_check, oracle_lib):
    rng = np.random.default_rng(3)
    I64x16 = C.c_int64 * 16
    for _ in range(4000):
        fill = rng.integers(0, 16)
        b = np.zeros(16, np.uint8)
        b[rng.choice(16, size=fill, replace=False)] = rng.integers(1, 12, size=fill)
        w = int(rng.integers(0, 2 ** 32))
        M = I64x16(*[0 if e == 0 else 1 << int(e) for e in b])
        oracle_lib.g2048o_add_tile(M, w)
        want = [0 if v == 0 else int(v).bit_length() - 1 for v in M]
        got = b.copy()
        host_check.hostcheck_add_tile(got.ctypes.data_as(U8P), w)
        assert list(got) == want


@pytest.mark.parametrize("name", TRAJECTORIES)
def test_golden_trajectories(host_check, name):
    replay_trajectory(lambda n, seed, off: HostCheckBatch(host_check, n, seed, off), load_golden(name))


@pytest.mark.parametrize("seed,offset,irw,max_exp,auto_reset", [
    (11, 0, 0.0, 0, True), (12, (1 << 32) - 4096, -1.0, 0, True), (13, 77, -2.5, 5, True), (14, 0, 0.0, 0, False)])
def test_random_rollouts_vs_oracle(host_check, seed, offset, irw, max_exp, auto_reset):
    n, steps = 4096, 96
    a, b = OracleBatch(n, seed, offset), HostCheckBatch(host_check, n, seed, offset)
    for o in (a, b):
        o.illegal_move_reward, o.max_exp = irw, max_exp
        o.reset()
    assert np.array_equal(a.boards, b.boards)
    for s in range(steps):
        for o in (a, b):
            o.step(None, auto_reset=auto_reset)
        for f in ("boards", "score", "reward", "terminated", "illegal", "highest", "last_score", "last_len",
                  "ep_count", "ep_start", "terminal_boards"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), (f, s)
    assert a.ep_count.sum() > 100


def test_reset_slots(host_check):
    for first_slot in range(0, 7):
        a, b = OracleBatch(64, 5), HostCheckBatch(host_check, 64, 5)
        a.reset(first_slot=first_slot, new_transaction=False)
        b.reset(first_slot=first_slot, new_transaction=False)
        assert np.array_equal(a.boards, b.boards), first_slot


# ------------------------------------------------------------------ board RECORD (cells + score deficit)

class HostRecordBatch(OracleBatch):
    """Mirror of step_kernel (g2048_kernels.hip) on the host: the state is the array of 16-byte records
    (+ the terminal records of finished episodes); plain boards and scores are exported after every call."""

    def __init__(self, hc, n, seed=0, board_offset=0, unpacked=False):
        super().__init__(n, seed, board_offset)
        self.hc = hc
        self.records = np.zeros((n, 16), np.uint8)
        self.last_records = np.zeros((n, 16), np.uint8)
        self.step_fn = hc.hostcheck_step_records_unpacked if unpacked else hc.hostcheck_step_records
        u64, vp = C.c_uint64, C.c_void_p
        hc.hostcheck_reset_records.restype = None
        hc.hostcheck_reset_records.argtypes = [C.POINTER(_Batch), vp, u64, u64, u64, u64, C.c_uint32]
        for f in (hc.hostcheck_step_records, hc.hostcheck_step_records_unpacked):
            f.restype = None
            f.argtypes = [C.POINTER(_Batch), vp, vp, u64, u64, u64, u64, C.c_float, C.c_int, C.c_int]

    def reset(self, first_slot=0, new_transaction=None):
        if new_transaction is None:
            new_transaction = not self.fresh
        if new_transaction:
            self.t += 1
        self.fresh = False
        b = self._batch()
        self.hc.hostcheck_reset_records(C.byref(b), self.records.ctypes.data, self.n, self.seed, self.t,
                                        self.board_offset, first_slot)

    def step(self, actions=None, auto_reset=True):
        if actions is not None:
            actions = np.ascontiguousarray(actions, dtype=np.uint8)
        self.t += 1
        self.fresh = False
        b = self._batch(actions)
        self.step_fn(C.byref(b), self.records.ctypes.data, self.last_records.ctypes.data, self.n, self.seed, self.t,
                     self.board_offset, self.illegal_move_reward, self.max_exp, int(auto_reset))


def test_move_lut(host_check):
    """move_sel (per-lane perm selectors) == move (transpose + selects) == the reference's move table."""
    m = load_golden("move_table")
    for i in range(len(m["boards"])):
        p = np.ascontiguousarray(m["boards"][i]).ctypes.data_as(U8P)
        for d in range(4):
            out = np.zeros(16, np.uint8)
            sc = C.c_uint32()
            legal = host_check.hostcheck_move_sel(p, d, out.ctypes.data_as(U8P), C.byref(sc))
            assert legal == m["legal"][i, d]
            assert np.array_equal(out, m["new"][i, d]) and sc.value == m["score"][i, d]


def test_potential_and_record_packing(host_check):
    rng = np.random.default_rng(8)
    host_check.hostcheck_potential.restype = C.c_uint32
    host_check.hostcheck_record_score.restype = C.c_uint32
    host_check.hostcheck_record_deficit.restype = C.c_uint32
    for trial in range(3000):
        cells = rng.integers(0, 18 if trial % 3 else 32, 16).astype(np.uint8)
        cells[rng.random(16) < 0.3] = 0
        e = cells.astype(np.int64)
        pot = int(np.where(e > 0, (e - 1) << e, 0).sum()) & 0xFFFFFFFF
        assert host_check.hostcheck_potential(cells.ctypes.data_as(U8P)) == pot
        score = int(rng.integers(0, 1 << 24))
        rec = np.zeros(16, np.uint8)
        host_check.hostcheck_make_record(cells.ctypes.data_as(U8P), score, rec.ctypes.data_as(U8P))
        assert np.array_equal(rec & 0x1F, cells) and not (rec[:8] & 0xE0).any()
        assert host_check.hostcheck_record_score(rec.ctypes.data_as(U8P)) == score
        d = host_check.hostcheck_record_deficit(rec.ctypes.data_as(U8P))
        assert d == (pot - score) & 0xFFFFFF
        # bit k of d sits in bit 5 + k % 3 of byte 8 + k // 3 (include/g2048.h)
        spread = sum(((int(rec[8 + k // 3]) >> (5 + k % 3)) & 1) << k for k in range(24))
        assert spread == d
        # "+4" carries ripple through the cell bits and wrap mod 2^24
        times = int(rng.integers(1, 40))
        host_check.hostcheck_record_bump(rec.ctypes.data_as(U8P), times)
        assert np.array_equal(rec & 0x1F, cells)
        assert host_check.hostcheck_record_score(rec.ctypes.data_as(U8P)) == (score - 4 * times) & 0xFFFFFF
    # carry chain across the whole field: d = 2^24 - 4, +4 -> 0
    cells = np.full(16, 31, np.uint8)
    rec = np.zeros(16, np.uint8)
    pot = (16 * (30 << 31)) & 0xFFFFFFFF
    host_check.hostcheck_make_record(cells.ctypes.data_as(U8P), (pot - (1 << 24) + 4) & 0xFFFFFF, rec.ctypes.data_as(U8P))
    assert host_check.hostcheck_record_deficit(rec.ctypes.data_as(U8P)) == (1 << 24) - 4
    host_check.hostcheck_record_bump(rec.ctypes.data_as(U8P), 1)
    assert host_check.hostcheck_record_deficit(rec.ctypes.data_as(U8P)) == 0 and np.array_equal(rec & 0x1F, cells)


def test_fresh_record_through_the_one_tile_table(host_check):
    rng = np.random.default_rng(4)
    a, b = np.zeros(16, np.uint8), np.zeros(16, np.uint8)
    words = rng.integers(0, 2 ** 32, size=(20000, 2), dtype=np.uint64)
    # directed: every cell of the first tile with BOTH tile values -- under the ABI-14 spawn rule the value comes from the
    # fraction the position leaves over, (w1 << 4) > 3865470566, so the low 28 bits are all-zero (a 2) / all-one (a 4) ...
    cell = np.arange(64) % 16
    words[:64, 0] = (cell << 28) | (0x0FFFFFFF * (np.arange(64) // 32))
    assert all(((int(w) << 4) & 0xFFFFFFFF > 3865470566) == (k >= 32) for k, w in enumerate(words[:64, 0]))
    # ... and the second tile just below / above the threshold of ITS rule, w2 * 15 mod 2^32 > 3865470566, in several cells
    thr = 3865470566
    for k in range(32):
        base = (k % 15) * 2 ** 32 + thr + (1 if k >= 16 else 0)          # w2 * 15 = base (+ up to 14): k-th empty cell, value 2 / 4
        w2 = -(-base // 15)
        words[64 + k, 1] = w2
        assert (w2 * 15) >> 32 == k % 15 and ((w2 * 15) & 0xFFFFFFFF > thr) == ((w2 * 15) & 0xFFFFFFFF >= thr + 1)
    fours_seen = 0
    for w1, w2 in words[:96]:
        fours_seen += ((int(w1) << 4) & 0xFFFFFFFF > thr) + ((int(w2) * 15) & 0xFFFFFFFF > thr)
    assert fours_seen >= 40                                               # the deficit bit of a spawned 4 is really exercised
    for w1, w2 in words:
        host_check.hostcheck_fresh_record_lut(int(w1), int(w2), a.ctypes.data_as(U8P), b.ctypes.data_as(U8P))
        assert np.array_equal(a, b), (hex(int(w1)), hex(int(w2)))


@pytest.mark.parametrize("unpacked", [False, True])
@pytest.mark.parametrize("name", TRAJECTORIES)
def test_golden_trajectories_records(host_check, name, unpacked):
    replay_trajectory(lambda n, seed, off: HostRecordBatch(host_check, n, seed, off, unpacked), load_golden(name))


@pytest.mark.parametrize("seed,offset,irw,max_exp,auto_reset", [
    (11, 0, 0.0, 0, True), (12, (1 << 32) - 4096, -1.0, 0, True), (13, 77, -2.5, 5, True), (14, 0, 0.0, 0, False)])
def test_random_rollouts_records_vs_oracle(host_check, seed, offset, irw, max_exp, auto_reset):
    n, steps = 4096, 96
    a, b = OracleBatch(n, seed, offset), HostRecordBatch(host_check, n, seed, offset)
    for o in (a, b):
        o.illegal_move_reward, o.max_exp = irw, max_exp
        o.reset()
    assert np.array_equal(a.boards, b.boards) and np.array_equal(a.score, b.score)
    for s in range(steps):
        for o in (a, b):
            o.step(None, auto_reset=auto_reset)
        for f in ("boards", "score", "reward", "terminated", "illegal", "highest", "last_score", "last_len",
                  "ep_count", "ep_start", "terminal_boards"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), (f, s)
    assert a.ep_count.sum() > 100


@pytest.mark.parametrize("obs_dtype,np_dtype", [(0, np.uint8), (1, np.float16), (2, np.float32)])
def test_onehot_chunks_of_a_wavefront(host_check, oracle_lib, obs_dtype, np_dtype):
    """The observation the step kernel writes (g2048_step_io.obs): onehot_chunk<OBS> of g2048_device.h -- the SWAR
    byte compare, the fp16 / fp32 expansion through v_perm, and the chunk -> (board, channel, cells) mapping of a
    wavefront's 64 boards -- against the reference's own stack() outputs (tests/golden/stack_table.npz, incl. 2^16 /
    2^17 tiles) and against the oracle for random boards, with the records' spare (deficit) bits set."""
    import ctypes as C
    t = load_golden("stack_table")
    rng = np.random.default_rng(obs_dtype)
    fn = host_check.hostcheck_onehot_wave
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    elem = np.dtype(np_dtype).itemsize
    batches = [(t["boards"][:64], t["onehot"][:64])]
    for _ in range(20):                                          # random exponents 0..17, many empties
        b = (rng.integers(0, 18, (64, 16)) * (rng.random((64, 16)) < 0.7)).astype(np.uint8)
        want = np.zeros((64, 16, 4, 4), np.uint8)
        oracle_lib.g2048o_onehot_batch(b.ctypes.data, 64, want.ctypes.data)
        batches.append((b, want))
    for boards, want in batches:
        rec = np.ascontiguousarray(boards, dtype=np.uint8).copy()
        rec[:, 8:] |= (rng.integers(0, 8, (64, 8)) << 5).astype(np.uint8)   # the packed score deficit must not leak
        out = np.zeros(64 * 256 * elem, np.uint8)
        fn(rec.ctypes.data, obs_dtype, out.ctypes.data)
        got = out.view(np_dtype).reshape(64, 16, 4, 4)
        assert np.array_equal(got, np.asarray(want).reshape(64, 16, 4, 4).astype(np_dtype))
