"""CPU ORACLE, Python half -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product package (``gym-2048_amd/``) never does and has no CPU fallback.

Contents
--------
* ``philox4x32_10`` / ``spawn_word`` / ``random_action`` -- the *spawn stream*, the counter-based
  RNG that replaces gymnasium's ``np_random`` (third-party numpy PCG64, unpinned).
* ``SpawnStream`` -- a duck-typed ``np_random`` (``.random()``, ``.shuffle(list)``) that is
  *injected into the unmodified reference env* by ``tests/golden/make_golden.py``; that is how
  "bit-exact vs the reference" is defined for an RNG the reference does not pin.
* ``RefEnv`` -- a small pure-Python restatement of ``Game2048Env`` (one board, tile values in a
  flat list) that follows ``/root/reference/env/envs/game2048_env.py`` line by line; each method
  cites the lines.  Used for small cases and as the "Python step()" CPU timing.
* ``onehot`` -- numpy restatement of ``stack()``.

Parity pin: ``tests/test_oracle_golden.py`` checks this module and the C oracle against the
vectors in ``tests/golden/`` that were captured from the imported reference.
"""
from __future__ import annotations

import numpy as np

M0, M1 = 0xD2511F53, 0xCD9E8D57  # Philox4x32 multipliers
W0, W1 = 0x9E3779B9, 0xBB67AE85  # Weyl key increments
MASK32 = 0xFFFFFFFF
TWO_THRESHOLD = 3865470566  # r / 2**32 < 0.9  <=>  r <= 3865470566, r = (w * n_empty) mod 2**32 (SURVEY 8c's constant)


def spawn_fraction(w: int, n_empty: int) -> float:
    """random() of the injected stream: the fraction of w * n_empty / 2**32 that the position k = (w * n_empty) >> 32
    leaves over -- uniform on [0, 1), independent of k up to the word's resolution (oracle/g2048_oracle.h)."""
    return ((w * n_empty) & MASK32) / 4294967296.0


def philox4x32_10(ctr, key):
    """Philox4x32-10 block function on Python ints."""
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & MASK32, p1 & MASK32, ((p0 >> 32) ^ c3 ^ k1) & MASK32, p0 & MASK32
        k0 = (k0 + W0) & MASK32
        k1 = (k1 + W1) & MASK32
    return c0, c1, c2, c3


def spawn_word(seed: int, t: int, board: int, slot: int) -> int:
    """word(seed, t, board, slot): see oracle/g2048_oracle.h."""
    ctr = (t & MASK32, (t >> 32) & MASK32, board & MASK32, (slot >> 2) & MASK32)
    key = (seed & MASK32, (seed >> 32) & MASK32)
    return philox4x32_10(ctr, key)[slot & 3]


def random_action(seed: int, t: int, board: int) -> int:
    """Synthetic uniform-random policy of the benchmark rollouts."""
    return spawn_word(seed, t, board, 3) >> 30


def philox4x32_10_np(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10 over numpy uint32 arrays (used to regenerate action tensors)."""
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint64) for x in (c0, c1, c2, c3))
    k0 = int(k0)
    k1 = int(k1)
    m32 = np.uint64(MASK32)
    for _ in range(10):
        p0 = np.uint64(M0) * c0
        p1 = np.uint64(M1) * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)) & m32, p1 & m32, \
                         ((p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)) & m32, p0 & m32
        k0 = (k0 + W0) & MASK32
        k1 = (k1 + W1) & MASK32
    return tuple(x.astype(np.uint32) for x in (c0, c1, c2, c3))


def random_actions_np(seed: int, t: int, board_offset: int, n: int) -> np.ndarray:
    """uint8[n] actions of transaction ``t`` for boards ``board_offset .. board_offset+n``."""
    boards = (np.arange(n, dtype=np.uint64) + np.uint64(board_offset)) & np.uint64(MASK32)
    full = np.full(n, 0, dtype=np.uint64)
    w = philox4x32_10_np(full + np.uint64(t & MASK32), full + np.uint64((t >> 32) & MASK32), boards, full,
                         seed & MASK32, (seed >> 32) & MASK32)
    return (w[3] >> np.uint32(30)).astype(np.uint8)


class SpawnStream:
    """Duck-typed ``np_random`` for RNG injection into the reference ``Game2048Env``.

    The reference's ``add_tile`` (game2048_env.py:166-176) calls ``random()`` once and then
    ``shuffle(positions)`` once.  Both are served from ONE 32-bit word of the spawn stream:

    * ``shuffle()`` -> moves the k-th empty cell of ``env.Matrix`` (row-major,
      ``k = (w * n_empty) >> 32``) to the front of the list, so the reference's "first empty
      cell in shuffled order" (game2048_env.py:171-175) is that cell;
    * ``random()``  -> ``((w * n_empty) mod 2**32) / 2**32``, the fraction that position left over
      (the reference itself compares it with 0.9; the board does not change between the two calls,
      game2048_env.py:168-170, so ``n_empty`` is the same in both).

    The driver calls ``begin_step()`` immediately before every ``env.step()``; ``env.reset()``
    without a seed simply continues with the next slots of the current transaction.
    """

    def __init__(self, env, seed: int, board: int = 0):
        self.env = env
        self.board = board
        self.reseed(seed)

    def reseed(self, seed: int):
        self.seed = seed
        self.t = 0
        self.slot = 0
        self._w = None

    def begin_step(self):
        self.t += 1
        self.slot = 0

    def random(self):
        self._w = spawn_word(self.seed, self.t, self.board, self.slot)
        return spawn_fraction(self._w, int((self.env.Matrix == 0).sum()))

    def shuffle(self, positions):
        w, self._w = self._w, None
        self.slot += 1
        matrix = self.env.Matrix
        empties = [p for p in positions if matrix[p[0], p[1]] == 0]
        if not empties:
            return
        target = empties[(w * len(empties)) >> 32]
        positions.remove(target)
        positions.insert(0, target)


class IllegalMoveError(Exception):
    """game2048_env.py:14-15 IllegalMove."""


class RefEnv:
    """One board; pure-Python restatement of Game2048Env (tile values, flat row-major list)."""

    def __init__(self, seed: int = 0, board: int = 0):
        self.M = [0] * 16
        self.score = 0
        self.illegal_move_reward = 0.0   # game2048_env.py:53
        self.max_tile = None             # game2048_env.py:54
        self.seed, self.board = seed, board
        self.t = 0
        self.slot = 0

    # -- game2048_env.py:243-260
    @staticmethod
    def shift(row):
        out = [0, 0, 0, 0]
        gained = 0
        n_out = 0
        mergeable = False
        for v in row:
            if v == 0:
                continue
            if mergeable and out[n_out - 1] == v:
                out[n_out - 1] = 2 * v
                gained += 2 * v
                mergeable = False
            else:
                out[n_out] = v
                n_out += 1
                mergeable = True
        return out, gained

    # -- game2048_env.py:194-241
    def move(self, direction, trial=False):
        vertical = (direction % 2 == 0)                    # :211,214
        towards_end = (direction % 2) ^ (direction // 2)   # :212
        changed = False
        total = 0
        for line in range(4):
            order = range(3, -1, -1) if towards_end else range(4)       # :218-219 / :230-231
            idx = [k * 4 + line if vertical else line * 4 + k for k in order]
            old = [self.M[i] for i in idx]
            new, gained = self.shift(old)                  # :220 / :232
            total += gained
            if new != old:                                 # :222 / :234
                changed = True
                if not trial:
                    for i, v in zip(idx, new):
                        self.M[i] = v
        if not changed:                                    # :238-239
            raise IllegalMoveError
        return total

    # -- game2048_env.py:190-192
    def highest(self):
        return max(self.M)

    # -- game2048_env.py:262-280
    def isend(self):
        if self.max_tile is not None and self.highest() == self.max_tile:
            return True
        if 0 in self.M:
            return False
        for d in range(4):
            try:
                self.move(d, trial=True)
                return False
            except IllegalMoveError:
                pass
        return True

    # -- game2048_env.py:166-176 with the injected spawn word
    def add_tile(self):
        w = spawn_word(self.seed, self.t, self.board, self.slot)
        self.slot += 1
        empties = [i for i in range(16) if self.M[i] == 0]
        value = 2 if spawn_fraction(w, len(empties)) < 0.9 else 4
        assert empties, "No empty cell found"
        self.M[empties[(w * len(empties)) >> 32]] = value

    # -- game2048_env.py:102-111
    def reset(self, seed=None):
        if seed is not None:
            self.seed, self.t, self.slot = seed, 0, 0
        self.M = [0] * 16
        self.score = 0
        self.add_tile()
        self.add_tile()

    # -- game2048_env.py:76-100; returns (reward, terminated, illegal, highest)
    def step(self, action):
        self.t += 1
        self.slot = 0
        try:
            gained = self.move(action)
            self.score += gained
            self.add_tile()
            terminated = self.isend()
            reward = float(gained)
            illegal = False
        except IllegalMoveError:
            illegal = True
            terminated = True
            reward = self.illegal_move_reward
        return reward, terminated, illegal, self.highest()


def onehot(values: np.ndarray) -> np.ndarray:
    """game2048_env.py:17-32 stack(): (...,4,4) tile values -> (...,16,4,4) {0,1} int64."""
    values = np.asarray(values)
    planes = [(values == 0)] + [(values == (1 << k)) for k in range(1, 16)]
    return np.stack(planes, axis=-3).astype(np.int64)


def values_to_exp(values) -> np.ndarray:
    v = np.asarray(values, dtype=np.int64)
    out = np.zeros(v.shape, dtype=np.uint8)
    nz = v > 0
    out[nz] = np.round(np.log2(v[nz])).astype(np.uint8)
    return out


def exp_to_values(exps) -> np.ndarray:
    e = np.asarray(exps).astype(np.int64)
    return np.where(e > 0, np.int64(1) << e, 0)


# ------------------------------------------------------------------ symmetries (training_data.py:257-299)
def symmetry_variant(board4x4, action, variant):
    """Variant ``2 * k + flip`` of training_data.augment(): hflip first (np.flip along the columns, actions
    1 <-> 3; training_data.py:257-273), then k clockwise quarter turns (np.rot90 with axes=(1, 0) on one
    board = axes=(2, 1) on the batch, action + k mod 4; :275-280)."""
    import numpy as np
    b = np.asarray(board4x4).reshape(4, 4)
    flip, k = variant & 1, variant >> 1
    if flip:
        b = np.flip(b, 1)
        action = {1: 3, 3: 1}.get(int(action), int(action))
    if k:
        b = np.rot90(b, k=k, axes=(1, 0))
        action = (int(action) + k) % 4
    return b, int(action)


def canonicalize(board4x4, action=0, next_board4x4=None):
    """Lexicographically smallest (row-major) of the eight symmetries, lowest variant index on ties.
    Returns (board, action, next_board | None, variant)."""
    best = min(range(8), key=lambda v: (tuple(symmetry_variant(board4x4, action, v)[0].reshape(16)), v))
    b, a = symmetry_variant(board4x4, action, best)
    nb = None if next_board4x4 is None else symmetry_variant(next_board4x4, 0, best)[0]
    return b, a, nb, best
