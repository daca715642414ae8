"""``Game2048Env`` -- the reference's single-env Gymnasium surface, served by the HIP engine.

Drop-in for ``env.envs.game2048_env.Game2048Env`` (game2048_env.py:34-288): same constructor,
``reset/step/render``, setters, test hooks and attributes.  The env is an N = 1 view over
``Batched2048``; every game computation (move, shift, spawn, done, one-hot) runs in
``libg2048_hip.so`` -- this file only converts formats (int64 tile values <-> uint8 exponents).

Randomness: the spawn stream (include/g2048.h).  ``reset(seed=s)`` restarts it; ``reset()``
continues it, as gymnasium does (game2048_env.py:103).  An env that was never seeded draws a seed
from the OS, like gymnasium.
"""
from __future__ import annotations

import os

import numpy as np

from .render import GRID_SIZE, render_board


def _env_base():
    """``gymnasium.Env`` when gymnasium is installed -- ``gym.make('2048-v0')`` and wrappers such as
    ``RecordVideo`` (ppo_train.py:102) insist on real ``gym.Env`` instances -- else ``object``."""
    try:
        import gymnasium
        return gymnasium.Env
    except ImportError:
        return object


class IllegalMove(Exception):
    """game2048_env.py:14-15."""


def stack(flat, layers=15):
    """game2048_env.py:17-32: (4,4) tile values -> (layers+1,4,4) one-hot (host arrays; the batched
    device version is ``Batched2048.observe_onehot`` / the ``obs`` output of ``step``)."""
    flat = np.asarray(flat)
    targets = np.concatenate([[0], 2 ** (np.arange(layers, dtype=np.int64) + 1)])
    return (flat[np.newaxis, :, :] == targets[:, np.newaxis, np.newaxis]).astype(int)


_M64 = (1 << 64) - 1
_CHANNELS = np.arange(16, dtype=np.uint8).reshape(16, 1, 1)
_TILE_VALUES = np.array([0] + [1 << e for e in range(1, 32)], dtype=np.int64)   # exponent -> tile value


def _stack_exp(cells):
    """``stack()`` of a uint8 (4,4) board of EXPONENTS: channel c = (exponent == c); a tile >= 2^16 matches no
    channel, like the reference (game2048_env.py:17-32).  Same dtype and shape as the reference's result."""
    return (cells == _CHANNELS).astype(int)


class _Space:
    """Stand-in used only when gymnasium is not installed."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def contains(self, x):
        return True


def make_spaces():
    """(action_space, observation_space) as game2048_env.py:49-52."""
    try:
        from gymnasium import spaces
        return spaces.Discrete(4), spaces.Box(0, 1, (16, 4, 4), dtype=int)
    except ImportError:
        return _Space(n=4, shape=(), dtype=np.int64), _Space(low=0, high=1, shape=(16, 4, 4), dtype=np.dtype(int))


def _values_to_exp(v):
    v = np.asarray(v).astype(np.int64)
    out = np.zeros(v.shape, np.uint8)
    nz = v > 0
    out[nz] = np.floor(np.log2(v[nz]) + 0.5).astype(np.uint8)
    return out


class SpawnStreamRNG:
    """What ``env.np_random`` is in the default RNG mode: the engine owns the randomness (the spawn stream of
    include/g2048.h, one Philox block per board and step, no generator object anywhere), so there is nothing for a
    caller to draw from.  Any use says so instead of silently handing out numbers the env will never consume."""

    _MESSAGE = ("the HIP engine owns the spawn stream (include/g2048.h): env.np_random is not what add_tile() draws "
                "from.  Construct the env with rng='numpy' for the reference's own numpy PCG64 generator "
                "(game2048_env.py:103,168,170), or use reset(seed=...) to restart the spawn stream.")

    def __getattr__(self, name):
        raise AttributeError(f"np_random.{name}: {self._MESSAGE}")

    def __repr__(self):
        return "<SpawnStreamRNG: the engine owns the spawn stream; rng='numpy' gives a numpy Generator>"


class Game2048Env(_env_base()):
    metadata = {"render_modes": ["ansi", "human", "rgb_array"], "render_fps": 4}  # game2048_env.py:35
    _all_positions = [(r, c) for r in range(4) for c in range(4)]                  # game2048_env.py:36

    def __init__(self, render_mode=None, *, device: int = 0, engine=None, rng: str = "philox"):
        # game2048_env.py:38-58
        self.size = 4
        self.w = self.size
        self.h = self.size
        self.squares = self.size * self.size
        self.score = 0
        self.action_space, self.observation_space = make_spaces()
        self._owns_engine = engine is None
        if engine is None:
            from .batched import Batched2048
            engine = Batched2048(1, device=device, seed=int.from_bytes(os.urandom(7), "little"), rng=rng)
        self._eng = engine
        self._io = engine.host_io()   # numpy views of the engine's pinned host block (actions in, outputs back)
        self._scratch = None    # a one-board engine for shift(), created on first use
        self._scratch_factory = getattr(engine, "scratch_factory", None)
        self._slot = 0          # next spawn slot of the current transaction
        self.set_illegal_move_reward(0.0)
        self.set_max_tile(None)
        self.grid_size = GRID_SIZE
        self.render_mode = render_mode

    # ------------------------------------------------------------------ configuration
    def set_illegal_move_reward(self, reward):
        """game2048_env.py:61-67."""
        self.illegal_move_reward = reward
        self.reward_range = (self.illegal_move_reward, float(2 ** self.squares))
        self._eng.set_illegal_move_reward(float(reward))

    def set_max_tile(self, max_tile):
        """game2048_env.py:69-73."""
        assert max_tile is None or isinstance(max_tile, int)
        self.max_tile = max_tile
        self._eng.set_max_tile(max_tile)

    # ------------------------------------------------------------------ gym interface
    def step(self, action):
        """game2048_env.py:76-100.  ONE call into the library: the action goes into the engine's pinned host block,
        the step kernel reads it and writes reward / flags / highest / the 16 cell bytes back into the same block, and
        ``stack()`` of those bytes is built here on the host -- no one-hot kernel, no staging copy for a single board."""
        io = self._io
        action = int(action)
        if not 0 <= action <= 3:
            # action_space is Discrete(4) (:49).  The reference does not check: `int(direction / 2)` / `direction % 2`
            # (:210-212) happen to play 4 as "down" and -1 as "right"; the engine would play the low two bits (4 -> "up").
            # Neither accident is worth keeping: refused, nothing is stepped.
            raise ValueError(f"action {action} is outside the action space Discrete(4) (0 up, 1 right, 2 down, 3 left)")
        io["actions"][0] = action
        self._eng.step_host(False)
        illegal = bool(io["illegal"][0])
        info = {"illegal_move": illegal}
        if illegal:
            reward = self.illegal_move_reward
            self._slot = 0
        else:
            reward = float(io["reward"][0])
            self.score += reward
            self._slot = 1
        info["highest"] = _TILE_VALUES[io["highest"][0]]   # :97 np.max(self.Matrix), an np.int64
        return _stack_exp(io["boards"][0]), reward, bool(io["terminated"][0]), False, info

    def reset(self, seed=None, options=None):
        """game2048_env.py:102-111."""
        if _env_base() is not object:
            super().reset(seed=seed)          # :103 (gymnasium bookkeeping: np_random, which this env does not draw from)
        if seed is not None:
            self._eng.seed(int(seed))
            self._slot = 0
        self._eng.reset(first_slot=self._slot, new_transaction=False)
        self._slot += 2
        self.score = 0
        return self._obs(), {}

    def render(self, mode=None):
        """game2048_env.py:113-163."""
        if mode is None:
            mode = self.render_mode or "human"
        return render_board(self.Matrix, self.score, mode)

    def close(self):
        """Drops the views of the engine's pinned host block (they die with the engine) and closes an engine this env
        created itself; an engine that was passed in stays the caller's."""
        self._io = None
        if self._owns_engine and self._eng is not None:
            self._eng.close()
        if self._scratch is not None:
            self._scratch.close()
            self._scratch = None

    # ------------------------------------------------------------------ np_random (game2048_env.py:103,168,170)
    @property
    def np_random(self):
        """The attribute the reference draws from.  ``rng='numpy'``: a ``numpy.random.Generator`` holding the CURRENT
        state of this board's PCG64 on the device -- a snapshot (drawing from it does not advance the engine; assign a
        generator to install its state).  Default mode: a ``SpawnStreamRNG`` whose every use raises with the reason."""
        if getattr(self._eng, "rng_mode", "philox") == "numpy":
            from .seeding import planes_to_generators
            return planes_to_generators(self._eng.get_numpy_rng())[0]
        return SpawnStreamRNG()

    @np_random.setter
    def np_random(self, generator):
        bg = getattr(generator, "bit_generator", None)
        st = getattr(bg, "state", None)
        if not isinstance(st, dict) or st.get("bit_generator") != "PCG64":
            raise TypeError("np_random must be a numpy.random.Generator over PCG64 (what gymnasium's seeding creates)")
        planes = np.array([[st["state"]["state"] & _M64], [st["state"]["state"] >> 64], [st["state"]["inc"] & _M64],
                           [st["state"]["inc"] >> 64], [st["uinteger"] | (st["has_uint32"] << 32)]], dtype=np.uint64)
        self._eng.set_numpy_rng(planes)     # switches the engine to the numpy-compatible RNG mode

    @property
    def unwrapped(self):
        return self

    # ------------------------------------------------------------------ board access
    def _obs(self):
        return _stack_exp(self._eng.fetch_host()["boards"][0])

    @property
    def Matrix(self):
        """game2048_env.py:104 -- int64 (4,4) tile values (a host copy of the device board)."""
        return _TILE_VALUES[self._eng.fetch_host()["boards"][0]]

    @Matrix.setter
    def Matrix(self, new_board):
        self._eng.set_boards(_values_to_exp(np.asarray(new_board).reshape(1, 4, 4)))

    def get(self, x, y):
        """game2048_env.py:178-180."""
        return self.Matrix[x, y]

    def set(self, x, y, val):
        """game2048_env.py:182-184."""
        m = self.Matrix
        m[x, y] = val
        self.Matrix = m

    def empties(self):
        """game2048_env.py:186-188."""
        return np.argwhere(self.Matrix == 0)

    def highest(self):
        """game2048_env.py:190-192."""
        return np.max(self.Matrix)

    def get_board(self):
        """game2048_env.py:282-284."""
        return self.Matrix

    def set_board(self, new_board):
        """game2048_env.py:286-288."""
        self.Matrix = new_board

    # ------------------------------------------------------------------ game primitives
    def add_tile(self):
        """game2048_env.py:166-176: one spawn from the next slot of the current transaction."""
        assert (self.Matrix == 0).any(), "No empty cell found"
        self._eng.add_tile(self._slot)
        self._slot += 1

    def move(self, direction, trial=False):
        """game2048_env.py:194-241."""
        score, legal = self._eng.move_numpy(np.array([int(direction)]), trial=bool(trial))
        if not legal[0]:
            raise IllegalMove
        return int(score[0])

    def shift(self, row):
        """game2048_env.py:243-260 (runs as a left move of a SCRATCH engine's board holding ``row``; the env's
        own board is not touched)."""
        if self._scratch is None:
            if self._scratch_factory is not None:
                self._scratch = self._scratch_factory()
            elif hasattr(self._eng, "device_index"):   # on the env's own device
                self._scratch = type(self._eng)(1, device=self._eng.device_index)
            else:
                self._scratch = type(self._eng)(1)
        board = np.zeros((1, 4, 4), np.uint8)
        board[0, 0] = _values_to_exp(row)
        self._scratch.set_boards(board)
        score, _ = self._scratch.move_numpy(np.array([3]), trial=False)
        out = _TILE_VALUES[self._scratch.fetch_host()["boards"][0, 0]]
        return [int(v) for v in out], int(score[0])

    def isend(self):
        """game2048_env.py:262-280."""
        return bool(self._eng.isend_numpy()[0])
