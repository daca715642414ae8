"""Test double for ``Batched2048``: the same numpy facade the host adapters use (``Game2048Env``,
``Vec2048``), backed by the CPU oracle.  Lives in tests/ only -- it exists so the adapters' host
logic (types, infos, slot bookkeeping, auto-reset handling) can be tested without a GPU.  The GPU
tests run the same adapter code against the real engine."""
import ctypes as C

import numpy as np

from oracle import OracleBatch

I64x16 = C.c_int64 * 16


class OracleEngine:
    def __init__(self, n_envs, seed=0, board_offset=0, rng="philox"):
        self.ob = OracleBatch(n_envs, seed, board_offset)
        self.n_envs = n_envs
        self.max_tile = None
        self.rng_mode = rng
        if rng == "numpy":
            self.ob.seed_numpy(int(seed))

    # configuration
    def set_illegal_move_reward(self, r):
        self.ob.illegal_move_reward = float(r)

    def set_max_tile(self, max_tile):
        self.max_tile = max_tile
        self.ob.max_exp = 0 if max_tile is None else int(max_tile).bit_length() - 1

    def seed(self, seed):
        if self.rng_mode == "numpy":
            self.ob.seed_numpy(int(seed))
        else:
            self.ob.seed_(int(seed))

    def close(self):
        pass

    def get_numpy_rng(self):
        return np.ascontiguousarray(self.ob.rng.T)          # uint64 [5, n], as Batched2048.get_numpy_rng

    def set_numpy_rng(self, planes):
        self.ob.rng = np.ascontiguousarray(np.asarray(planes, dtype=np.uint64).T)
        self.rng_mode = "numpy"

    # reset / step
    def reset(self, seed=None, first_slot=0, new_transaction=None, mask=None):
        if seed is not None:
            self.seed(seed)
        assert mask is None
        if self.rng_mode == "numpy":
            self.ob.reset_numpy()
        else:
            self.ob.reset(first_slot=first_slot, new_transaction=new_transaction)

    def observe_onehot(self, dtype=np.float32):
        return self.ob.onehot().astype(dtype)

    def step_numpy(self, actions, auto_reset=True, obs_dtype=None):
        if self.rng_mode == "numpy":
            self.ob.step_numpy(np.asarray(actions) & 3, auto_reset=auto_reset)
        else:
            self.ob.step(np.asarray(actions) & 3, auto_reset=auto_reset)
        o = self.ob
        out = dict(reward=o.reward.copy(), terminated=o.terminated.astype(bool), illegal=o.illegal.astype(bool),
                   highest=o.highest.copy(), terminal_boards=o.terminal_boards.reshape(-1, 4, 4).copy(),
                   boards=o.boards.reshape(-1, 4, 4).copy())
        if obs_dtype is not None:
            out["obs"] = o.onehot().astype(obs_dtype)
        return out

    # host-resident I/O (Batched2048.host_io / step_host / fetch_host)
    def host_io(self):
        if not hasattr(self, "_io"):
            n = self.n_envs
            self._io = dict(actions=np.zeros(n, np.int64), reward=np.zeros(n, np.float32), terminated=np.zeros(n, np.uint8),
                            illegal=np.zeros(n, np.uint8), highest=np.zeros(n, np.uint8),
                            boards=np.zeros((n, 4, 4), np.uint8), terminal_boards=np.zeros((n, 4, 4), np.uint8),
                            scores=np.zeros(n, np.int32))
        return self._io

    def step_host(self, auto_reset=True):
        io = self.host_io()
        res = self.step_numpy(io["actions"], auto_reset=auto_reset)
        for k in ("reward", "terminated", "illegal", "highest", "boards", "terminal_boards"):
            io[k][...] = res[k]
        io["scores"][...] = self.ob.score
        return io

    def fetch_host(self):
        io = self.host_io()
        io["boards"][...] = self.ob.boards.reshape(-1, 4, 4)
        io["scores"][...] = self.ob.score
        return io

    # observations / state
    def get_boards(self):
        return self.ob.boards.reshape(-1, 4, 4).copy()

    def set_boards(self, boards):
        self.ob.boards[:] = np.asarray(boards, dtype=np.uint8).reshape(-1, 16)

    def get_scores(self):
        return self.ob.score.copy()

    def onehot_numpy(self, dtype=np.uint8):
        return self.ob.onehot().astype(dtype, copy=False)

    def _values(self, i):
        return I64x16(*[0 if e == 0 else 1 << int(e) for e in self.ob.boards[i]])

    def move_numpy(self, actions, trial=False):
        score = np.zeros(self.n_envs, np.int32)
        legal = np.zeros(self.n_envs, bool)
        for i in range(self.n_envs):
            M, sc = self._values(i), C.c_int64()
            legal[i] = bool(self.ob.lib.g2048o_move(M, int(actions[i]) & 3, int(trial), C.byref(sc)))
            score[i] = sc.value if legal[i] else 0
            if legal[i] and not trial:
                self.ob.boards[i] = [0 if v == 0 else int(v).bit_length() - 1 for v in M]
        return score, legal

    def legal_actions(self):
        mask = np.zeros(self.n_envs, np.uint8)
        for d in range(4):
            mask |= self.move_numpy(np.full(self.n_envs, d), trial=True)[1].astype(np.uint8) << d
        return mask

    def isend_numpy(self):
        mt = 0 if self.max_tile is None else int(self.max_tile)
        return np.array([bool(self.ob.lib.g2048o_isend(self._values(i), mt)) for i in range(self.n_envs)])

    def add_tile(self, slot):
        for i in range(self.n_envs):
            M = self._values(i)
            w = self.ob.lib.g2048o_spawn_word(self.ob.seed, self.ob.t, self.ob.board_offset + i, slot)
            if self.ob.lib.g2048o_add_tile(M, w) >= 0:
                self.ob.boards[i] = [0 if v == 0 else int(v).bit_length() - 1 for v in M]

    def render(self, index=0, mode="ansi"):
        from gym2048_amd.render import render_board
        vals = np.where(self.ob.boards[index] > 0, 1 << self.ob.boards[index].astype(np.int64), 0)
        return render_board(vals.reshape(4, 4), int(self.ob.score[index]), mode)
