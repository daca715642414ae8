"""``Vec2048`` -- a Stable-Baselines3 ``VecEnv``-shaped adapter over ``Batched2048``.

The reference trains with ``make_vec_env("2048-v0", n_envs)`` (ppo_train.py:123), i.e. SB3's serial
``DummyVecEnv`` of ``Monitor(Game2048Env)``.  This adapter offers the same surface with all envs
stepped by one HIP launch.  It mirrors what that stack hands to the learner:

* ``step_wait() -> (obs [n,16,4,4], rewards float32[n], dones bool[n], infos list[dict])``;
* auto-reset: a finished env returns the FIRST observation of its next episode, and its info carries
  ``terminal_observation``, ``episode = {"r", "l", "t"}`` (what Monitor adds), ``highest`` and
  ``illegal_move`` (what ppo_train.py:77-79 reads) and ``TimeLimit.truncated = False``;
* envs that did not finish share one read-only empty info dict (SB3 only reads infos).

When stable_baselines3 is importable the class derives from its ``VecEnv`` (and ``Game2048Env`` from
``gymnasium.Env``), so ``PPO(env=Vec2048(...))`` / ``gym.make('2048-v0')`` accept them unchanged; without
those packages (this image) the same classes are plain objects with the same surface.
"""
from __future__ import annotations

import time

import numpy as np

from .env import make_spaces

_EMPTY_INFO: dict = {}


def _vec_env_base():
    """SB3's ``VecEnv`` when stable_baselines3 is installed -- ``PPO(env=...)`` wraps anything that is not
    a ``VecEnv`` instance into a ``DummyVecEnv`` (ppo_train.py:123) -- else ``object``."""
    try:
        from stable_baselines3.common.vec_env import VecEnv
        return VecEnv
    except ImportError:
        return object


class Vec2048(_vec_env_base()):
    metadata = {"render_modes": ["ansi", "human", "rgb_array"], "render_fps": 4}

    def __init__(self, n_envs: int, device: int = 0, seed: int = 0, board_offset: int = 0,
                 illegal_move_reward: float = 0.0, max_tile=None, obs_dtype=np.int64, engine=None,
                 rng: str = "philox"):
        if engine is None:
            from .batched import Batched2048
            engine = Batched2048(n_envs, device=device, seed=seed, board_offset=board_offset, rng=rng)
        self.engine = engine
        self.action_space, self.observation_space = make_spaces()
        self.render_mode = None             # before VecEnv.__init__: SB3 reads it through get_attr("render_mode")
        self.num_envs = int(n_envs)
        self.obs_dtype = np.dtype(obs_dtype)
        self._illegal_move_reward = float(illegal_move_reward)
        if _vec_env_base() is not object:
            super().__init__(int(n_envs), self.observation_space, self.action_space)
        self.engine.set_illegal_move_reward(self._illegal_move_reward)
        self.engine.set_max_tile(max_tile)
        self._actions = None
        self._t0 = time.time()
        self._seed = seed
        # Monitor-style per-env episode accounting (SB3's Monitor keeps these on the host as well)
        self._ep_return = np.zeros(self.num_envs, np.float64)
        self._ep_length = np.zeros(self.num_envs, np.int64)

    # ------------------------------------------------------------------ VecEnv API
    def seed(self, seed=None):
        if seed is not None:
            self._seed = int(seed)
            self.engine.seed(self._seed)
        return [self._seed + i for i in range(self.num_envs)]

    def reset(self):
        self.engine.reset()
        self._ep_return[:] = 0
        self._ep_length[:] = 0
        return self._obs()

    def step_async(self, actions):
        self._actions = np.asarray(actions).reshape(self.num_envs)

    def step_wait(self):
        res = self.engine.step_numpy(self._actions, auto_reset=True, obs_dtype=self.obs_dtype)   # ONE launch, obs included
        dones = np.asarray(res["terminated"], dtype=bool)
        rewards = np.asarray(res["reward"], dtype=np.float32)
        obs = res["obs"]
        self._ep_return += rewards
        self._ep_length += 1
        infos = [_EMPTY_INFO] * self.num_envs
        idx = np.flatnonzero(dones)
        if idx.size:
            terminal = _onehot_host(res["terminal_boards"][idx], self.obs_dtype)
            now = round(time.time() - self._t0, 6)
            for j, i in enumerate(idx):
                infos[i] = {
                    "illegal_move": bool(res["illegal"][i]),
                    "highest": int(1 << int(res["highest"][i])) if res["highest"][i] else 0,
                    "episode": {"r": float(self._ep_return[i]), "l": int(self._ep_length[i]), "t": now},
                    "terminal_observation": terminal[j],
                    "TimeLimit.truncated": False,
                }
            self._ep_return[idx] = 0
            self._ep_length[idx] = 0
        return obs, rewards, dones, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        close = getattr(self.engine, "close", None)
        if close:
            close()

    def get_attr(self, attr_name, indices=None):
        if attr_name == "illegal_move_reward":
            value = self._illegal_move_reward
        elif attr_name == "render_mode":
            value = self.render_mode
        else:
            value = getattr(self.engine, attr_name)
        return [value] * len(self._indices(indices))

    def set_attr(self, attr_name, value, indices=None):
        if attr_name == "illegal_move_reward":
            self._illegal_move_reward = float(value)
            self.engine.set_illegal_move_reward(float(value))
        elif attr_name == "max_tile":
            self.engine.set_max_tile(value)
        else:
            raise AttributeError(f"cannot set {attr_name} on a batched env")

    def action_masks(self) -> np.ndarray:
        """bool ``[num_envs, 4]``: which moves are legal right now (the four trial moves of
        game2048_env.py:273-280 for every env in one launch) -- the method sb3-contrib's MaskablePPO asks a
        vector env for through ``env_method("action_masks")``."""
        mask = self.engine.legal_actions()
        mask = mask.cpu().numpy() if hasattr(mask, "cpu") else np.asarray(mask)
        return ((mask[:, None] >> np.arange(4, dtype=np.uint8)) & 1).astype(bool)

    def env_method(self, method_name, *args, indices=None, **kwargs):
        if method_name == "action_masks":
            rows = self.action_masks()
            return [rows[i] for i in self._indices(indices)]
        if method_name == "set_illegal_move_reward":
            self.set_attr("illegal_move_reward", *args)
        elif method_name == "set_max_tile":
            self.set_attr("max_tile", *args)
        else:
            raise AttributeError(f"env_method {method_name} is not available on a batched env")
        return [None] * len(self._indices(indices))

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * len(self._indices(indices))

    def get_images(self):
        return [self.engine.render(i, "rgb_array") for i in range(self.num_envs)]

    def render(self, mode=None):
        return self.engine.render(0, mode or "rgb_array")

    # ------------------------------------------------------------------ helpers
    def _indices(self, indices):
        if indices is None:
            return range(self.num_envs)
        if isinstance(indices, int):
            return [indices]
        return list(indices)

    def _obs(self):
        return self.engine.onehot_numpy(self.obs_dtype)


def _onehot_host(boards_exp: np.ndarray, dtype) -> np.ndarray:
    """One-hot of a handful of terminal boards (host side, only for finished envs' infos)."""
    b = np.asarray(boards_exp).reshape(-1, 1, 4, 4)
    return (b == np.arange(16, dtype=np.uint8).reshape(1, 16, 1, 1)).astype(dtype)
