"""``Batched2048`` -- N independent 2048 boards on one MI355X, driven through the C ABI.

Host-side mirror of the reference's env object (``/root/reference/env/envs/game2048_env.py``,
cited as ``game2048_env.py:LINE``) for a whole batch.  PyTorch-ROCm supplies device buffers and the
current HIP stream; every compute step is a call into ``libg2048_hip.so``.

Boards are ``uint8`` exponents (0 = empty, k = tile 2**k); ``tile_values()`` gives the reference's
int64 tile values.  Actions: 0 up, 1 right, 2 down, 3 left (game2048_env.py:196).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import G2048Error, HostIO, StepIO, Stats, check

_ACTION_DTYPES = {torch.uint8: _lib.ACT_U8, torch.int32: _lib.ACT_I32, torch.int64: _lib.ACT_I64}
_OBS_DTYPES = {torch.uint8: _lib.OBS_U8, torch.float16: _lib.OBS_F16, torch.float32: _lib.OBS_F32}


class _DeviceView:
    """Minimal ``__cuda_array_interface__`` carrier for zero-copy torch views of engine memory."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2}


def max_tile_to_exp(max_tile):
    """set_max_tile argument (game2048_env.py:69-73) -> exponent, 0 for None."""
    if max_tile is None:
        return 0
    if not isinstance(max_tile, (int, np.integer)):
        raise AssertionError("max_tile must be None or an int")  # game2048_env.py:72
    max_tile = int(max_tile)
    exp = max_tile.bit_length() - 1
    if max_tile < 2 or (1 << exp) != max_tile:
        # the reference compares highest() == max_tile (game2048_env.py:267); a value that is not a
        # tile can never be reached, which is "no limit"
        return 0
    return exp


class _RolloutPlan:
    """A validated rollout: the C struct with the buffer pointers (the tensors are kept alive here)."""

    def __init__(self, engine, fn, k, io, auto_reset, keep):
        self._engine, self._fn, self._k, self._io, self._auto_reset, self._keep = engine, fn, k, io, auto_reset, keep
        self._io_ref = C.byref(io)

    def run(self):
        e = self._engine
        check(self._fn(e._h, self._k, self._io_ref, e.n_envs, self._auto_reset, e._stream()))
        e._fresh = False

    def prepare_graph(self):
        """Build the cached hipGraph of this plan's launch train now (``g2048_rollout_prepare``: no step is executed), so
        that the first ``run()`` already is a replay.  A no-op where the form does not apply (large batches, optional
        outputs, numpy-RNG mode, fused plans).  The engine keeps the graphs of its four most recently built plans."""
        e = self._engine
        if self._fn is e._lib.g2048_rollout:
            check(e._lib.g2048_rollout_prepare(e._h, self._k, self._io_ref, e.n_envs, self._auto_reset))
        return self


class Batched2048:
    """``n_envs`` boards resident on ``cuda:device``; the batched counterpart of ``Game2048Env``.

    Global board index of local board ``i`` is ``board_offset + i``: the spawn randomness is keyed
    by the global index, so a batch sharded over several GPUs plays exactly the same games as the
    same batch on one GPU.
    """

    def __init__(self, n_envs: int, device: int = 0, seed: int = 0, board_offset: int = 0,
                 illegal_move_reward: float = 0.0, max_tile=None, rng: str = "philox", last_records: bool = True,
                 chains=None, strict_actions: bool = False):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        if not torch.cuda.is_available():
            raise G2048Error("Batched2048 needs a ROCm GPU (torch.cuda.is_available() is False); "
                             "there is no CPU fallback")
        self.n_envs = int(n_envs)
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        self.board_offset = int(board_offset)
        check(self._lib.g2048_create(self.n_envs, self.device_index, int(seed) & (2**64 - 1), self.board_offset,
                                     C.byref(self._h)))
        self._fresh = True
        if rng not in ("philox", "numpy"):
            raise ValueError("rng must be 'philox' (spawn stream) or 'numpy' (the reference's own PCG64)")
        self.rng_mode = rng
        self.illegal_move_reward = 0.0
        self.max_tile = None
        self.set_illegal_move_reward(illegal_move_reward)
        self.set_max_tile(max_tile)
        n, dev = self.n_envs, self.device
        # per-step outputs (reused every step; callers that need history pass their own buffers)
        self.reward = torch.zeros(n, dtype=torch.float32, device=dev)
        self.terminated = torch.zeros(n, dtype=torch.uint8, device=dev)
        self.illegal = torch.zeros(n, dtype=torch.uint8, device=dev)
        self.highest = torch.zeros(n, dtype=torch.uint8, device=dev)
        self.terminal_boards = torch.zeros((n, 16), dtype=torch.uint8, device=dev)
        self._boards_view = None
        if not last_records:
            self.set_last_records(False)
        # two-chain rollouts (g2048_set_chains) are OPT-IN: they start a launch thread and a high-priority stream per
        # device and only ever fire for rollouts of >= 12 steps whose actions are all supplied up front -- which a
        # trainer that steps one action at a time (ppo_train.py) never issues
        if chains is None:
            chains = 1
        if chains != 1:
            self.set_chains(chains)
        if strict_actions:
            self.set_strict_actions(True)
        if self.rng_mode == "numpy":
            self.seed(seed)
        # The engine may be used from ANY stream once the constructor has returned: what was enqueued here (the fills of
        # the output buffers, set_last_records, the numpy-mode seeding) went to the constructing thread's current
        # stream, and another stream is not ordered behind it.
        torch.cuda.current_stream(self.device).synchronize()

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            # the host-resident views alias pinned memory that g2048_destroy frees: drop them first, so that a stale
            # reference fails as "engine is closed" instead of reading unmapped memory
            self._host_io = None
            self._step_host_fn = self._fetch_host_fn = None
            self._boards_view = None
            self._lib.g2048_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------ configuration
    def set_illegal_move_reward(self, reward: float):
        """game2048_env.py:61-67."""
        self.illegal_move_reward = float(reward)
        check(self._lib.g2048_set_illegal_move_reward(self._h, self.illegal_move_reward))

    def set_max_tile(self, max_tile):
        """game2048_env.py:69-73."""
        exp = max_tile_to_exp(max_tile)
        self.max_tile = max_tile
        check(self._lib.g2048_set_max_tile(self._h, exp))

    def set_strict_actions(self, enable: bool):
        """``g2048_set_strict_actions``: with it on, a step / rollout whose action tensor holds a value outside 0..3
        (game2048_env.py:49 declares ``Discrete(4)``; :210-212 would play 4 as "down", this library plays the low two bits)
        still runs, but the NEXT call on the engine after that launch has completed raises ``G2048Error`` once; host-array
        steps (``step_host`` / ``step_numpy``) are refused before anything is stepped.  Default off."""
        check(self._lib.g2048_set_strict_actions(self._h, int(bool(enable))))

    @property
    def strict_actions(self) -> bool:
        return bool(self._lib.g2048_get_strict_actions(self._h))

    def set_last_records(self, enable: bool):
        """Keep (default) or stop keeping the terminal record of every board's most recent finished episode
        (``g2048_set_last_records``): with it off a step skips one sparse 16-byte store per finished episode
        (-0.85 us per launch at 2^20 boards); ``last_scores`` / ``last_records`` / the ``last_*`` statistics then are
        unavailable, ``episodes`` / ``illegal_ends`` / the exact ``return_sum`` are not affected."""
        check(self._lib.g2048_set_last_records(self._h, int(bool(enable)), self._stream()))

    def set_chains(self, chains: int):
        """1: a rollout is one chain of launches on the current stream.  2: ``rollout`` cuts the batch in two and runs
        the halves as two chains -- the lower half on the current stream from this thread, the upper half on an
        engine-owned side stream from an engine-owned launch thread -- forked from and joined back into the current
        stream inside every call; bit-identical results, 12-15 % less time per step at 2^19 .. 2^20 boards for rollouts of
        a few hundred steps (``g2048_set_chains``).  Opt-in (the default is 1): only a ``rollout`` of >= 12 steps with all
        its actions supplied up front can be split at all; the launch thread spins for ``G2048_SIDE_SPIN_US`` (default
        200 us, read when the device's side chain is created) after a job before it sleeps."""
        check(self._lib.g2048_set_chains(self._h, int(chains)))

    @property
    def chains(self) -> int:
        return int(self._lib.g2048_get_chains(self._h))

    @property
    def chains_used(self) -> int:
        """How many chains the most recent ``rollout`` ran as (a two-chain engine splits only rollouts that are long
        enough to pay: from 12 steps while the device's side chain is warm, i.e. used within the last 50 ms, 64 cold)."""
        return int(self._lib.g2048_get_chains_used(self._h))

    @property
    def graph_replays(self) -> int:
        """Rollouts of this engine that were replayed from the cached hipGraph of their launch train (small batches,
        standard outputs, the same buffers as the rollout before: ``g2048_get_graph_replays``)."""
        return int(self._lib.g2048_get_graph_replays(self._h))

    @property
    def graph_status(self) -> str:
        """``""`` while cached-graph replays are available to this engine, else why they are off (``g2048_graph_status``)."""
        return self._lib.g2048_graph_status(self._h).decode(errors="replace")

    @property
    def last_records_enabled(self) -> bool:
        return bool(self._lib.g2048_get_last_records(self._h))

    def seed(self, seed: int):
        """Seeding half of ``reset(seed=...)`` (game2048_env.py:103)."""
        self._fresh = True
        if self.rng_mode == "numpy":
            # board i <- numpy PCG64(SeedSequence(seed + global index)), as gymnasium / SB3 seed env i;
            # hashed and seeded on the device (g2048_pcg64.h), identical to numpy's own result
            check(self._lib.g2048_seed_numpy(self._h, int(seed) & (2**64 - 1), self._stream()))
        else:
            check(self._lib.g2048_seed(self._h, int(seed) & (2**64 - 1), self._stream()))

    def set_numpy_rng(self, planes):
        """Install per-board numpy PCG64 states (uint64 ``[5, n]``, see ``seeding.pcg64_planes``) and
        switch to the numpy-compatible RNG mode; ``None`` returns to the spawn stream."""
        if planes is None:
            check(self._lib.g2048_set_numpy_rng(self._h, None, self._stream()))
            self.rng_mode = "philox"
            return
        planes = np.ascontiguousarray(planes, dtype=np.uint64)
        if planes.shape != (5, self.n_envs):
            raise ValueError(f"planes must be uint64 [5, {self.n_envs}]")
        check(self._lib.g2048_set_numpy_rng(self._h, planes.ctypes.data, self._stream()))
        self.rng_mode = "numpy"

    def get_numpy_rng(self) -> np.ndarray:
        planes = np.empty((5, self.n_envs), np.uint64)
        check(self._lib.g2048_get_numpy_rng(self._h, planes.ctypes.data, self._stream()))
        return planes

    @property
    def clock(self) -> int:
        t = C.c_uint64()
        check(self._lib.g2048_get_clock(self._h, C.byref(t)))
        return t.value

    # ------------------------------------------------------------------ reset / step
    def reset(self, seed=None, mask=None, first_slot: int = 0, new_transaction=None):
        """game2048_env.py:102-111 for every board (or those with ``mask != 0``)."""
        if seed is not None:
            self.seed(seed)
        if new_transaction is None:
            new_transaction = not self._fresh
        mptr = None
        if mask is not None:
            mask = self._as_device(mask, torch.uint8)
            mptr = C.c_void_p(mask.data_ptr())
        check(self._lib.g2048_reset(self._h, int(bool(new_transaction)), int(first_slot), mptr, self._stream()))
        self._fresh = False

    def _as_device(self, x, dtype=None):
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(np.ascontiguousarray(x))
        if dtype is not None and x.dtype != dtype:
            x = x.to(dtype)
        if x.device != self.device:
            x = x.to(self.device)
        return x.contiguous()

    def _check_obs(self, obs, lead=()):
        """A fused-observation buffer: contiguous ``lead + (n, 16, 4, 4)`` uint8 / float16 / float32 on the engine's device."""
        if (obs.dtype not in _OBS_DTYPES or tuple(obs.shape) != tuple(lead) + (self.n_envs, 16, 4, 4)
                or not obs.is_contiguous() or obs.device != self.device):
            raise ValueError(f"obs must be a contiguous {tuple(lead) + (self.n_envs, 16, 4, 4)} uint8/float16/float32 "
                             f"tensor on {self.device}")
        return obs

    def _io(self, actions, reward, terminated, illegal, highest, terminal_boards, obs=None):
        io = StepIO()
        if actions is None:
            io.actions, io.action_dtype = None, _lib.ACT_RANDOM
        else:
            if actions.dtype not in _ACTION_DTYPES:
                raise TypeError(f"actions dtype {actions.dtype} not supported (uint8, int32, int64)")
            io.actions, io.action_dtype = actions.data_ptr(), _ACTION_DTYPES[actions.dtype]
        for name, t in (("reward", reward), ("terminated", terminated), ("illegal", illegal),
                        ("highest", highest), ("terminal_boards", terminal_boards)):
            setattr(io, name, None if t is None else t.data_ptr())
        if obs is not None:
            io.obs, io.obs_dtype = obs.data_ptr(), _OBS_DTYPES[obs.dtype]
        return io

    def step(self, actions=None, auto_reset: bool = True, want_info: bool = True, obs=None):
        """game2048_env.py:76-100 for every board; ``actions=None`` plays the synthetic random policy.

        Returns ``(reward, terminated)`` device tensors (views of ``self.reward`` / ``self.terminated``,
        overwritten by the next step).  With ``want_info`` also ``self.illegal``, ``self.highest`` and
        -- for boards that terminated -- ``self.terminal_boards`` are filled.  ``obs``: a ``[n, 16, 4, 4]``
        uint8 / float16 / float32 tensor that receives ``stack(board)`` of every board after the step (of the
        fresh board where an episode ended and ``auto_reset``), written by the SAME launch (:100).
        """
        if actions is not None:
            actions = self._as_device(actions)
            if actions.shape != (self.n_envs,):
                raise ValueError(f"actions must have shape ({self.n_envs},), got {tuple(actions.shape)}")
        io = self._io(actions, self.reward, self.terminated,
                      self.illegal if want_info else None, self.highest if want_info else None,
                      self.terminal_boards if want_info else None, None if obs is None else self._check_obs(obs))
        check(self._lib.g2048_step(self._h, C.byref(io), int(auto_reset), self._stream()))
        self._fresh = False
        return self.reward, self.terminated

    def rollout(self, actions, reward=None, terminated=None, illegal=None, highest=None, auto_reset: bool = True,
                fused: bool = False, terminal_boards=None, obs=None):
        """``k`` steps without returning to Python: ``actions`` is ``[k, n]`` (or ``k`` as an int for
        the synthetic random policy); optional outputs are ``[k, n]`` rollout buffers (``terminal_boards``:
        ``[k, n, 16]``, rows written only where an episode ended; ``obs``: ``[k, n, 16, 4, 4]``, the one-hot
        observation after every step, written by the step launches themselves; neither with ``fused``).
        ``fused=True`` runs them as ONE launch with the boards in registers (same outputs, ``g2048_rollout_fused``)."""
        self.prepare_rollout(actions, reward, terminated, illegal, highest, auto_reset, fused, terminal_boards, obs).run()

    def prepare_rollout(self, actions, reward=None, terminated=None, illegal=None, highest=None,
                        auto_reset: bool = True, fused: bool = False, terminal_boards=None, obs=None):
        """Validate the buffers of a rollout and build its launch descriptor once; ``.run()`` then is a
        single call into ``g2048_rollout`` (the argument checks stay out of a latency-critical loop)."""
        if isinstance(actions, int):
            k, act = actions, None
        else:
            act = self._as_device(actions)
            if act.dim() != 2 or act.shape[1] != self.n_envs:
                raise ValueError("actions must be [k, n_envs]")
            k = act.shape[0]
        for t, dt in ((reward, torch.float32), (terminated, torch.uint8), (illegal, torch.uint8), (highest, torch.uint8)):
            if t is None:
                continue
            if t.shape != (k, self.n_envs) or not t.is_contiguous() or t.device != self.device:
                raise ValueError("rollout buffers must be contiguous [k, n_envs] tensors on the engine's device")
            if t.dtype != dt and not (dt == torch.uint8 and t.dtype == torch.bool):
                raise TypeError(f"rollout buffer has dtype {t.dtype}; the kernels write {dt} "
                                "(reward float32; terminated / illegal / highest uint8 or bool)")
        if terminal_boards is not None and (tuple(terminal_boards.shape) != (k, self.n_envs, 16)
                                            or terminal_boards.dtype != torch.uint8
                                            or not terminal_boards.is_contiguous() or terminal_boards.device != self.device):
            raise ValueError("terminal_boards must be a contiguous uint8 [k, n_envs, 16] tensor on the engine's device")
        if obs is not None:
            self._check_obs(obs, (k,))
        io = self._io(act, reward, terminated, illegal, highest, terminal_boards, obs)
        fn = self._lib.g2048_rollout_fused if fused else self._lib.g2048_rollout
        return _RolloutPlan(self, fn, k, io, int(auto_reset), (act, reward, terminated, illegal, highest, terminal_boards, obs))

    def rollout_random(self, k_steps: int):
        """ONE fused launch: ``k_steps`` of the synthetic random policy, boards kept in registers."""
        check(self._lib.g2048_rollout_random(self._h, int(k_steps), self._stream()))
        self._fresh = False

    def random_actions(self, k_steps: int, t_first=None, out=None):
        """uint8 ``[k_steps, n]`` actions of the synthetic policy for transactions ``t_first..``
        (default: the next ``k_steps`` steps)."""
        if t_first is None:
            t_first = self.clock + 1
        if out is None:
            out = torch.empty((k_steps, self.n_envs), dtype=torch.uint8, device=self.device)
        if (out.dtype != torch.uint8 or tuple(out.shape) != (k_steps, self.n_envs) or not out.is_contiguous()
                or out.device != self.device):
            raise ValueError(f"out must be a contiguous uint8 [{k_steps}, {self.n_envs}] tensor on {self.device}")
        check(self._lib.g2048_fill_random_actions(self._h, int(t_first), int(k_steps), out.data_ptr(), self._stream()))
        return out

    # ------------------------------------------------------------------ observations
    def records(self) -> torch.Tensor:
        """Zero-copy ``uint8 [n, 16]`` view of the engine's board RECORDS (include/g2048.h): cell j of a
        board is ``byte j & 0x1f``; the spare bits of bytes 8..15 carry the packed score deficit."""
        if self._boards_view is None:
            ptr = self._lib.g2048_records_ptr(self._h)
            view = torch.as_tensor(_DeviceView(ptr, (self.n_envs, 16), "|u1"), device=self.device)
            if view.data_ptr() != ptr:
                raise G2048Error("torch did not alias the engine's record memory")
            self._boards_view = view
        return self._boards_view

    def boards(self, out=None) -> torch.Tensor:
        """``uint8 [n, 4, 4]`` exponents on the device (a snapshot written by a kernel into ``out``)."""
        if out is None:
            out = torch.empty((self.n_envs, 4, 4), dtype=torch.uint8, device=self.device)
        if out.dtype != torch.uint8 or out.numel() != self.n_envs * 16 or not out.is_contiguous() or out.device != self.device:
            raise ValueError("out must be a contiguous uint8 tensor of n*16 elements on the engine's device")
        check(self._lib.g2048_get_boards(self._h, out.data_ptr(), self._stream()))
        return out

    def scores(self, out=None) -> torch.Tensor:
        """``int32 [n]`` episodic merge scores (game2048_env.py:86) on the device (computed from the records)."""
        if out is None:
            out = torch.empty(self.n_envs, dtype=torch.int32, device=self.device)
        if out.dtype != torch.int32 or tuple(out.shape) != (self.n_envs,) or not out.is_contiguous() or out.device != self.device:
            raise ValueError("out must be a contiguous int32 [n] tensor on the engine's device")
        check(self._lib.g2048_get_scores(self._h, out.data_ptr(), self._stream()))
        return out

    def last_scores(self, out=None) -> torch.Tensor:
        """``int32 [n]`` on the device: final score of each board's most recently finished episode (0 before
        the first one), computed from the stored terminal records."""
        if out is None:
            out = torch.empty(self.n_envs, dtype=torch.int32, device=self.device)
        if out.dtype != torch.int32 or tuple(out.shape) != (self.n_envs,) or not out.is_contiguous() or out.device != self.device:
            raise ValueError("out must be a contiguous int32 [n] tensor on the engine's device")
        check(self._lib.g2048_get_last_scores(self._h, out.data_ptr(), self._stream()))
        return out

    def last_records(self) -> torch.Tensor:
        """Zero-copy ``uint8 [n, 16]``: the record each board's most recent episode ended on (all-zero: none)."""
        ptr = self._lib.g2048_last_records_ptr(self._h)
        if not ptr:
            raise G2048Error("this engine does not keep terminal records (set_last_records(True) turns them on)")
        return torch.as_tensor(_DeviceView(ptr, (self.n_envs, 16), "|u1"), device=self.device)

    def tile_values(self) -> torch.Tensor:
        """int64 ``[n, 4, 4]`` tile values as the reference holds them (game2048_env.py:104)."""
        e = self.boards().to(torch.int64)
        return torch.where(e > 0, torch.ones_like(e) << e, torch.zeros_like(e))

    def observe_onehot(self, dtype=torch.uint8, out=None) -> torch.Tensor:
        """``stack()`` (game2048_env.py:17-32) for the batch: ``[n, 16, 4, 4]`` of ``dtype``."""
        if out is None:
            out = torch.empty((self.n_envs, 16, 4, 4), dtype=dtype, device=self.device)
        if (out.dtype not in _OBS_DTYPES or tuple(out.shape) != (self.n_envs, 16, 4, 4) or not out.is_contiguous()
                or out.device != self.device):
            raise ValueError(f"one-hot output must be a contiguous [n,16,4,4] uint8/float16/float32 tensor on {self.device}")
        check(self._lib.g2048_onehot(self._h, out.data_ptr(), _OBS_DTYPES[out.dtype], self._stream()))
        return out

    # ------------------------------------------------------------------ host copies
    def get_boards(self) -> np.ndarray:
        """Host copy ``uint8 [n, 4, 4]`` (get_board, game2048_env.py:282-284)."""
        buf = np.empty((self.n_envs, 4, 4), np.uint8)
        check(self._lib.g2048_get_boards(self._h, buf.ctypes.data, self._stream()))
        return buf

    def set_boards(self, boards):
        """set_board (game2048_env.py:286-288): ``uint8 [n,4,4]`` / ``[n,16]`` exponents, host or device."""
        if isinstance(boards, torch.Tensor):
            b = boards.to(torch.uint8).contiguous()
            assert b.numel() == self.n_envs * 16
            if b.device.type == "cuda" and b.device != self.device:
                b = b.to(self.device)
            check(self._lib.g2048_set_boards(self._h, b.data_ptr(), self._stream()))
            if b.device.type == "cuda":
                b.record_stream(torch.cuda.current_stream(self.device))
        else:
            b = np.ascontiguousarray(boards, dtype=np.uint8)
            assert b.size == self.n_envs * 16
            check(self._lib.g2048_set_boards(self._h, b.ctypes.data, self._stream()))

    def get_scores(self) -> np.ndarray:
        buf = np.empty(self.n_envs, np.int32)
        check(self._lib.g2048_get_scores(self._h, buf.ctypes.data, self._stream()))
        return buf

    def set_scores(self, scores):
        """self.score for every board (int32 in 0 .. 2^24-1), host array or device tensor."""
        if isinstance(scores, torch.Tensor) and scores.device.type == "cuda":
            s = scores.to(self.device, torch.int32).contiguous()
            assert s.numel() == self.n_envs
            check(self._lib.g2048_set_scores(self._h, s.data_ptr(), self._stream()))
            s.record_stream(torch.cuda.current_stream(self.device))
            return
        s = np.ascontiguousarray(scores.cpu().numpy() if isinstance(scores, torch.Tensor) else scores, dtype=np.int32)
        assert s.size == self.n_envs
        check(self._lib.g2048_set_scores(self._h, s.ctypes.data, self._stream()))

    def get_last_scores(self) -> np.ndarray:
        """Host copy ``int32[n]``: final score of each board's most recently finished episode."""
        buf = np.empty(self.n_envs, np.int32)
        check(self._lib.g2048_get_last_scores(self._h, buf.ctypes.data, self._stream()))
        return buf

    def episode_stats(self) -> dict:
        """Episode statistics since create/seed, reduced on the device: ``episodes`` / ``illegal_ends`` count
        every finished episode, ``return_sum`` is the exact sum of their final merge scores (``mean_episode_score`` =
        the mean; ``mean_episode_return`` adds the illegal-move rewards, i.e. what SB3's Monitor reports as the mean of
        ``info["episode"]["r"]``, ppo_train.py:123); ``last_*`` describe each board's most recent finished episode."""
        st = Stats()
        check(self._lib.g2048_episode_stats(self._h, C.byref(st), self._stream()))
        out = parse_stats(bytes(st))
        out["mean_episode_return"] = ((out["return_sum"] + out["illegal_ends"] * self.illegal_move_reward) / out["episodes"]
                                      if out["episodes"] else 0.0)
        return out

    def episode_stats_device(self, out=None, returns_only: bool = False) -> torch.Tensor:
        """The same reduction as ``episode_stats`` left ON THE DEVICE: a ``uint8 [sizeof(g2048_stats)]`` tensor
        holding the C struct, enqueued on the current stream without a host sync (what a multi-GPU job
        all-gathers once per rollout; decode with ``parse_stats``).  ``returns_only``: only ``episodes``, ``illegal_ends``
        and the exact ``return_sum`` (histogram zero, ``last_*`` not computed: ``parse_stats`` gives None) -- the terminal records are not read."""
        nbytes = C.sizeof(Stats)
        if out is None:
            out = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        if out.dtype != torch.uint8 or out.numel() != nbytes or not out.is_contiguous() or out.device != self.device:
            raise ValueError(f"out must be a contiguous uint8 [{nbytes}] tensor on the engine's device")
        fn = self._lib.g2048_returns_summary_async if returns_only else self._lib.g2048_episode_stats_async
        check(fn(self._h, out.data_ptr(), self._stream()))
        return out

    # ------------------------------------------------------------------ checkpoint / resume
    def state_dict(self) -> dict:
        nbytes = self._lib.g2048_state_bytes(self._h)
        blob = np.empty(nbytes, np.uint8)
        check(self._lib.g2048_get_state(self._h, blob.ctypes.data, self._stream()))
        return {"blob": blob, "fresh": self._fresh, "rng_mode": self.rng_mode}

    def load_state_dict(self, state: dict):
        blob = np.ascontiguousarray(state["blob"], dtype=np.uint8)
        base = self._lib.g2048_state_bytes(self._h) - (40 * self.n_envs if self.rng_mode == "numpy" else 0)
        if blob.size not in (base, base + 40 * self.n_envs):
            raise ValueError("state blob size does not match this engine")
        check(self._lib.g2048_set_state(self._h, blob.ctypes.data, blob.size, self._stream()))
        self._fresh = bool(state.get("fresh", False))
        self.rng_mode = state.get("rng_mode", "philox")

    # ------------------------------------------------------------------ host-resident I/O
    def stream_signal(self) -> int:
        """Enqueue a completion ticket behind everything already enqueued on the current stream (one-wave kernel that
        publishes the ticket to the engine's pinned completion word); ``stream_wait(ticket)`` then returns when the device
        has got there -- the host polls the word, no runtime synchronisation call (``g2048_stream_signal`` / ``_wait``)."""
        t = C.c_uint64()
        check(self._lib.g2048_stream_signal(self._h, self._stream(), C.byref(t)))
        return t.value

    def stream_wait(self, ticket: int):
        rc = self._lib.g2048_stream_wait(self._h, int(ticket), self._stream())
        if rc:
            check(rc)

    def host_io(self) -> dict:
        """numpy views of the engine's pinned, device-mapped host block (``g2048_host_io``): ``actions`` int64[n]
        (IN), ``reward`` float32[n], ``terminated`` / ``illegal`` / ``highest`` uint8[n], ``boards`` and
        ``terminal_boards`` uint8[n,4,4], ``scores`` int32[n].  ``step_host`` / ``fetch_host`` fill them in place.
        The views alias engine-owned memory: they are valid until ``close()`` (copy what must outlive the engine)."""
        if getattr(self, "_host_io", None) is None:
            if not self._h:
                raise G2048Error("engine is closed")
            raw = HostIO()
            check(self._lib.g2048_host_io_map(self._h, C.byref(raw)))
            n = self.n_envs

            def view(ptr, ctype, shape):
                return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=shape)

            self._host_io = dict(
                actions=view(raw.actions, C.c_int64, (n,)), reward=view(raw.reward, C.c_float, (n,)),
                terminated=view(raw.terminated, C.c_uint8, (n,)), illegal=view(raw.illegal, C.c_uint8, (n,)),
                highest=view(raw.highest, C.c_uint8, (n,)), boards=view(raw.boards, C.c_uint8, (n, 4, 4)),
                terminal_boards=view(raw.terminal_boards, C.c_uint8, (n, 4, 4)), scores=view(raw.scores, C.c_int32, (n,)))
            # pre-bound call for the latency-critical single-env path
            self._step_host_fn, self._fetch_host_fn = self._lib.g2048_step_host, self._lib.g2048_fetch_host
        return self._host_io

    def step_host(self, auto_reset: bool = True) -> dict:
        """game2048_env.py:76-100 with host arrays: the caller has written ``host_io()['actions']``; ONE launch reads
        them across the bus and writes every output into the views, and the call returns when they are there (the
        host polls a completion word: no staging copy, no stream synchronisation).  Returns the same dict."""
        io = self.host_io()
        if not self._h:
            raise G2048Error("engine is closed")
        rc = self._step_host_fn(self._h, 1 if auto_reset else 0, self._stream())
        if rc:
            check(rc)
        self._fresh = False
        return io

    def fetch_host(self) -> dict:
        """Current boards and scores into the host views (after reset / set_boards / add_tile / move)."""
        io = self.host_io()
        if not self._h:
            raise G2048Error("engine is closed")
        check(self._fetch_host_fn(self._h, self._stream()))
        return io

    # ------------------------------------------------------------------ numpy facade (used by the
    # single-env and VecEnv adapters; tests replace this object by an oracle-backed fake)
    def _host_staging(self):
        """Buffers of the host-array path, created on first use: a pinned int64 action buffer and ONE packed
        output buffer on the device (reward | terminated | illegal | highest | terminal boards | boards, each
        part 16-byte aligned) with its pinned host twin -- a host step is one upload, two kernels, one download."""
        if getattr(self, "_stage", None) is None:
            n = self.n_envs
            up = lambda x: (x + 15) & ~15  # noqa: E731
            off, sizes = {}, (("reward", 4 * n), ("terminated", n), ("illegal", n), ("highest", n),
                              ("terminal_boards", 16 * n), ("boards", 16 * n), ("obs", 256 * n))
            pos = 0
            for name, size in sizes:
                off[name] = (pos, size)
                pos = up(pos + size)
            dev = torch.zeros(pos, dtype=torch.uint8, device=self.device)
            view = {k: dev[o:o + sz] for k, (o, sz) in off.items()}
            self._stage = dict(
                off=off, dev=dev, host=torch.zeros(pos, dtype=torch.uint8).pin_memory(),
                act_host=torch.zeros(n, dtype=torch.int64).pin_memory(),
                act_dev=torch.zeros(n, dtype=torch.int64, device=self.device),
                reward=view["reward"].view(torch.float32), terminated=view["terminated"], illegal=view["illegal"],
                highest=view["highest"], terminal_boards=view["terminal_boards"].view(n, 16), boards=view["boards"],
                obs=view["obs"].view(n, 16, 4, 4), small=off["obs"][0])
        return self._stage

    def step_numpy(self, actions, auto_reset: bool = True, obs_dtype=None) -> dict:
        """Step with host actions and bring every per-step output back in ONE device-to-host copy (reward,
        terminated, illegal, highest, terminal boards, boards after the step) through pinned memory.  ONE launch:
        the plain boards are the kernel's ``boards_out`` and, with ``obs_dtype``, the uint8 one-hot observation is
        its fused ``obs`` output, both written into the same packed buffer (``obs`` in the result; an integer type
        wider than uint8 is widened on the device and copied separately)."""
        st = self._host_staging()
        n = self.n_envs
        st["act_host"].numpy()[:] = np.asarray(actions).reshape(n)
        st["act_dev"].copy_(st["act_host"], non_blocking=True)
        io = self._io(st["act_dev"], st["reward"], st["terminated"], st["illegal"], st["highest"], st["terminal_boards"],
                      st["obs"] if obs_dtype is not None else None)
        io.boards_out = st["boards"].data_ptr()
        check(self._lib.g2048_step(self._h, C.byref(io), int(auto_reset), self._stream()))
        self._fresh = False
        want_u8_obs = obs_dtype is not None and np.dtype(obs_dtype) == np.uint8
        nbytes = st["dev"].numel() if want_u8_obs else st["small"]           # the observation rides along only as uint8
        st["host"][:nbytes].copy_(st["dev"][:nbytes], non_blocking=True)
        wide = None
        if obs_dtype is not None and not want_u8_obs:
            wide = st["obs"].to(getattr(torch, np.dtype(obs_dtype).name)).cpu().numpy()
        torch.cuda.current_stream(self.device).synchronize()
        raw = st["host"].numpy()
        part = lambda k: raw[st["off"][k][0]: st["off"][k][0] + st["off"][k][1]]  # noqa: E731
        # keep the engine's public per-step views coherent with what was just computed
        self.reward, self.terminated, self.illegal, self.highest = st["reward"], st["terminated"], st["illegal"], st["highest"]
        self.terminal_boards = st["terminal_boards"]
        out = dict(reward=part("reward").view(np.float32).copy(),
                   terminated=part("terminated").astype(bool), illegal=part("illegal").astype(bool),
                   highest=part("highest").copy(),
                   terminal_boards=part("terminal_boards").reshape(n, 4, 4).copy(),
                   boards=part("boards").reshape(n, 4, 4).copy())
        if want_u8_obs:
            out["obs"] = part("obs").reshape(n, 16, 4, 4).copy()
        elif wide is not None:
            out["obs"] = wide
        return out

    def onehot_numpy(self, dtype=np.uint8) -> np.ndarray:
        """Host copy of the one-hot observation in ``dtype``.  The kernel writes uint8; a wider integer type is
        produced by widening ON THE DEVICE and copying once (a host-side ``astype`` of 2 KiB per env was the
        slowest part of the VecEnv adapter)."""
        obs = self.observe_onehot(torch.uint8)
        dtype = np.dtype(dtype)
        if dtype != np.uint8:
            obs = obs.to(getattr(torch, dtype.name))
        return obs.cpu().numpy()

    def move(self, actions, trial: bool = False):
        """game2048_env.py:194-241 alone: returns device ``(score int32[n], legal uint8[n])``."""
        a = self._as_device(actions)
        if a.dtype not in _ACTION_DTYPES or a.shape != (self.n_envs,):
            raise ValueError("actions must be a uint8/int32/int64 tensor of shape (n_envs,)")
        score = torch.empty(self.n_envs, dtype=torch.int32, device=self.device)
        legal = torch.empty(self.n_envs, dtype=torch.uint8, device=self.device)
        check(self._lib.g2048_move(self._h, a.data_ptr(), _ACTION_DTYPES[a.dtype], int(bool(trial)),
                                   score.data_ptr(), legal.data_ptr(), self._stream()))
        return score, legal

    def move_numpy(self, actions, trial: bool = False):
        score, legal = self.move(torch.as_tensor(np.ascontiguousarray(actions, dtype=np.int64)), trial)
        return score.cpu().numpy(), legal.cpu().numpy().astype(bool)

    def query(self):
        """Device ``(isend uint8[n], highest-exponent uint8[n])`` (game2048_env.py:262-280, :190-192)."""
        end = torch.empty(self.n_envs, dtype=torch.uint8, device=self.device)
        hi = torch.empty(self.n_envs, dtype=torch.uint8, device=self.device)
        check(self._lib.g2048_query(self._h, end.data_ptr(), hi.data_ptr(), self._stream()))
        return end, hi

    def legal_actions(self, out=None) -> torch.Tensor:
        """Device ``uint8[n]`` mask, bit ``d`` set = move ``d`` is legal (the four trial moves of
        game2048_env.py:273-280 in one launch); 0 = the board has no move left."""
        if out is None:
            out = torch.empty(self.n_envs, dtype=torch.uint8, device=self.device)
        if out.dtype != torch.uint8 or out.shape != (self.n_envs,) or not out.is_contiguous() or out.device != self.device:
            raise ValueError(f"out must be a contiguous uint8 [{self.n_envs}] tensor on the engine's device")
        check(self._lib.g2048_legal_actions(self._h, out.data_ptr(), self._stream()))
        return out

    def isend_numpy(self) -> np.ndarray:
        return self.query()[0].cpu().numpy().astype(bool)

    def highest_numpy(self) -> np.ndarray:
        return self.query()[1].cpu().numpy()

    def set_clock(self, t: int):
        """Set the transaction counter (the next step is transaction ``t + 1``)."""
        check(self._lib.g2048_set_clock(self._h, int(t)))
        self._fresh = False

    def add_tile(self, slot: int):
        """game2048_env.py:166-176 from spawn slot ``slot`` of the current transaction."""
        check(self._lib.g2048_add_tile(self._h, int(slot), self._stream()))
        self._fresh = False

    def render(self, index: int = 0, mode: str = "ansi"):
        """game2048_env.py:113-163 for board ``index``: only that board's 16-byte record crosses PCIe."""
        from .render import render_board
        cells, score = decode_record(self.records()[int(index)].cpu().numpy())
        return render_board(exp_to_values(cells.reshape(4, 4)), score, mode)


def decode_record(raw) -> tuple:
    """Host-side reading of one 16-byte board record (include/g2048.h): ``(cells uint8[16], score)`` with
    score = potential - deficit mod 2^24, potential = sum of (e - 1) * 2^e over the tiles."""
    raw = np.asarray(raw, dtype=np.uint8).reshape(16)
    cells = raw & 0x1F
    deficit = sum(((int(raw[8 + k // 3]) >> (5 + k % 3)) & 1) << k for k in range(24))
    potential = sum((int(e) - 1) << int(e) for e in cells if e)
    return cells, (potential - deficit) & 0xFFFFFF


def parse_stats(raw) -> dict:
    """Decode one ``g2048_stats`` struct (bytes / uint8 array / tensor row from ``episode_stats_device``)."""
    if isinstance(raw, torch.Tensor):
        raw = raw.cpu().numpy()
    if not isinstance(raw, (bytes, bytearray)):
        raw = np.ascontiguousarray(raw, dtype=np.uint8).tobytes()
    st = Stats.from_buffer_copy(bytes(raw))
    # last_score_max == -1: the terminal records were not read (a returns-only summary, or an engine that does not keep
    # them) -- the last_* keys are then None, not a measured 0
    known = st.last_score_max >= 0
    return dict(episodes=st.episodes, illegal_ends=st.illegal_ends, last_known=known, last_count=st.last_count if known else None,
                last_score_sum=st.last_score_sum if known else None, last_score_max=st.last_score_max if known else None,
                max_exp=st.max_exp,
                mean_last_score=((st.last_score_sum / st.last_count) if st.last_count else 0.0) if known else None,
                highest_hist=[int(x) for x in st.highest_hist],
                return_sum=st.return_sum,      # exact: final merge scores of ALL finished episodes
                mean_episode_score=(st.return_sum / st.episodes) if st.episodes else 0.0)


def exp_to_values(exps) -> np.ndarray:
    e = np.asarray(exps).astype(np.int64)
    return np.where(e > 0, np.int64(1) << e, np.int64(0))


def values_to_exp(values) -> np.ndarray:
    v = np.asarray(values).astype(np.int64)
    if np.any(v < 0) or np.any((v & (v - 1)) != 0) or np.any(v == 1):
        raise ValueError("board values must be 0 or powers of two >= 2")
    out = np.zeros(v.shape, np.uint8)
    nz = v > 0
    out[nz] = np.floor(np.log2(v[nz]) + 0.5).astype(np.uint8)
    return out
