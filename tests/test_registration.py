"""``import gym2048_amd`` registers ``'2048-v0'`` like ``import env`` does in the reference (env/__init__.py:1-6), and the
registered env serves the callers that go through the registry (ppo_train.py:102: ``RecordVideo(gym.make("2048-v0",
render_mode="rgb_array"))``).  gymnasium is not installed in this image, so the contract is exercised against a test-only
stand-in for the handful of gymnasium symbols involved (the same approach as tests/golden/make_golden.py's)."""
import importlib
import sys
import types

import numpy as np
import pytest

from fake_engine import OracleEngine


def gymnasium_stand_in():
    """gymnasium as far as registration, make() and a RecordVideo-like wrapper touch it."""
    gym = types.ModuleType("gymnasium")
    registry = {}

    class Env:
        metadata = {"render_modes": []}
        render_mode = None

        def reset(self, *, seed=None, options=None):
            if seed is not None:
                self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))

        @property
        def unwrapped(self):
            return self

        def close(self):
            pass

    class Spec:
        def __init__(self, id, entry_point, kwargs):
            self.id, self.entry_point, self.kwargs = id, entry_point, dict(kwargs)

    def register(id, entry_point=None, **kwargs):
        registry[id] = Spec(id, entry_point, kwargs.get("kwargs") or {})

    def make(id, **kwargs):
        spec = registry[id]                               # KeyError = gymnasium's "environment doesn't exist"
        creator = spec.entry_point
        if isinstance(creator, str):                      # "module:attr", loaded the way gymnasium.envs.registration does
            mod_name, attr = creator.split(":")
            creator = getattr(importlib.import_module(mod_name), attr)
        env = creator(**{**spec.kwargs, **kwargs})
        mode = kwargs.get("render_mode")
        assert mode is None or mode in env.metadata["render_modes"], "gymnasium.make checks the render mode against metadata"
        env.spec = spec
        return env

    spaces = types.ModuleType("gymnasium.spaces")

    class Discrete:
        def __init__(self, n):
            self.n = n

    class Box:
        def __init__(self, low, high, shape, dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, shape, np.dtype(dtype)

    spaces.Discrete, spaces.Box = Discrete, Box
    envs = types.ModuleType("gymnasium.envs")
    registration = types.ModuleType("gymnasium.envs.registration")
    registration.register, registration.registry = register, registry
    envs.registration = registration
    gym.Env, gym.spaces, gym.envs, gym.make, gym.register, gym.registry = Env, spaces, envs, make, register, registry
    return {"gymnasium": gym, "gymnasium.spaces": spaces, "gymnasium.envs": envs, "gymnasium.envs.registration": registration}


class RecordVideoLike:
    """What gymnasium.wrappers.RecordVideo does with an env (ppo_train.py:99-106): insists on render_mode 'rgb_array',
    reads metadata['render_fps'] if there, calls render() with NO arguments after reset and after every step and
    collects (H, W, 3) uint8 frames."""

    def __init__(self, env):
        assert env.render_mode in ("rgb_array", "rgb_array_list"), "RecordVideo refuses other render modes"
        self.env, self.frames = env, []
        self.fps = env.metadata.get("render_fps", 30)

    def _capture(self):
        frame = self.env.render()
        assert isinstance(frame, np.ndarray) and frame.dtype == np.uint8 and frame.ndim == 3 and frame.shape[2] == 3
        self.frames.append(frame)

    def reset(self, **kw):
        out = self.env.reset(**kw)
        self._capture()
        return out

    def step(self, action):
        out = self.env.step(action)
        self._capture()
        return out


def probe():
    """Runs in a FRESH interpreter (the package picks its base class, spaces and the registration at import time, and
    reloading modules inside the test process would leave other test modules holding stale classes)."""
    mods = gymnasium_stand_in()
    sys.modules.update(mods)
    import gym2048_amd as pkg                             # `import gym2048_amd` with gymnasium present
    if True:
        gym = mods["gymnasium"]
        spec = gym.registry["2048-v0"]                    # env/__init__.py:3-6: id '2048-v0'
        assert spec.entry_point == "gym2048_amd:Game2048Env" and pkg.ENV_ID == "2048-v0"
        assert issubclass(pkg.Game2048Env, gym.Env)
        # ppo_train.py:102 -- the extra keyword reaches the constructor (gymnasium.make passes them on): the engine here
        # is the oracle-backed fake, on a GPU box it is the HIP engine
        env = gym.make("2048-v0", render_mode="rgb_array", engine=OracleEngine(1, 7))
        assert type(env) is pkg.Game2048Env and env.render_mode == "rgb_array"
        assert "rgb_array" in env.metadata["render_modes"] and "ansi" in env.metadata["render_modes"]
        assert isinstance(env.action_space, mods["gymnasium.spaces"].Discrete) and env.action_space.n == 4
        assert env.observation_space.shape == (16, 4, 4)
        rec = RecordVideoLike(env)
        obs, info = rec.reset(seed=11)
        assert obs.shape == (16, 4, 4) and info == {}
        done, steps = False, 0
        while not done and steps < 400:                   # the loop of ppo_train.py:108-112 with a fixed policy
            obs, reward, terminated, truncated, info = rec.step(steps % 4)
            assert isinstance(reward, float) and truncated is False
            done = terminated or truncated
            steps += 1
        assert done and len(rec.frames) == steps + 1
        assert all(f.shape == (280, 280, 3) for f in rec.frames)                       # game2048_env.py:117-118,143
        assert any(not np.array_equal(rec.frames[0], f) for f in rec.frames[1:])       # the frames follow the game
        # the default constructor path is what a bare gym.make("2048-v0") takes: without a GPU it must fail loudly
        import torch
        if not torch.cuda.is_available():
            from gym2048_amd import G2048Error
            with pytest.raises(G2048Error):
                gym.make("2048-v0")
    print("registration probe ok:", steps, "steps,", len(rec.frames), "frames")


def probe_existing():
    """The reference's own package was imported first (env/__init__.py:3-6 registers the same id): importing this package
    must NOT silently take '2048-v0' over -- it warns and keeps the existing entry; an explicit register() takes it."""
    import warnings
    mods = gymnasium_stand_in()
    sys.modules.update(mods)
    gym = mods["gymnasium"]
    gym.register(id="2048-v0", entry_point="env.envs:Game2048Env")
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        import gym2048_amd as pkg
    assert gym.registry["2048-v0"].entry_point == "env.envs:Game2048Env"
    assert any("already registered" in str(w.message) and "env.envs:Game2048Env" in str(w.message) for w in caught), caught
    assert pkg.register() is True and gym.registry["2048-v0"].entry_point == "gym2048_amd:Game2048Env"
    assert pkg.register(force=False) is True          # already ours: nothing to warn about
    print("existing registration probe ok")


def test_import_keeps_an_existing_registration_of_the_id_and_warns():
    if importlib.util.find_spec("gymnasium") is not None:
        pytest.skip("the real gymnasium is installed")
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    code = "import sys; sys.path[:0] = [%r, %r]; import test_registration as t; t.probe_existing()" % (here, os.path.dirname(here))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "existing registration probe ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_import_registers_2048_v0_and_the_registry_env_renders_for_record_video():
    if importlib.util.find_spec("gymnasium") is not None:
        pytest.skip("the real gymnasium is installed: tests/test_host_logic.py::test_real_gymnasium_and_sb3_accept_the_drop_ins covers it")
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    code = "import sys; sys.path[:0] = [%r, %r]; import test_registration as t; t.probe()" % (here, os.path.dirname(here))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "registration probe ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_without_gymnasium_import_registers_nothing_and_register_says_why():
    if importlib.util.find_spec("gymnasium") is not None:
        pytest.skip("gymnasium is installed")
    import gym2048_amd
    with pytest.raises(ImportError):
        gym2048_amd.register()
