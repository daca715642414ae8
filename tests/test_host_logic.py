"""CPU: host-side logic of the product package -- the Game2048Env / Vec2048 adapters (against an
oracle-backed fake engine, tests/fake_engine.py), rendering, helpers, sharding arithmetic.  Mirrors
the reference's own unit tests (env/envs/test_game2048_env.py) on the drop-in class."""
import numpy as np
import pytest

from conftest import load_golden
from fake_engine import OracleEngine
from gym2048_amd import Game2048Env, IllegalMove, Vec2048, stack
from gym2048_amd.render import render_board, tile_colour
from gym2048_amd.sharding import shard_range, weak_shard


def make_env(seed=None, **kw):
    env = Game2048Env(engine=OracleEngine(1), **kw)
    if seed is not None:
        env.reset(seed=seed)
    return env


# ---- the reference's unit tests, re-stated for the drop-in class (test_game2048_env.py:10-231)

def test_shift_rows():
    b = make_env()
    k = load_golden("reference_test_kats")
    for row, out, score in zip(k["shift_rows"], k["shift_out"], k["shift_score"]):
        assert b.shift([int(v) for v in row]) == ([int(v) for v in out], int(score))


def test_move_all_directions_and_illegal_repeat():
    k = load_golden("reference_test_kats")
    b = make_env()
    for d in range(4):
        b.set_board(k["move_board"].copy())
        assert b.move(d) == k["move_score"][d]
        assert np.array_equal(b.get_board(), k["move_out"][d])
    with pytest.raises(IllegalMove):
        b.move(3)
    assert b.move(2) == 8
    assert np.array_equal(b.get_board(), k["follow_board"])
    b.set_board(k["move_board"].copy())
    assert b.move(0, trial=True) == 12 and np.array_equal(b.get_board(), k["move_board"])


def test_highest_and_isend():
    b = make_env()
    b.set_board(np.array([[0, 2, 0, 4], [2, 2, 8, 0], [2, 2, 2048, 8], [2, 2, 4, 4]]))
    assert b.highest() == 2048
    b.set_board(np.full((4, 4), 2))
    assert b.isend() is False
    checker = np.array([[2, 4, 8, 16], [4, 8, 16, 2], [8, 16, 2, 4], [16, 2, 4, 8]])
    b.set_board(checker)
    assert b.isend() is True
    one_empty = checker.copy()
    one_empty[3, 3] = 0
    b.set_board(one_empty)
    assert b.isend() is False
    b.set_max_tile(2048)
    top = np.zeros((4, 4), int)
    top[0, 0] = 2048
    b.set_board(top)
    assert b.isend() is True
    top[0, 0] = 1024
    b.set_board(top)
    assert b.isend() is False
    with pytest.raises(AssertionError):
        b.set_max_tile(2048.0)


def test_step_shapes_types_and_info():
    b = make_env(seed=0)
    obs, reward, terminated, truncated, info = b.step(0)
    assert obs.shape == (16, 4, 4) and obs.dtype == np.dtype(int)
    assert isinstance(reward, float) and isinstance(terminated, bool) and truncated is False
    assert set(info) == {"illegal_move", "highest"}


def test_step_reward_score_and_illegal():
    b = make_env(seed=0)
    b.set_board(np.array([[0, 0, 0, 0], [0, 0, 0, 0], [2, 0, 0, 0], [2, 0, 0, 0]]))
    _, reward, _, _, _ = b.step(0)
    assert reward == 4.0
    b.set_board(np.array([[0, 0, 0, 0], [0, 0, 0, 0], [4, 0, 0, 0], [4, 0, 0, 0]]))
    b.step(0)
    assert b.score == 12.0
    checker = np.array([[2, 4, 8, 16], [4, 8, 16, 2], [8, 16, 2, 4], [16, 2, 4, 8]])
    b.set_board(checker)
    _, reward, terminated, _, info = b.step(0)
    assert terminated is True and info["illegal_move"] is True and reward == 0.0
    assert np.array_equal(b.get_board(), checker)            # an illegal move changes nothing
    c = make_env()
    c.set_illegal_move_reward(-0.1)                          # not representable in float32
    c.reset(seed=0)
    c.set_board(checker)
    assert c.step(0)[1] == -0.1 and c.reward_range == (-0.1, 65536.0)


def test_step_refuses_actions_outside_discrete4():
    """action_space is Discrete(4) (game2048_env.py:49).  The reference never checks and `int(direction / 2)` /
    `direction % 2` (:210-212) happen to play 4 as "down" and -1 as "right"; the engine would play the low two bits.
    The drop-in refuses instead: ValueError, board, score and spawn slot untouched."""
    b = make_env(seed=3)
    before, score = b.get_board().copy(), b.score
    for bad in (4, -1, 255, np.int64(7)):
        with pytest.raises(ValueError, match=r"Discrete\(4\)"):
            b.step(bad)
    assert np.array_equal(b.get_board(), before) and b.score == score
    for good in (0, np.int64(1), np.uint8(2), 3):
        b.step(good)


def test_observation_is_one_hot_and_matches_stack():
    b = make_env(seed=0)
    b.set_board(np.array([[2, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 4, 0]]))
    obs, _, _, _, _ = b.step(1)
    assert obs.sum(axis=0).max() <= 1 and set(obs.flatten().tolist()) == {0, 1}
    assert np.array_equal(obs, stack(b.get_board()))
    s = load_golden("stack_table")
    for e, want in zip(s["boards"], s["onehot"]):
        vals = np.where(e > 0, 1 << e.astype(np.int64), 0).reshape(4, 4)
        assert np.array_equal(stack(vals), want)


# ---- behaviour beyond the reference's tests

def test_env_replays_golden_trajectory_with_manual_resets():
    d = load_golden("traj_greedy_irw")
    env = Game2048Env(engine=OracleEngine(1, board_offset=3))
    env.set_illegal_move_reward(-1.0)
    env.reset(seed=int(d["meta"][0]))
    vals = lambda e: np.where(e > 0, 1 << e.astype(np.int64), 0).reshape(4, 4)  # noqa: E731
    assert np.array_equal(env.Matrix, vals(d["initial_boards"][3]))
    for s in range(400):
        _, reward, term, _, info = env.step(int(d["actions"][3, s]))
        assert (reward, term, info["illegal_move"]) == (d["reward"][3, s], bool(d["terminated"][3, s]),
                                                        bool(d["illegal"][3, s]))
        if term:
            env.reset()      # continues the spawn stream: slots 1,2 after a legal move, 0,1 after an illegal one
        assert np.array_equal(env.get_board(), vals(d["boards"][3, s]))
        assert env.score == d["score"][3, s]


def test_reseed_restarts_and_reset_without_seed_continues():
    a, b = make_env(seed=5), make_env(seed=5)
    assert np.array_equal(a.get_board(), b.get_board())
    a.reset()
    assert not np.array_equal(a.get_board(), b.get_board()) or True   # different slots of transaction 0
    first = make_env(seed=5).get_board()
    a.reset(seed=5)
    assert np.array_equal(a.get_board(), first)
    e = make_env(seed=1)
    boards = set()
    for _ in range(6):                                       # repeated resets never replay the same slots
        e.reset()
        boards.add(e.get_board().tobytes())
    assert len(boards) > 3


def test_add_tile_get_set_empties():
    e = make_env(seed=2)
    n0 = len(e.empties())
    e.add_tile()
    assert len(e.empties()) == n0 - 1
    e.set(0, 0, 64)
    assert e.get(0, 0) == 64 and e.highest() >= 64
    e.set_board(np.full((4, 4), 2))
    with pytest.raises(AssertionError):
        e.add_tile()


def test_render_modes():
    e = make_env(seed=3, render_mode="ansi")
    out = e.render()
    text = out.getvalue()
    assert text.startswith("Score: 0\nHighest: ") and text.count("\n") == 6
    e.set_board(np.array([[2, 4, 8, 16], [32, 64, 128, 256], [512, 1024, 2048, 4096], [0, 0, 0, 2]]))
    img = e.render(mode="rgb_array")
    assert img.shape == (280, 280, 3) and img.dtype == np.uint8
    assert tuple(img[2, 2]) == (255, 0, 0) and tuple(img[278, 100]) == (128, 128, 128)
    assert tile_colour(2) == (255, 0, 0) and tile_colour(512) == (0, 255, 0) and tile_colour(4096) == (0, 160, 96)
    assert tile_colour(4) == (224, 32, 0) and tile_colour(256) == (32, 224, 0) and tile_colour(1024) == (0, 224, 32)
    # all twelve entries of the reference's colour map (game2048_env.py:120-133), as data
    ramp = [(255, 0, 0), (224, 32, 0), (192, 64, 0), (160, 96, 0), (128, 128, 0), (96, 160, 0), (64, 192, 0), (32, 224, 0),
            (0, 255, 0), (0, 224, 32), (0, 192, 64), (0, 160, 96)]
    assert [tile_colour(1 << k) for k in range(1, 13)] == ramp
    with pytest.raises(KeyError):
        tile_colour(8192)
    assert "4.0" in render_board(np.zeros((4, 4), int), 4.0, "ansi").getvalue()


def test_vec_env_surface_and_infos():
    n = 64
    ve = Vec2048(n, seed=11, illegal_move_reward=-1.0, engine=OracleEngine(n, 11))
    obs = ve.reset()
    assert obs.shape == (n, 16, 4, 4) and obs.dtype == np.int64 and ve.num_envs == n
    rng = np.random.default_rng(0)
    ret = np.zeros(n)
    length = np.zeros(n, int)
    finished = 0
    for _ in range(80):
        actions = rng.integers(0, 4, n)
        ve.step_async(actions)
        obs, rewards, dones, infos = ve.step_wait()
        assert rewards.dtype == np.float32 and dones.dtype == bool and len(infos) == n
        ret += rewards
        length += 1
        for i in range(n):
            if dones[i]:
                info = infos[i]
                assert info["episode"]["r"] == ret[i] and info["episode"]["l"] == length[i]
                assert info["terminal_observation"].shape == (16, 4, 4)
                assert info["TimeLimit.truncated"] is False and info["highest"] >= 2
                if info["illegal_move"]:
                    assert rewards[i] == -1.0
                # the returned obs is the FIRST observation of the next episode: exactly two tiles
                assert obs[i, 0].sum() == 14
                ret[i] = 0
                length[i] = 0
                finished += 1
            else:
                assert infos[i] == {}
    assert finished > 50
    assert ve.get_attr("illegal_move_reward") == [-1.0] * n
    ve.env_method("set_illegal_move_reward", -2.0)
    assert ve.get_attr("illegal_move_reward", indices=[0, 1]) == [-2.0, -2.0]
    assert ve.env_is_wrapped(object) == [False] * n and ve.seed(3)[:2] == [3, 4]
    assert ve.get_images()[0].shape == (280, 280, 3)
    # action masks (sb3-contrib's MaskablePPO convention): an illegal move really is the reference's IllegalMove
    masks = ve.action_masks()
    assert masks.shape == (n, 4) and masks.dtype == bool
    assert [m.tolist() for m in ve.env_method("action_masks", indices=[1, 5])] == [masks[1].tolist(), masks[5].tolist()]
    ve.step_async(np.array([int(np.argmin(m)) if not m.all() else 0 for m in masks]))      # an illegal move where one exists
    _, rewards, dones, infos = ve.step_wait()
    for i in range(n):
        if not masks[i].all():
            assert dones[i] and infos[i]["illegal_move"] and rewards[i] == -2.0


def test_vec_env_equals_independent_single_envs():
    """Batched auto-reset == N reference-style loops `step; if terminated: reset()`."""
    n, seed = 8, 21
    ve = Vec2048(n, seed=seed, engine=OracleEngine(n, seed))
    singles = [Game2048Env(engine=OracleEngine(1, seed, board_offset=i)) for i in range(n)]
    obs = ve.reset()
    for i, e in enumerate(singles):
        o, _ = e.reset(seed=seed)
        assert np.array_equal(o, obs[i])
    rng = np.random.default_rng(1)
    for _ in range(60):
        actions = rng.integers(0, 4, n)
        obs, rewards, dones, infos = ve.step(actions)
        for i, e in enumerate(singles):
            o, r, term, _, info = e.step(int(actions[i]))
            assert r == rewards[i] and term == dones[i]
            if term:
                assert np.array_equal(o, infos[i]["terminal_observation"])
                assert info["highest"] == infos[i]["highest"] and info["illegal_move"] == infos[i]["illegal_move"]
                o, _ = e.reset()
            assert np.array_equal(o, obs[i])


def test_shard_arithmetic():
    for n, w in [(10, 3), (1 << 23, 8), (7, 8), (1, 1)]:
        shards = [shard_range(n, r, w) for r in range(w)]
        assert sum(s.n_local for s in shards) == n
        assert all(a.stop == b.offset for a, b in zip(shards, shards[1:]))
        assert shards[0].offset == 0 and max(s.n_local for s in shards) - min(s.n_local for s in shards) <= 1
    s = weak_shard(1 << 20, 5, 8)
    assert (s.offset, s.n_local, s.n_global) == (5 << 20, 1 << 20, 1 << 23)
    with pytest.raises(ValueError):
        shard_range(4, 4, 4)


def test_drop_ins_take_gymnasium_and_sb3_base_classes_when_present(monkeypatch):
    """``gym.make('2048-v0')`` needs a ``gymnasium.Env``, ``PPO(env=...)`` a ``VecEnv`` (ppo_train.py:102,123):
    the base classes are chosen at import time.  gymnasium / SB3 are not installed in this image, so minimal
    stand-ins are put in ``sys.modules`` and the two modules re-imported."""
    import importlib
    import sys
    import types

    gym = types.ModuleType("gymnasium")

    class Env:
        def reset(self, *, seed=None, options=None):
            self.seeded_with = seed

    spaces = types.ModuleType("gymnasium.spaces")
    spaces.Discrete = lambda n: ("Discrete", n)
    spaces.Box = lambda lo, hi, shape, dtype=None: ("Box", shape)
    gym.Env, gym.spaces = Env, spaces
    sb3 = types.ModuleType("stable_baselines3")
    common = types.ModuleType("stable_baselines3.common")
    vec = types.ModuleType("stable_baselines3.common.vec_env")

    class VecEnv:
        def __init__(self, num_envs, observation_space, action_space):
            self.init_args = (num_envs, observation_space, action_space)
            # what SB3's VecEnv.__init__ does: it reads every env's render_mode through get_attr BEFORE the subclass
            # body has run any further -- the adapter must have set what get_attr touches by then
            self.render_modes_seen = self.get_attr("render_mode")

    vec.VecEnv = VecEnv
    for name, mod in (("gymnasium", gym), ("gymnasium.spaces", spaces), ("stable_baselines3", sb3),
                      ("stable_baselines3.common", common), ("stable_baselines3.common.vec_env", vec)):
        monkeypatch.setitem(sys.modules, name, mod)
    import gym2048_amd.env as env_mod
    import gym2048_amd.vec_env as vec_mod
    try:
        env_mod = importlib.reload(env_mod)
        vec_mod = importlib.reload(vec_mod)
        assert issubclass(env_mod.Game2048Env, Env) and issubclass(vec_mod.Vec2048, VecEnv)
        e = env_mod.Game2048Env(engine=OracleEngine(1, 3))
        obs, info = e.reset(seed=5)
        assert e.seeded_with == 5 and obs.shape == (16, 4, 4) and info == {}
        assert e.action_space == ("Discrete", 4) and e.observation_space == ("Box", (16, 4, 4))
        v = vec_mod.Vec2048(8, engine=OracleEngine(8, 3))
        assert v.init_args[0] == 8 and v.num_envs == 8 and v.reset().shape == (8, 16, 4, 4)
        assert v.render_modes_seen == [None] * 8
    finally:
        for name in ("gymnasium", "gymnasium.spaces", "stable_baselines3", "stable_baselines3.common",
                     "stable_baselines3.common.vec_env"):
            monkeypatch.delitem(sys.modules, name, raising=False)
        importlib.reload(env_mod)
        importlib.reload(vec_mod)
    assert env_mod.Game2048Env.__mro__[1] is object


def test_real_gymnasium_and_sb3_accept_the_drop_ins():
    """With the real packages installed: gym.make('2048-v0') returns our env, SB3 keeps Vec2048 as is."""
    gymnasium = pytest.importorskip("gymnasium")
    import gym2048_amd
    gym2048_amd.register(entry_point=lambda **kw: gym2048_amd.Game2048Env(engine=OracleEngine(1, 3), **kw))
    env = gymnasium.make("2048-v0")
    assert isinstance(env.unwrapped, gym2048_amd.Game2048Env)
    pytest.importorskip("stable_baselines3")
    from stable_baselines3.common.vec_env import VecEnv
    assert isinstance(gym2048_amd.Vec2048(4, engine=OracleEngine(4, 3)), VecEnv)


def eval_fixture_policy(weights):
    """The linear stand-in policy of tests/golden/eval_table.npz (make_golden.eval_policy_weights): exact integer
    scores + quarter-step tie-breakers, on numpy or torch observations."""
    w = weights.astype(np.float32)
    bias = np.array([0.75, 0.5, 0.25, 0.0], np.float32)

    def policy(obs):
        if isinstance(obs, np.ndarray):
            return np.einsum("bcyx,acyx->ba", obs.astype(np.float32), w) + bias
        import torch
        return torch.einsum("bcyx,acyx->ba", obs.float(), torch.from_numpy(w).to(obs.device)) + torch.from_numpy(bias).to(obs.device)
    return policy


@pytest.mark.parametrize("tag,eps", [("eps0", 0.0), ("eps03", 0.3)])
def test_evaluate_model_reproduces_the_reference_evaluation_loop(tag, eps, tmp_path):
    """evaluate.py (host logic) over the oracle in numpy-RNG mode == train.py's evaluate_episode / evaluate_model /
    report_evaluation_results run on the unmodified reference (fixture made by make_golden.gen_eval_table)."""
    from gym2048_amd.evaluate import evaluate_model, report_evaluation_results
    g = load_golden("eval_table")
    n = int(g["episodes"])
    res = evaluate_model(eval_fixture_policy(g["weights"]), n, eps, engine=OracleEngine(n, seed=456, rng="numpy"),
                         obs_dtype=np.float32)
    rows = res["Episodes"]
    assert [r["total_reward"] for r in rows] == g[f"{tag}_total_reward"].tolist()
    assert [r["highest"] for r in rows] == g[f"{tag}_highest"].tolist()
    assert [r["moves"] for r in rows] == g[f"{tag}_moves"].tolist()
    assert [r["illegal_moves"] for r in rows] == g[f"{tag}_illegal_moves"].tolist()
    assert res["Average score"] == sum(g[f"{tag}_total_reward"].tolist()) / n                 # train.py:203
    assert res["Max score"] == g[f"{tag}_total_reward"].max() and res["Highest tile"] == g[f"{tag}_highest"].max()
    name = report_evaluation_results(res, label=tag, path=str(tmp_path / f"scores_{tag}.csv"))
    assert open(name, "rb").read() == bytes(g[f"{tag}_csv"])                                  # train.py:216-229


def test_np_random_attribute_without_gymnasium():
    """Game2048Env.np_random (game2048_env.py:103,168,170; SURVEY 8b attribute list) exists whether or not gymnasium
    is installed.  Default mode: the engine owns the spawn stream and any use of the attribute says so.  rng='numpy':
    a real numpy Generator carrying the CURRENT state of the board's PCG64 -- what the reference's next add_tile() will
    draw -- and assigning a generator installs its state."""
    env = Game2048Env(engine=OracleEngine(1, 7))
    env.reset(seed=7)
    with pytest.raises(AttributeError, match="owns the spawn stream"):
        env.np_random.random()
    with pytest.raises(AttributeError, match="rng='numpy'"):
        env.np_random.shuffle([1, 2, 3])
    assert "SpawnStreamRNG" in repr(env.np_random)

    env = Game2048Env(engine=OracleEngine(1, 42, rng="numpy"))
    env.reset(seed=42)                                        # consumed exactly what the reference's reset(seed=42) consumes
    want = np.random.Generator(np.random.PCG64(np.random.SeedSequence(42)))
    for _ in range(2):                                        # reset() = two add_tile(): random() then shuffle(16 positions)
        want.random()
        want.shuffle(list(range(16)))
    got = env.np_random
    assert isinstance(got, np.random.Generator)
    assert got.bit_generator.state == want.bit_generator.state
    assert got.random() == want.random()                      # a snapshot: drawing from it ...
    assert env.np_random.bit_generator.state != got.bit_generator.state     # ... does not advance the engine's generator
    # installing a generator: the env then plays the reference's game for that generator
    other = Game2048Env(engine=OracleEngine(1, 0))
    other.np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(42)))
    other.reset()
    assert np.array_equal(other.Matrix, env.Matrix)           # same first board as reset(seed=42)
    with pytest.raises(TypeError):
        other.np_random = np.random.Generator(np.random.MT19937(1))
    env.close()
    assert env._io is None


def test_stats_rows_without_terminal_records_read_as_unknown_not_zero():
    """g2048_returns_summary_async (and an engine that keeps no terminal records) leaves last_score_max = -1: parse_stats
    turns the last_* keys into None, and merge_stats refuses to average a known shard with an unknown one."""
    from gym2048_amd._lib import Stats
    from gym2048_amd.batched import parse_stats
    from gym2048_amd.sharding import merge_stats
    full = Stats(episodes=10, illegal_ends=1, last_count=4, last_score_sum=400, last_score_max=150, max_exp=7, return_sum=900)
    summary = Stats(episodes=6, illegal_ends=0, last_count=0, last_score_sum=0, last_score_max=-1, max_exp=5, return_sum=300)
    a, b = parse_stats(bytes(full)), parse_stats(bytes(summary))
    assert (a["last_count"], a["last_score_sum"], a["last_score_max"], a["mean_last_score"]) == (4, 400, 150, 100.0)
    assert a["last_known"] is True and b["last_known"] is False      # the explicit flag next to the None values
    assert b["last_count"] is b["last_score_sum"] is b["last_score_max"] is b["mean_last_score"] is None
    assert (b["episodes"], b["return_sum"], b["mean_episode_score"]) == (6, 300, 50.0)
    both = merge_stats([bytes(full), bytes(summary)])
    assert (both["episodes"], both["illegal_ends"], both["return_sum"], both["max_exp"]) == (16, 1, 1200, 7)
    assert both["mean_episode_score"] == 75.0
    assert both["last_count"] is both["mean_last_score"] is both["last_score_max"] is None
    known = merge_stats([bytes(full), bytes(full)])
    assert (known["last_count"], known["last_score_sum"], known["last_score_max"], known["mean_last_score"]) == (8, 800, 150, 100.0)
    # a board that finished with score 0 is a measured 0, not "unknown"
    zero = parse_stats(bytes(Stats(episodes=1, last_count=1, last_score_sum=0, last_score_max=0)))
    assert zero["last_count"] == 1 and zero["mean_last_score"] == 0.0
