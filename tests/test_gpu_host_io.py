"""GPU: the host-resident forms of the C ABI (g2048_host_io_map / g2048_step_host / g2048_fetch_host: actions read
from, outputs written to, pinned device-mapped host memory; the host polls a completion word) and the adapters that
ride on them -- against the oracle and the golden trajectories captured from the reference."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,auto_reset", [(1, False), (1, True), (7, True), (256, True), (257, True), (5000, False)])
def test_step_host_vs_oracle(torch_cuda, n, auto_reset):
    """One-block launches (the kernel publishes the completion word itself) and multi-block launches (signal kernel
    behind the step): every output of every step equals the oracle's, incl. boards after the auto-reset."""
    from gym2048_amd.batched import Batched2048
    from oracle import OracleBatch
    seed = 9
    eng, ora = Batched2048(n, seed=seed, illegal_move_reward=-1.0), OracleBatch(n, seed)
    ora.illegal_move_reward = -1.0
    eng.reset()
    ora.reset()
    io = eng.host_io()
    assert np.array_equal(eng.fetch_host()["boards"].reshape(n, 16), ora.boards)
    rs = np.random.default_rng(n)
    for s in range(120 if n < 1000 else 30):
        a = rs.integers(0, 4, n)
        io["actions"][:] = a
        out = eng.step_host(auto_reset)
        ora.step(a.astype(np.uint8), auto_reset=auto_reset)
        assert np.array_equal(out["reward"], ora.reward), s
        assert np.array_equal(out["terminated"], ora.terminated), s
        assert np.array_equal(out["illegal"], ora.illegal), s
        assert np.array_equal(out["highest"], ora.highest), s
        assert np.array_equal(out["boards"].reshape(n, 16), ora.boards), s
        done = ora.terminated.astype(bool)
        assert np.array_equal(out["terminal_boards"].reshape(n, 16)[done], ora.terminal_boards[done]), s
    f = eng.fetch_host()
    assert np.array_equal(f["scores"], ora.score) and np.array_equal(f["boards"].reshape(n, 16), ora.boards)
    assert np.array_equal(eng.get_boards().reshape(n, 16), ora.boards)


def test_step_host_numpy_rng_mode(torch_cuda):
    from gym2048_amd.batched import Batched2048
    from oracle import OracleBatch
    n, seed = 300, 42
    eng, ora = Batched2048(n, seed=seed, rng="numpy"), OracleBatch(n, seed)
    ora.seed_numpy(seed)
    eng.reset()
    ora.reset_numpy()
    io = eng.host_io()
    rs = np.random.default_rng(3)
    for s in range(60):
        a = rs.integers(0, 4, n)
        io["actions"][:] = a
        out = eng.step_host(True)
        ora.step_numpy(a.astype(np.uint8))
        assert np.array_equal(out["boards"].reshape(n, 16), ora.boards), s
        assert np.array_equal(out["reward"], ora.reward) and np.array_equal(out["terminated"], ora.terminated), s
        assert np.array_equal(out["scores"], ora.score), s


def test_boards_out_through_step_and_rollout(torch_cuda):
    """g2048_step_io.boards_out with device buffers: plain cells after every step of a [k][n] rollout."""
    torch = torch_cuda
    import ctypes as C
    from gym2048_amd import _lib
    from gym2048_amd.batched import Batched2048
    from oracle import OracleBatch
    n, k, seed = 3001, 10, 4
    eng, ora = Batched2048(n, seed=seed), OracleBatch(n, seed)
    eng.reset()
    ora.reset()
    acts = eng.random_actions(k)
    out = torch.zeros((k, n, 16), dtype=torch.uint8, device=eng.device)
    io = eng._io(acts, None, None, None, None, None)
    io.boards_out = out.data_ptr()
    _lib.check(eng._lib.g2048_rollout(eng._h, k, C.byref(io), n, 1, eng._stream()))
    got, a = out.cpu().numpy(), acts.cpu().numpy()
    for j in range(k):
        ora.step(a[j])
        assert np.array_equal(got[j], ora.boards), j


def test_single_env_golden_trajectory_on_the_host_path(torch_cuda):
    """Game2048Env (N = 1, one library call per step, stack() built on the host from the 16 cell bytes) replays a
    golden trajectory of the reference: observations, rewards, flags, infos, score."""
    from gym2048_amd import Game2048Env
    from gym2048_amd.env import stack
    d = load_golden("traj_greedy_irw")
    seed, offset, n, steps, max_exp, auto_reset = (int(x) for x in d["meta"])
    assert offset == 0
    env = Game2048Env()
    env.set_illegal_move_reward(float(d["illegal_move_reward"][0]))
    obs, info = env.reset(seed=seed)            # board 0 of the batch = global board index 0
    vals = lambda e: np.where(e > 0, np.int64(1) << e.astype(np.int64), 0).reshape(4, 4)   # noqa: E731
    assert np.array_equal(obs, stack(vals(d["initial_boards"][0]))) and obs.dtype == np.dtype(int) and info == {}
    for s in range(min(steps, 400)):
        obs, reward, term, trunc, info = env.step(int(d["actions"][0, s]))
        assert isinstance(reward, float) and isinstance(term, bool) and trunc is False
        assert reward == float(d["reward"][0, s]) and term == bool(d["terminated"][0, s]), s
        assert info["illegal_move"] == bool(d["illegal"][0, s])
        want_board = d["terminal_boards"][0, s] if term else d["boards"][0, s]
        assert np.array_equal(obs, stack(vals(want_board))), s
        assert int(info["highest"]) == (1 << int(d["highest"][0, s])), s
        if term:
            env.reset()
        assert np.array_equal(env.get_board(), vals(d["boards"][0, s])), s


def test_vec_env_one_launch_per_step_with_fused_observation(torch_cuda):
    """Vec2048.step_wait: observations come from the step launch (uint8 in the packed download; int64 widened on
    the device) and equal the oracle-backed adapter's, infos included."""
    from gym2048_amd import Vec2048
    from fake_engine import OracleEngine
    for dtype in (np.uint8, np.int64):
        n, seed = 64, 5
        real = Vec2048(n, seed=seed, obs_dtype=dtype)
        fake = Vec2048(n, seed=seed, obs_dtype=dtype, engine=OracleEngine(n, seed))
        assert np.array_equal(real.reset(), fake.reset())
        rs = np.random.default_rng(0)
        for s in range(150):
            a = rs.integers(0, 4, n)
            o1, r1, d1, i1 = real.step(a)
            o2, r2, d2, i2 = fake.step(a)
            assert o1.dtype == np.dtype(dtype) and np.array_equal(o1, o2), s
            assert np.array_equal(r1, r2) and np.array_equal(d1, d2), s
            for x, y in zip(i1, i2):
                assert set(x) == set(y)
                for key in x:
                    if key == "terminal_observation":
                        assert np.array_equal(x[key], y[key])
                    elif key == "episode":
                        assert x[key]["r"] == y[key]["r"] and x[key]["l"] == y[key]["l"]
                    else:
                        assert x[key] == y[key]
        real.close()


def test_stream_signal_wait_and_capture_refusal(torch_cuda):
    """g2048_stream_signal / g2048_stream_wait: the completion word behind a launch train.  When the wait returns, every
    output of the launches enqueued before the signal is visible to the host (checked through mapped host memory, with
    no runtime synchronisation in between); tickets grow monotonically; an old ticket returns at once; a ticket that
    was never issued and a capturing stream are refused (a poll on a captured -- never executed -- kernel would not end)."""
    torch = torch_cuda
    from gym2048_amd._lib import G2048Error
    from gym2048_amd.batched import Batched2048
    from oracle import OracleBatch
    n, seed, k = 1 << 16, 4, 24
    eng, ora = Batched2048(n, seed=seed), OracleBatch(n, seed)
    eng.reset()
    ora.reset()
    io = eng.host_io()                                      # the rollout's last rewards land in mapped host memory
    acts = eng.random_actions(k)
    rew = torch.zeros((k, n), dtype=torch.float32, device=eng.device)
    tickets = []
    for rep in range(3):
        eng.rollout(acts if rep == 0 else k, reward=rew)
        tickets.append(eng.stream_signal())
        eng.stream_wait(tickets[-1])       # (the stream itself may still report "busy" for a moment: the word is
                                           #  published by the signal kernel's store, its retirement follows)
        for _ in range(k):
            ora.step(None)
        assert np.array_equal(rew[-1].cpu().numpy(), ora.reward)
    assert tickets == sorted(tickets) and len(set(tickets)) == 3
    eng.stream_wait(tickets[0])                             # already passed: returns at once
    with pytest.raises(G2048Error, match="not issued"):
        eng.stream_wait(tickets[-1] + 5)
    # a capturing stream: signal, step_host and fetch_host refuse instead of spinning on a kernel that never runs
    side = torch.cuda.Stream(device=eng.device)
    graph = torch.cuda.CUDAGraph()
    side.wait_stream(torch.cuda.current_stream(eng.device))
    with torch.cuda.stream(side):
        graph.capture_begin()
        try:
            with pytest.raises(G2048Error, match="capturing stream"):
                eng.stream_signal()
            with pytest.raises(G2048Error, match="capturing stream"):
                eng.step_host()
            with pytest.raises(G2048Error, match="capturing stream"):
                eng.fetch_host()
        finally:
            graph.capture_end()
    torch.cuda.synchronize()
    assert io["boards"].shape == (n, 4, 4)
    assert np.array_equal(eng.get_boards().reshape(n, 16), ora.boards)     # the refused calls changed nothing


def test_views_die_with_the_engine(torch_cuda):
    """close(): the numpy views of the pinned host block are dropped before the block is freed, and the host-resident
    calls of a closed engine raise instead of touching unmapped memory; Game2048Env.close() closes the engine it made."""
    from gym2048_amd import Game2048Env
    from gym2048_amd._lib import G2048Error
    from gym2048_amd.batched import Batched2048
    eng = Batched2048(64, seed=1)
    eng.reset()
    eng.host_io()["actions"][:] = 1
    eng.step_host()
    eng.close()
    assert eng._host_io is None and eng._boards_view is None
    for call in (eng.step_host, eng.fetch_host, eng.host_io):
        with pytest.raises(G2048Error):
            call()
    env = Game2048Env()
    env.reset(seed=3)
    env.step(0)
    env.close()
    assert env._io is None and not env._eng._h
    mine = Batched2048(1, seed=2)
    env = Game2048Env(engine=mine)
    env.reset(seed=2)
    env.close()
    assert mine._h and mine.host_io() is not None            # a caller's engine stays the caller's
    mine.close()
