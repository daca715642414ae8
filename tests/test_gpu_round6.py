"""GPU tests of what round 6 added, all through the C ABI and checked against the oracle:

* the returns summary as ONE launch ("last block out" merge) at sizes that exercise one block, ragged last wavefronts, several
  groups of blocks and more than one trip of the grid-stride loop, back to back (the counters must come back to zero);
* ``g2048_allgather_summary``: the once-per-rollout exchange on the launch stream through the library's own communicator;
* strict actions (``g2048_set_strict_actions``): SURVEY 8b "Errors" -- an action outside 0..3 is reported, not silently
  played (reference: game2048_env.py:49 ``Discrete(4)``, :210-212).
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ro(eng):
    from gym2048_amd.batched import parse_stats
    return parse_stats(eng.episode_stats_device(returns_only=True))


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1023, 4133, (1 << 16) + 17, 1 << 20, (1 << 21) + 4133])
def test_one_launch_returns_summary_vs_oracle_books(torch_cuda, n):
    """g2048_returns_summary_async == the oracle's books (episodes, illegal ends, exact return sum) and == the full
    two-stage reduction, after auto-reset steps and after steps WITHOUT auto-reset (pending boards: an ended episode's score
    is still in the record and must not be subtracted), three calls in a row each time."""
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048
    from oracle import OracleBatch
    seed = 1234 + n % 97
    threads = 1 if n < 4096 else 16
    eng, ora = Batched2048(n, seed=seed, illegal_move_reward=-2.0), OracleBatch(n, seed, threads=threads)
    ora.illegal_move_reward = -2.0
    eng.reset()
    ora.reset()
    steps = 48 if n <= (1 << 16) + 17 else 24

    def check(where):
        full = eng.episode_stats()
        want = (int(ora.ep_count.sum()), ora.return_sum)
        for rep in range(3):                                   # the "last one out" counters are back at zero every time
            ro = _ro(eng)
            assert (ro["episodes"], ro["return_sum"]) == want, (where, rep, ro, want)
            assert ro["illegal_ends"] == full["illegal_ends"], (where, rep)
            assert ro["last_known"] is False and ro["last_score_max"] is None and ro["max_exp"] == 0
            assert sum(ro["highest_hist"]) == 0
        assert (full["episodes"], full["return_sum"]) == want, where

    check("reset")
    for s in range(steps):
        eng.step(None)
        ora.step(None)
    assert ora.ep_count.sum() > 0 or n < 64
    check("auto-reset steps")
    for s in range(steps // 2):
        eng.step(None, auto_reset=False)
        ora.step(None, auto_reset=False)
    assert ora.pending.any() or n < 64
    check("steps without auto-reset (pending marks)")
    eng.reset()
    ora.reset()
    check("reset of running and pending boards")
    assert np.array_equal(eng.get_scores(), ora.score)


def test_one_launch_summary_of_two_engines_interleaved(torch_cuda):
    """Every engine has its own scratch: summaries of two engines enqueued back to back on one stream, and on two streams
    at once, do not disturb each other."""
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048, parse_stats
    a, b = Batched2048(1 << 18, seed=5), Batched2048((1 << 17) + 77, seed=6)
    for e in (a, b):
        e.reset()
        e.rollout_random(80)
    want_a, want_b = a.episode_stats(), b.episode_stats()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for rep in range(8):
        with torch.cuda.stream(sa):
            oa = a.episode_stats_device(returns_only=True)
        with torch.cuda.stream(sb):
            ob = b.episode_stats_device(returns_only=True)
        outs.append((oa, ob))
    torch.cuda.synchronize()
    for oa, ob in outs:
        pa, pb = parse_stats(oa), parse_stats(ob)
        assert (pa["episodes"], pa["illegal_ends"], pa["return_sum"]) == (want_a["episodes"], want_a["illegal_ends"], want_a["return_sum"])
        assert (pb["episodes"], pb["illegal_ends"], pb["return_sum"]) == (want_b["episodes"], want_b["illegal_ends"], want_b["return_sum"])


def test_allgather_summary_on_the_launch_stream_one_rank(torch_cuda):
    """g2048_allgather_summary with a one-rank communicator (what a 1-GPU box can run): row 0 of the gathered array is
    this engine's returns-only summary, bit for bit, on the stream the rollout ran on, repeatedly; the Python wrapper
    (sharding.SummaryExchange, what bench.py's N > 1 path uses) gives the same rows and merge_stats reads them."""
    torch = torch_cuda
    from gym2048_amd import _lib
    from gym2048_amd.batched import Batched2048, parse_stats
    from gym2048_amd.sharding import SummaryExchange, merge_stats
    lib = _lib.load()
    n = 1 << 18
    eng = Batched2048(n, seed=9, last_records=False)
    eng.reset()
    ident = (C.c_uint8 * _lib.COMM_ID_BYTES)()
    _lib.check(lib.g2048_comm_unique_id(ident))
    comm = C.c_void_p()
    _lib.check(lib.g2048_comm_create(1, 0, ident, 0, C.byref(comm)))
    assert lib.g2048_comm_world(comm) == 1 and lib.g2048_comm_rank(comm) == 0
    row = C.sizeof(_lib.Stats)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        out = torch.full((1, row), 0xAB, dtype=torch.uint8, device=eng.device)
        for rep in range(3):
            eng.rollout_random(40)
            _lib.check(lib.g2048_allgather_summary(eng._h, comm, out.data_ptr(), C.c_void_p(stream.cuda_stream)))
            got = out.clone()
            want = eng.episode_stats_device(returns_only=True)
            stream.synchronize()
            assert bytes(got.cpu().numpy().tobytes()) == bytes(want.cpu().numpy().tobytes())
            p = parse_stats(got[0])
            assert p["episodes"] > 0 and p["last_known"] is False
        # argument checks: a host pointer is refused, so is a communicator of another device's engine (n/a here) / NULLs
        host = (C.c_uint8 * row)()
        assert lib.g2048_allgather_summary(eng._h, comm, C.addressof(host), None) == -1
        assert lib.g2048_allgather_summary(eng._h, None, out.data_ptr(), None) == -1
    _lib.check(lib.g2048_comm_destroy(comm))
    ex = SummaryExchange(eng)                       # no process group: a one-rank communicator
    assert (ex.world, ex.rank) == (1, 0)
    rows = ex.gather()
    torch.cuda.synchronize()
    g = merge_stats(rows)
    st = eng.episode_stats()
    assert (g["episodes"], g["illegal_ends"], g["return_sum"]) == (st["episodes"], st["illegal_ends"], st["return_sum"])
    assert g["last_count"] is None
    ex.close()


@pytest.mark.parametrize("rng", ["philox", "numpy"])
@pytest.mark.parametrize("dtype", ["uint8", "int32", "int64"])
def test_strict_actions_report_out_of_range_values(torch_cuda, rng, dtype):
    """With strict actions ON a step whose action tensor holds a value outside 0..3 still plays the low two bits (the boards
    are what the oracle gets for `action & 3`), and the NEXT call on the engine fails ONCE with G2048_ERR_INVALID naming an
    offending board; in-range steps never report; with strict actions OFF nothing is reported (the documented default)."""
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048, G2048Error
    from oracle import OracleBatch
    n, seed = 3000, 31
    tdt = getattr(torch, dtype)
    eng, ora = Batched2048(n, seed=seed, rng=rng, strict_actions=True), OracleBatch(n, seed)
    assert eng.strict_actions
    if rng == "numpy":
        ora.seed_numpy(seed)
        ora.step, ora.reset = ora.step_numpy, (lambda mask=None: ora.reset_numpy())
    eng.reset()
    ora.reset()
    rs = np.random.default_rng(3)
    for s in range(6):                                             # in range: never a report
        a = rs.integers(0, 4, n)
        eng.step(torch.as_tensor(a).to(eng.device, tdt))
        ora.step(a.astype(np.uint8))
    torch.cuda.synchronize()
    assert np.array_equal(eng.get_boards().reshape(n, 16), ora.boards)
    a = rs.integers(0, 4, n)
    bad_at = 1777
    a[bad_at] = 4 if dtype == "uint8" else -1                      # the reference's accidents: 4 -> "down", -1 -> "right"
    eng.step(torch.as_tensor(a).to(eng.device, tdt))
    ora.step((a & 3).astype(np.uint8))                             # what this library plays: the low two bits
    torch.cuda.synchronize()
    with pytest.raises(G2048Error, match=r"strict actions.*board 1777"):
        eng.get_boards()
    assert np.array_equal(eng.get_boards().reshape(n, 16), ora.boards)   # reported once; the engine goes on
    # a [k, n] rollout with one bad value somewhere in the middle
    acts = rs.integers(0, 4, (5, n))
    acts[3, 42] = 7
    eng.rollout(torch.as_tensor(acts).to(eng.device, tdt))
    for j in range(5):
        ora.step((acts[j] & 3).astype(np.uint8))
    torch.cuda.synchronize()
    with pytest.raises(G2048Error, match=r"strict actions.*board 42"):
        eng.step(None)
    # the fused forms (both RNG modes) check what they fetch as well
    acts = rs.integers(0, 4, (9, n))
    acts[8, 2999] = 5
    rew = torch.zeros((9, n), dtype=torch.float32, device=eng.device)
    eng.prepare_rollout(torch.as_tensor(acts).to(eng.device, tdt), reward=rew, fused=True).run()
    for j in range(9):
        ora.step((acts[j] & 3).astype(np.uint8))
    torch.cuda.synchronize()
    with pytest.raises(G2048Error, match=r"strict actions.*board 2999"):
        eng.get_scores()
    assert np.array_equal(eng.get_boards().reshape(n, 16), ora.boards)
    assert np.array_equal(eng.get_scores(), ora.score)
    # OFF (the default): the same value is played silently
    eng.set_strict_actions(False)
    a = rs.integers(0, 4, n)
    a[5] = 6
    eng.step(torch.as_tensor(a).to(eng.device, tdt))
    ora.step((a & 3).astype(np.uint8))
    torch.cuda.synchronize()
    assert np.array_equal(eng.get_boards().reshape(n, 16), ora.boards)


def test_strict_actions_host_step_is_refused_before_stepping(torch_cuda):
    """g2048_step_host reads its actions from host memory: with strict actions on a value outside 0..3 is refused BEFORE
    anything is stepped (boards and clock unchanged); Game2048Env.step raises ValueError for a Python int outside 0..3."""
    torch = torch_cuda
    from gym2048_amd import Game2048Env
    from gym2048_amd.batched import Batched2048, G2048Error
    eng = Batched2048(8, seed=2, strict_actions=True)
    eng.reset()
    io = eng.host_io()
    before, t0 = eng.get_boards().copy(), eng.clock
    io["actions"][:] = [0, 1, 2, 3, 4, 0, 0, 0]
    with pytest.raises(G2048Error, match=r"action 4 of board 4 is outside 0\.\.3"):
        eng.step_host(True)
    assert np.array_equal(eng.get_boards(), before) and eng.clock == t0
    io["actions"][:] = [0, 1, 2, 3, 3, 0, 0, 0]
    eng.step_host(True)
    assert eng.clock == t0 + 1
    env = Game2048Env()
    env.reset(seed=1)
    m = env.Matrix.copy()
    for bad in (4, -1, 17):
        with pytest.raises(ValueError, match="Discrete\\(4\\)"):
            env.step(bad)
    assert np.array_equal(env.Matrix, m)
    env.step(np.int64(3))


@pytest.mark.parametrize("auto_reset,irw,max_tile", [(True, -1.0, None), (False, 0.0, None), (True, 0.0, 64)])
def test_numpy_rng_mode_fused_rollout_vs_oracle(torch_cuda, auto_reset, irw, max_tile):
    """g2048_rollout_fused in numpy-RNG mode (record AND generator in registers, the lanes of a wavefront drifting in
    time: a lane resets in its own one or two trips of the loop while its neighbours go on stepping) == the C oracle's
    numpy mode stepped one step at a time: every per-step output at [j][i], the final boards, scores, generator states,
    the episode books -- for a ragged batch, with and without auto-reset, with max_tile, int64 actions."""
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048
    from oracle import OracleBatch
    n, seed, k = 4133, 19, 97
    eng = Batched2048(n, seed=seed, rng="numpy", illegal_move_reward=irw, max_tile=max_tile)
    ob = OracleBatch(n, seed, threads=8)
    ob.illegal_move_reward = irw
    ob.max_exp = int(np.log2(max_tile)) if max_tile else 0
    ob.seed_numpy(seed)
    eng.reset()
    ob.reset_numpy()
    rs = np.random.default_rng(8)
    acts = rs.integers(0, 4, (k, n))
    dev = eng.device
    rew = torch.full((k, n), -7.0, dtype=torch.float32, device=dev)
    term = torch.full((k, n), 9, dtype=torch.uint8, device=dev)
    ill = torch.full((k, n), 9, dtype=torch.uint8, device=dev)
    high = torch.full((k, n), 99, dtype=torch.uint8, device=dev)
    eng.prepare_rollout(torch.as_tensor(acts).to(dev), reward=rew, terminated=term, illegal=ill, highest=high,
                        auto_reset=auto_reset, fused=True).run()
    torch.cuda.synchronize()
    rew, term, ill, high = (x.cpu().numpy() for x in (rew, term, ill, high))
    for j in range(k):
        ob.step_numpy(acts[j].astype(np.uint8), auto_reset=auto_reset)
        assert np.array_equal(rew[j], ob.reward), j
        assert np.array_equal(term[j], ob.terminated), j
        assert np.array_equal(ill[j], ob.illegal), j
        assert np.array_equal(high[j], ob.highest), j
    assert ob.ep_count.sum() > n                                   # every board has been through resets
    assert np.array_equal(eng.get_boards().reshape(n, 16), ob.boards)
    assert np.array_equal(eng.get_scores(), ob.score)
    assert np.array_equal(eng.get_numpy_rng().T, ob.rng)
    assert eng.clock == k
    st = eng.episode_stats()
    assert st["episodes"] == int(ob.ep_count.sum()) and st["return_sum"] == ob.return_sum
    if auto_reset:
        assert st["return_sum"] == ob.finished_return_sum
    assert np.array_equal(eng.get_last_scores()[ob.ep_count > 0], ob.last_score[ob.ep_count > 0])
    # ... and the per-step form goes on from there in step with the oracle (nothing is left half-reset)
    for s in range(6):
        a = rs.integers(0, 4, n)
        eng.step(torch.as_tensor(a).to(dev), auto_reset=auto_reset)
        ob.step_numpy(a.astype(np.uint8), auto_reset=auto_reset)
    assert np.array_equal(eng.get_boards().reshape(n, 16), ob.boards) and np.array_equal(eng.get_numpy_rng().T, ob.rng)


def test_numpy_rng_mode_rollout_random_and_fused_equal_per_step_at_scale(torch_cuda):
    """2^18 boards x 160 steps in numpy-RNG mode three ways -- g2048_rollout (one launch per step), g2048_rollout_fused
    over the same [k][n] actions, g2048_rollout_random (the synthetic policy drawn in the kernel) -- identical boards,
    scores, generators, per-step outputs and books."""
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048
    n, seed, k = 1 << 18, 4, 160
    a, b, c = (Batched2048(n, seed=seed, rng="numpy") for _ in range(3))
    for e in (a, b, c):
        e.reset()
    acts = a.random_actions(k)
    dev = a.device
    ra, rb = (torch.zeros((k, n), dtype=torch.float32, device=dev) for _ in range(2))
    ta, tb = (torch.zeros((k, n), dtype=torch.uint8, device=dev) for _ in range(2))
    a.rollout(acts, reward=ra, terminated=ta)
    b.rollout(acts, reward=rb, terminated=tb, fused=True)
    c.rollout_random(k)
    torch.cuda.synchronize()
    assert torch.equal(ra, rb) and torch.equal(ta, tb) and int(ta.sum()) > n
    for other in (b, c):
        assert torch.equal(a.boards(), other.boards()) and torch.equal(a.scores(), other.scores())
        assert np.array_equal(a.get_numpy_rng(), other.get_numpy_rng())
        assert a.episode_stats() == other.episode_stats() and a.clock == other.clock == k
