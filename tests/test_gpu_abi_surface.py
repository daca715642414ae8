"""GPU: the parts of the C ABI that nothing else drives -- add_tile in both RNG modes, set_scores / set_clock,
terminal_boards through g2048_rollout, the action generator's step-axis slabs, canonicalisation, the RCCL
collective in its degenerate one-rank form."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
I64x16 = C.c_int64 * 16


def _values(e):
    return I64x16(*[0 if x == 0 else 1 << int(x) for x in e])


def _exps(M):
    return [0 if v == 0 else int(v).bit_length() - 1 for v in M]


def _random_boards(rng, n):
    b = rng.integers(1, 12, (n, 16)).astype(np.uint8)
    b[rng.random((n, 16)) < rng.random((n, 1))] = 0      # every fill level, including full and empty boards
    return b


def test_add_tile_spawn_stream_vs_oracle(torch_cuda):
    """g2048_add_tile (game2048_env.py:166-176) for several slots of a transaction, incl. slots >= 4 (second
    Philox block) and full boards (the reference asserts there; the engine leaves them alone)."""
    from gym2048_amd.batched import Batched2048
    import oracle
    lib = oracle.load()
    rng = np.random.default_rng(2)
    n, seed, off = 4096, 99, 1 << 21
    boards = _random_boards(rng, n)
    eng = Batched2048(n, seed=seed, board_offset=off)
    eng.set_boards(boards)
    eng.set_scores(np.arange(n, dtype=np.int32) * 4)
    eng.set_clock(17)
    want = boards.copy()
    for slot in (0, 1, 2, 5):
        eng.add_tile(slot)
        for i in range(n):
            M = _values(want[i])
            if lib.g2048o_add_tile(M, lib.g2048o_spawn_word(seed, 17, off + i, slot)) >= 0:
                want[i] = _exps(M)
        assert np.array_equal(eng.get_boards().reshape(n, 16), want), slot
    assert np.array_equal(eng.get_scores(), np.arange(n, dtype=np.int32) * 4)     # add_tile never scores
    assert eng.clock == 17


def test_add_tile_numpy_mode_vs_oracle(torch_cuda):
    from gym2048_amd.batched import Batched2048
    from oracle import OracleBatch, load
    lib = load()
    rng = np.random.default_rng(3)
    n, seed = 2048, 5
    boards = _random_boards(rng, n)
    eng = Batched2048(n, seed=seed, rng="numpy")
    ob = OracleBatch(n, seed)
    ob.seed_numpy(seed)
    eng.set_boards(boards)
    want = boards.copy()
    for _ in range(3):
        eng.add_tile(0)
        for i in range(n):
            M = _values(want[i])
            if (want[i] == 0).any() and lib.g2048o_add_tile_numpy(M, ob.rng[i].ctypes.data) >= 0:
                want[i] = _exps(M)
        assert np.array_equal(eng.get_boards().reshape(n, 16), want)
    assert np.array_equal(eng.get_numpy_rng().T, ob.rng)      # full boards consumed nothing


def test_set_scores_set_clock_round_trips(torch_cuda):
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048
    from oracle import OracleBatch
    n, seed = 3000, 8
    eng, ora = Batched2048(n, seed=seed), OracleBatch(n, seed, threads=0)
    eng.reset()
    ora.reset()
    for _ in range(9):
        eng.step(None)
        ora.step(None)
    rng = np.random.default_rng(0)
    scores = rng.integers(0, 1 << 24, n).astype(np.int32)
    scores[:4] = (0, 1, (1 << 24) - 1, 3)                      # odd values and both ends of the range
    eng.set_scores(scores)                                     # host array
    ora.score[:] = scores
    assert np.array_equal(eng.get_scores(), scores) and np.array_equal(eng.get_boards().reshape(n, 16), ora.boards)
    eng.set_scores(torch.as_tensor(scores[::-1].copy()).to(eng.device))   # device tensor
    assert np.array_equal(eng.scores().cpu().numpy(), scores[::-1])
    scores[2] = 12345                                          # (play on below: stay clear of the 2^24 wrap)
    eng.set_scores(scores)
    ora.score[:] = scores
    with pytest.raises(Exception):
        eng.set_scores(np.full(n, 1 << 24, np.int32))          # outside the 24-bit score range
    # set_boards keeps the scores (game2048_env.py:286-288)
    eng.set_boards(ora.boards[::-1].copy())
    assert np.array_equal(eng.get_scores(), scores)
    eng.set_boards(ora.boards)
    # jump the clock on both sides and keep playing: the spawn stream is a pure function of (seed, t, board)
    eng.set_clock(1 << 33)
    ora.t = 1 << 33
    assert eng.clock == 1 << 33
    for _ in range(12):
        eng.step(None)
        ora.step(None)
        assert np.array_equal(eng.get_boards().reshape(n, 16), ora.boards)
        assert np.array_equal(eng.get_scores(), ora.score)
    assert eng.clock == (1 << 33) + 12


@pytest.mark.parametrize("rng", ["philox", "numpy"])
def test_return_sum_accounting_through_every_call_that_moves_a_score(torch_cuda, rng):
    """g2048_stats.return_sum -- the exact sum of the final scores of ALL finished episodes -- is kept by conservation
    (sum of every wavefront's running gain total minus the scores of the episodes still running), so every call that
    moves a score has to keep the books: steps with and without auto-reset (an ended, un-reset episode is finished
    although its score is still in the record), per-step / [k,n] / fused / random rollouts, full and masked resets
    (a running episode is abandoned), set_scores, set_boards, reseeding (books restart at zero), state round trips.
    After every call: both statistics flavours == the oracle's books; with auto-reset only, == the plain sum of the
    final scores the oracle saw episode by episode."""
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048, parse_stats
    from oracle import OracleBatch
    n, seed = 5000, 77
    numpy_mode = rng == "numpy"
    eng, ora = Batched2048(n, seed=seed, illegal_move_reward=-1.0, rng=rng), OracleBatch(n, seed)
    ora.illegal_move_reward = -1.0
    if numpy_mode:
        ora.seed_numpy(seed)
        ora.step, ora.reset = ora.step_numpy, (lambda mask=None: ora.reset_numpy())
    rs = np.random.default_rng(5)

    def check(where, plain_sum=False):
        st = eng.episode_stats()
        assert st["episodes"] == int(ora.ep_count.sum()), where
        assert st["return_sum"] == ora.return_sum, (where, st["return_sum"], ora.return_sum)
        if plain_sum:
            assert st["return_sum"] == ora.finished_return_sum, where
        ro = parse_stats(eng.episode_stats_device(returns_only=True))
        assert (ro["episodes"], ro["illegal_ends"], ro["return_sum"]) == (st["episodes"], st["illegal_ends"], st["return_sum"]), where
        assert np.array_equal(eng.get_scores(), ora.score), where

    eng.reset()
    ora.reset()
    check("reset")
    for s in range(40):                                          # auto-reset: the plain per-episode sum
        a = rs.integers(0, 4, n).astype(np.uint8)
        eng.step(a)
        ora.step(a)
        if s % 8 == 7:
            check(f"auto-reset step {s}", plain_sum=True)
    assert ora.finished_return_sum > 0
    for s in range(30):                                          # no auto-reset: ended episodes stay on the board (pending)
        a = rs.integers(0, 4, n).astype(np.uint8)
        eng.step(a, auto_reset=False)
        ora.step(a, auto_reset=False)
        if s % 6 == 5:
            check(f"no-reset step {s}")
    assert ora.pending.any()
    if not numpy_mode:
        mask = (rs.random(n) < 0.4).astype(np.uint8)             # resets some pending, some running (abandoned) boards
        eng.reset(mask=mask)
        ora.reset(mask=mask)
        check("masked reset")
    sc = rs.integers(0, 1 << 22, n).astype(np.int32)
    eng.set_scores(sc)
    ora.set_scores(sc)
    check("set_scores (host)")
    sc = rs.integers(0, 1 << 22, n).astype(np.int32)
    eng.set_scores(torch.as_tensor(sc).to(eng.device))
    ora.set_scores(sc)
    check("set_scores (device)")
    b = (rs.integers(0, 12, (n, 16)) * (rs.random((n, 16)) < 0.6)).astype(np.uint8)
    eng.set_boards(b)
    ora.boards[:] = b
    check("set_boards")
    k = 9
    acts = torch.as_tensor(rs.integers(0, 4, (k, n)).astype(np.uint8)).to(eng.device)
    for fused, auto in ((False, False), (True, False), (True, True), (False, True)):
        if fused and numpy_mode:
            continue
        eng.rollout(acts, auto_reset=auto, fused=fused)
        for j in range(k):
            ora.step(acts[j].cpu().numpy(), auto_reset=auto)
        check(f"rollout fused={fused} auto_reset={auto}")
    if not numpy_mode:
        eng.rollout_random(37)
        for _ in range(37):
            ora.step(None)
        check("rollout_random")
    eng.reset()                                                  # full reset: pending marks cleared, running episodes abandoned
    ora.reset()
    check("full reset")
    blob = eng.state_dict()
    other = Batched2048(n, seed=1, rng=rng)
    other.load_state_dict(blob)
    other.set_illegal_move_reward(-1.0)
    eng.close()
    eng = other
    check("state round trip")
    eng.seed(seed + 1)                                           # the books restart: nothing finished, live scores carried
    ora.seed_numpy(seed + 1) if numpy_mode else ora.seed_(seed + 1)
    ora.ep_count[:] = 0
    check("reseed")
    assert eng.episode_stats()["return_sum"] == 0
    for s in range(12):
        a = rs.integers(0, 4, n).astype(np.uint8)
        eng.step(a)
        ora.step(a)
    check("after reseed")
    st = eng.episode_stats()
    assert st["mean_episode_return"] == (st["return_sum"] - st["illegal_ends"]) / st["episodes"]     # illegal_move_reward = -1


def test_episode_slot_carries_into_the_high_dwords(torch_cuda):
    """A launch stores only the LOW dwords of its wavefront's episode slot; the high halves follow on a carry, once in
    2^32 counts -- a path no ordinary test reaches.  Here the slots of a checkpoint are patched to sit a few counts
    below 2^32 (episodes, illegal ends and the gain total G), loaded back, and the next steps must carry: the statistics
    move by exactly what the oracle counts from there on.  Also: a fused random rollout longer than the 1 024-step fold of
    its per-lane 32-bit gain accumulator."""
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048
    from oracle import OracleBatch
    n, seed = 4096, 31
    eng, ora = Batched2048(n, seed=seed), OracleBatch(n, seed)
    eng.reset()
    ora.reset()
    for _ in range(20):
        eng.step(None)
        ora.step(None)
    st0 = eng.episode_stats()
    assert st0["return_sum"] == ora.finished_return_sum and st0["episodes"] == int(ora.ep_count.sum())
    state = eng.state_dict()
    blob = state["blob"].copy()
    header, up = 56, lambda x: (x + 255) & ~255             # StateHeader; slab = records | terminal records | slots | stats
    slots = blob[header + 2 * up(n * 16): header + 2 * up(n * 16) + (n // 64) * 32].view(np.uint32).reshape(n // 64, 8)
    old = slots.astype(np.int64)
    assert not slots[:, 4:7].any()                           # nothing has carried yet
    near = np.uint32(0xFFFFFFF0)
    slots[:, 0:3] = near                                     # episodes.lo, illegal_ends.lo, G.lo of every wavefront
    shift = ((int(near) - old[:, 0]).sum(), (int(near) - old[:, 1]).sum(), (int(near) - old[:, 2]).sum())
    eng.load_state_dict(dict(state, blob=blob))
    st1 = eng.episode_stats()
    assert st1["episodes"] == st0["episodes"] + shift[0] and st1["illegal_ends"] == st0["illegal_ends"] + shift[1]
    assert st1["return_sum"] == st0["return_sum"] + shift[2]
    ep0, ret0, ill0 = int(ora.ep_count.sum()), ora.finished_return_sum, 0
    for s in range(30):                                      # every wavefront passes 2^32 in all three counters
        eng.step(None)
        ora.step(None)
        ill0 += int((ora.illegal.astype(bool) & ora.terminated.astype(bool)).sum())
    st2 = eng.episode_stats()
    assert st2["episodes"] - st1["episodes"] == int(ora.ep_count.sum()) - ep0 > 64 * 30 // 20
    assert st2["illegal_ends"] - st1["illegal_ends"] == ill0
    assert st2["return_sum"] - st1["return_sum"] == ora.finished_return_sum - ret0 > 0
    raw = eng.state_dict()["blob"][header + 2 * up(n * 16): header + 2 * up(n * 16) + (n // 64) * 32].view(np.uint32).reshape(-1, 8)
    assert (raw[:, 4] == 1).all() and (raw[:, 6] == 1).all() and (raw[:, 5] == 1).all()      # carried, once
    # ---- a fused rollout across the accumulator fold
    eng2, ora2 = Batched2048(512, seed=seed), OracleBatch(512, seed)
    eng2.reset()
    ora2.reset()
    eng2.rollout_random(2500)
    for _ in range(2500):
        ora2.step(None)
    st = eng2.episode_stats()
    assert st["return_sum"] == ora2.finished_return_sum == ora2.return_sum and st["episodes"] == int(ora2.ep_count.sum())
    assert np.array_equal(eng2.get_boards().reshape(512, 16), ora2.boards) and np.array_equal(eng2.get_scores(), ora2.score)


@pytest.mark.parametrize("n", [512, 777, 5000, 65536 + 300])
def test_two_chain_rollout_is_bit_identical(torch_cuda, n):
    """g2048_set_chains(2): g2048_rollout cuts the batch at a block boundary and runs the halves as two chains of
    launches (caller's stream / engine's side stream, two host threads).  Boards are independent, so EVERYTHING must be
    bit-identical to one chain and to the oracle: every per-step output in [k, n] buffers (reward, terminated, illegal,
    highest, terminal boards, plain boards, the fused observation), every action dtype, the state, the statistics and
    the terminal records -- for batches whose second half is ragged, with work enqueued right before and right behind
    the call on the caller's stream (fork / join), and interleaved with single-chain calls."""
    torch = torch_cuda
    from gym2048_amd import _lib
    from gym2048_amd.batched import Batched2048
    from oracle import OracleBatch
    import ctypes as C
    seed, k = 21, 70                                    # (a cold engine runs rollouts shorter than 64 steps as one chain)
    two, one, ora = Batched2048(n, seed=seed, chains=2), Batched2048(n, seed=seed, chains=1), OracleBatch(n, seed)
    assert two.chains == 2 and one.chains == 1
    for e in (two, one, ora):
        e.reset()
    rs = np.random.default_rng(n)
    dev = two.device

    def buffers():
        return dict(reward=torch.zeros((k, n), dtype=torch.float32, device=dev), terminated=torch.zeros((k, n), dtype=torch.uint8, device=dev),
                    illegal=torch.zeros((k, n), dtype=torch.uint8, device=dev), highest=torch.zeros((k, n), dtype=torch.uint8, device=dev),
                    terminal_boards=torch.zeros((k, n, 16), dtype=torch.uint8, device=dev),
                    obs=torch.zeros((k, n, 16, 4, 4), dtype=torch.uint8, device=dev))

    for rep, adt in enumerate((torch.uint8, torch.int32, torch.int64, None)):
        acts = None if adt is None else torch.as_tensor(rs.integers(0, 4, (k, n))).to(dev).to(adt)
        b2, b1 = buffers(), buffers()
        bo2 = torch.zeros((k, n, 16), dtype=torch.uint8, device=dev)
        for eng, b, bo in ((two, b2, bo2), (one, b1, None)):
            a = eng.random_actions(k) if acts is None else acts
            if bo is None:
                eng.rollout(a, **b)
            else:       # boards_out rides along through the raw ABI (the Python rollout has no argument for it)
                io = eng._io(a, b["reward"], b["terminated"], b["illegal"], b["highest"], b["terminal_boards"], b["obs"])
                io.boards_out = bo.data_ptr()
                marker = torch.ones(1 << 20, device=dev).cumsum(0)        # work in flight on the caller's stream: the fork waits for it
                _lib.check(eng._lib.g2048_rollout(eng._h, k, C.byref(io), n, 1, eng._stream()))
                after = b["reward"].sum()                                   # enqueued behind the call: must see BOTH halves (join)
        host_a = (one.random_actions(k, t_first=one.clock - k + 1) if acts is None else acts).cpu().numpy() & 3
        total = 0.0
        for j in range(k):
            ora.step(host_a[j].astype(np.uint8))
            total += float(ora.reward.sum())
            for name in ("reward", "terminated", "illegal", "highest"):
                want = getattr(ora, name)
                assert np.array_equal(b2[name][j].cpu().numpy(), want), (name, rep, j)
                assert np.array_equal(b1[name][j].cpu().numpy(), want), (name, rep, j)
            done = ora.terminated.astype(bool)
            assert np.array_equal(b2["terminal_boards"][j].cpu().numpy()[done], ora.terminal_boards[done]), (rep, j)
            assert np.array_equal(b2["obs"][j].cpu().numpy(), ora.onehot()), (rep, j)
            assert np.array_equal(bo2[j].cpu().numpy(), ora.boards), (rep, j)
        assert float(after) == total
        assert torch.equal(b2["obs"], b1["obs"]) and torch.equal(b2["terminal_boards"], b1["terminal_boards"])
        for eng in (two, one):
            assert np.array_equal(eng.get_boards().reshape(n, 16), ora.boards) and np.array_equal(eng.get_scores(), ora.score)
            assert np.array_equal(eng.get_last_scores(), ora.last_score)
        s2, s1 = two.episode_stats(), one.episode_stats()
        assert s2 == s1 and s2["return_sum"] == ora.finished_return_sum and s2["episodes"] == int(ora.ep_count.sum())
        assert two.clock == one.clock == ora.t
        two.step(None)                                             # a single-chain call in between
        one.step(None)
        ora.step(None)
        # a SHORT rollout right behind a long one: the side chain is warm, so 16 steps go out as two chains as well
        r2, r1 = torch.zeros((16, n), dtype=torch.float32, device=dev), torch.zeros((16, n), dtype=torch.float32, device=dev)
        two.rollout(16, reward=r2)
        one.rollout(16, reward=r1)
        for j in range(16):
            ora.step(None)
            assert np.array_equal(r2[j].cpu().numpy(), ora.reward) and np.array_equal(r1[j].cpu().numpy(), ora.reward), (rep, j)
    assert torch.equal(two.records(), one.records()) and torch.equal(two.last_records(), one.last_records())
    two.close()
    one.close()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_two_chain_rollout_on_whatever_stream_the_caller_brings(torch_cuda):
    """The caller's stream of a two-chain rollout can be any stream: the legacy default stream, torch's current stream,
    one of MANY pool streams in use at once (more streams than hardware queues), or a HIGH-priority stream -- which has
    the side stream's own priority and may share its hardware queue, so that rollout must go out as one chain
    (g2048_get_chains_used) instead of parking a ticket wait in front of its ticket.  All bit-identical, none slow."""
    torch = torch_cuda
    import time
    from gym2048_amd.batched import Batched2048
    n, k, seed = 1 << 16, 70, 5
    ref = Batched2048(n, seed=seed, chains=1)
    ref.reset()
    want = torch.zeros((k, n), dtype=torch.float32, device=ref.device)
    ref.rollout(k, reward=want)
    torch.cuda.synchronize()
    crowd = [torch.cuda.Stream() for _ in range(12)]
    high = torch.cuda.Stream(priority=-1)
    cases = [("default", None, 2)] + [("pool%d" % i, st, 2) for i, st in enumerate(crowd[:3])] + [("high", high, 1)]
    for name, st, chains in cases:
        eng = Batched2048(n, seed=seed, chains=2)
        got = torch.zeros((k, n), dtype=torch.float32, device=eng.device)
        junk = [torch.zeros(1 << 18, device=eng.device) for _ in crowd]
        t0 = time.perf_counter()
        ctx = torch.cuda.stream(st) if st is not None else torch.cuda.stream(torch.cuda.default_stream())
        with ctx:
            eng.reset()
            for c, j in zip(crowd, junk):          # every pool stream busy while the rollout is enqueued
                with torch.cuda.stream(c):
                    j.add_(1.0).cumsum_(0)
            eng.rollout(k, reward=got)
            total = got.sum()
        torch.cuda.synchronize()
        took = time.perf_counter() - t0
        assert eng.chains_used == chains, (name, eng.chains_used)
        assert torch.equal(got, want) and float(total) == float(want.sum()), name
        assert took < 5.0, (name, took)            # (a ticket wait that never sees its ticket gives up after ~a minute)
        eng.close()
    ref.close()


def test_two_chain_ticket_that_does_not_arrive_in_time_is_loud(torch_cuda, monkeypatch):
    """The ticket waits of a two-chain rollout are bounded, and a wait that runs out must not pass silently (its chain
    then runs unordered against the other one): flag_wait_kernel reports through a pinned host word and EVERY later call
    on the engine fails with G2048_ERR_HIP.  Provoked with a tiny bound (G2048_FLAG_WAIT_POLLS, read by g2048_set_chains)
    and a caller's stream that is busy for tens of milliseconds in front of the fork ticket."""
    torch = torch_cuda
    import time
    from gym2048_amd._lib import G2048Error
    from gym2048_amd.batched import Batched2048
    monkeypatch.setenv("G2048_TWO_CHAIN_MIN_STEPS", "2")
    n, k = 4096, 8
    rew = torch.zeros((k, n), dtype=torch.float32, device="cuda")
    # a healthy engine first: the same rollout with the default bound is fine
    ok = Batched2048(n, seed=3, chains=2)
    ok.reset()
    ok.rollout(k, reward=rew)
    torch.cuda.synchronize()
    assert ok.chains_used == 2 and ok.episode_stats()["episodes"] >= 0
    ok.close()
    # how long does torch's spin kernel take per million cycles on this box?  (aim at ~60 ms in front of the ticket)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    torch.cuda._sleep(1_000_000)
    torch.cuda.synchronize()
    per_million = max(time.perf_counter() - t0, 1e-5)
    cycles = int(min(0.06 / per_million, 400.0) * 1_000_000)
    monkeypatch.setenv("G2048_FLAG_WAIT_POLLS", "300")           # ~0.3-1 ms
    eng = Batched2048(n, seed=3, chains=2)
    eng.reset()
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        torch.cuda._sleep(cycles)                                # the fork ticket is queued behind this
        eng.rollout(k, reward=rew)                               # the call itself cannot know yet
    assert eng.chains_used == 2
    torch.cuda.synchronize()
    for call in (lambda: eng.step(None), eng.get_boards, eng.episode_stats, lambda: eng.rollout(k, reward=rew), eng.reset):
        with pytest.raises(G2048Error, match="ordering ticket"):
            call()
    eng.close()                                                  # destroying it is what is left to do, and works


@pytest.mark.parametrize("n,adt", [(65536, "uint8"), (5000, "int64"), (1 << 17, "int32")])
def test_small_batch_rollouts_are_replayed_from_a_cached_graph(torch_cuda, monkeypatch, n, adt):
    """BASELINE configs[1] territory (<= 2^17 boards): the second and every further g2048_rollout in a row over the same
    buffers is a hipGraph replay of its launch train (step_graph_kernel reads the clock from device memory).  Same games:
    every replay's rewards and flags, the boards, scores and the return accounting against the oracle; whole-block and
    ragged batches, every action dtype, a non-default stream; a single step between two replays (the clock moves on); a
    change of what the graph froze (terminal records on) falls back to stream launches once and rebuilds; the knob."""
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048
    from oracle import OracleBatch
    seed, k = 9, 24
    eng, ora = Batched2048(n, seed=seed, last_records=False), OracleBatch(n, seed, threads=0)
    eng.reset()
    ora.reset()
    dev = eng.device
    acts8 = torch.zeros((k, n), dtype=torch.uint8, device=dev)
    acts = torch.zeros((k, n), dtype=getattr(torch, adt), device=dev)
    rew = torch.zeros((k, n), dtype=torch.float32, device=dev)
    term = torch.zeros((k, n), dtype=torch.uint8, device=dev)
    stream = torch.cuda.Stream()

    def one_rollout(expect_replays):
        with torch.cuda.stream(stream):
            eng.random_actions(k, out=acts8)
            acts.copy_(acts8)                               # same buffer every time: that is what makes the train replayable
            rew.zero_()
            term.zero_()
            plan.run()
        stream.synchronize()
        for j in range(k):
            ora.step(None)
            assert np.array_equal(rew[j].cpu().numpy(), ora.reward), (j, expect_replays)
            assert np.array_equal(term[j].cpu().numpy(), ora.terminated), (j, expect_replays)
        assert eng.graph_replays == expect_replays

    with torch.cuda.stream(stream):
        plan = eng.prepare_rollout(acts, reward=rew, terminated=term)
    one_rollout(0)                                          # first sight of these buffers: launched kernel by kernel
    one_rollout(1)                                          # second in a row: graph built, replayed
    one_rollout(2)
    eng.step(None)                                          # something else in between: the clock the next replay gets has moved
    ora.step(None)
    one_rollout(3)
    assert np.array_equal(eng.get_boards().reshape(n, 16), ora.boards) and np.array_equal(eng.get_scores(), ora.score)
    eng.set_last_records(True)                              # the graph froze "no terminal records": not this rollout's train any more
    one_rollout(3)
    one_rollout(4)                                          # rebuilt
    st = eng.episode_stats()
    assert st["episodes"] == int(ora.ep_count.sum()) and st["return_sum"] == ora.return_sum == ora.finished_return_sum
    assert np.array_equal(eng.get_boards().reshape(n, 16), ora.boards) and np.array_equal(eng.get_scores(), ora.score)
    eng.close()
    # the knob: read by g2048_create
    monkeypatch.setenv("G2048_ROLLOUT_GRAPH", "0")
    off = Batched2048(n, seed=seed)
    off.reset()
    for _ in range(3):
        off.rollout(acts8, reward=rew, terminated=term)
    torch.cuda.synchronize()
    assert off.graph_replays == 0
    off.close()
    monkeypatch.delenv("G2048_ROLLOUT_GRAPH")
    # optional outputs are not the standard train: never replayed
    opt = Batched2048(n, seed=seed)
    opt.reset()
    ill = torch.zeros((k, n), dtype=torch.uint8, device=dev)
    for _ in range(3):
        opt.rollout(acts8, reward=rew, terminated=term, illegal=ill)
    torch.cuda.synchronize()
    assert opt.graph_replays == 0
    opt.close()
    # a PREPARED plan is a replay from its first run on (g2048_rollout_prepare executes nothing), four plans are kept
    pre, twin = Batched2048(n, seed=seed), OracleBatch(n, seed, threads=0)
    pre.reset()
    twin.reset()
    bufs = [(torch.zeros((k, n), dtype=torch.float32, device=dev), torch.zeros((k, n), dtype=torch.uint8, device=dev)) for _ in range(4)]
    plans = [pre.prepare_rollout(k, reward=r, terminated=t).prepare_graph() for r, t in bufs]     # the synthetic policy (ACT 0)
    assert pre.clock == 0 and pre.graph_replays == 0
    for rep in range(2):
        for q, (r, t) in enumerate(bufs):
            plans[q].run()
            torch.cuda.synchronize()
            for j in range(k):
                twin.step(None)
                assert np.array_equal(r[j].cpu().numpy(), twin.reward) and np.array_equal(t[j].cpu().numpy(), twin.terminated), (rep, q, j)
    assert pre.graph_replays == 8
    assert np.array_equal(pre.get_boards().reshape(n, 16), twin.boards) and np.array_equal(pre.get_scores(), twin.score)
    # replays enqueued BACK TO BACK, nothing synchronised in between -- the same graph twice and all four in a row: every
    # replay must see its own clock (it is written by an ordinary launch in front of each graph launch, not by rewriting
    # the executable graph's parameters under an earlier launch's feet)
    with torch.cuda.stream(stream):
        plans[0].run()
        first = bufs[0][0].clone()
        plans[0].run()
        for q in (1, 2, 3):
            plans[q].run()
    stream.synchronize()
    assert pre.graph_replays == 13
    want = []
    for _ in range(5):
        rows = []
        for j in range(k):
            twin.step(None)
            rows.append(twin.reward.copy())
        want.append(np.stack(rows))
    assert np.array_equal(first.cpu().numpy(), want[0]) and np.array_equal(bufs[0][0].cpu().numpy(), want[1])
    for q in (1, 2, 3):
        assert np.array_equal(bufs[q][0].cpu().numpy(), want[1 + q]), q
    assert np.array_equal(pre.get_boards().reshape(n, 16), twin.boards) and np.array_equal(pre.get_scores(), twin.score)
    pre.close()


def test_rollout_writes_terminal_boards(torch_cuda):
    """terminal_boards through g2048_rollout: row [j, i] is written exactly where step j ended board i's
    episode and holds the board the episode ended on."""
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048
    from oracle import OracleBatch
    n, k, seed = 5000, 30, 4
    eng, ora = Batched2048(n, seed=seed), OracleBatch(n, seed, threads=0)
    eng.reset()
    ora.reset()
    acts = eng.random_actions(k)
    term = torch.zeros((k, n), dtype=torch.uint8, device=eng.device)
    tb = torch.full((k, n, 16), 0xEE, dtype=torch.uint8, device=eng.device)
    eng.rollout(acts, terminated=term, terminal_boards=tb)
    tb_h, term_h = tb.cpu().numpy(), term.cpu().numpy()
    for j in range(k):
        ora.step(None)
        done = ora.terminated.astype(bool)
        assert np.array_equal(term_h[j], ora.terminated)
        assert np.array_equal(tb_h[j][done], ora.terminal_boards[done])
        assert (tb_h[j][~done] == 0xEE).all()                  # untouched where the episode goes on
    with pytest.raises(Exception):
        eng.rollout(acts, terminal_boards=tb, fused=True)


def test_fill_random_actions_walks_the_step_axis_in_slabs(torch_cuda):
    """More than 32 768 steps: the launcher splits gridDim.y (g2048_kernels.hip launch_fill_actions)."""
    from gym2048_amd.batched import Batched2048
    from oracle.cpu_ref import random_actions_np
    n, k, seed, off = 8, 70001, 6, 12345
    eng = Batched2048(n, seed=seed, board_offset=off)
    a = eng.random_actions(k, t_first=5).cpu().numpy()
    assert a.shape == (k, n)
    for j in (0, 1, 32767, 32768, 32769, 65535, 65536, 70000):
        assert np.array_equal(a[j], random_actions_np(seed, 5 + j, off, n)), j


def test_canonicalize_vs_reference_table_and_oracle(torch_cuda):
    from gym2048_amd.transitions import canonicalize_device
    from oracle.cpu_ref import canonicalize
    c = load_golden("canonical_table")
    b, nb, a, sym = canonicalize_device(c["boards"], c["actions"], c["next_boards"])
    assert np.array_equal(sym, c["symmetry"]) and np.array_equal(a, c["canon_actions"])
    assert np.array_equal(b, c["canon_boards"]) and np.array_equal(nb, c["canon_next"])
    # idempotent; boards only
    b2, _, _, sym2 = canonicalize_device(b)
    assert np.array_equal(b2, b) and not sym2.any()
    rng = np.random.default_rng(1)
    boards = _random_boards(rng, 20000)
    got, _, acts, sym = canonicalize_device(boards, rng.integers(0, 4, 20000))
    for i in range(0, 20000, 97):
        wb, _, _, wv = canonicalize(boards[i].reshape(4, 4))
        assert wv == sym[i] and np.array_equal(wb.reshape(16), got[i])
    # the engine-less entry points find their device from the pointer; a HOST buffer is refused, not dereferenced
    from gym2048_amd import _lib
    lib = _lib.load()
    host = np.zeros((64, 16), np.uint8)
    host_actions = np.zeros(64, np.uint8)
    assert lib.g2048_canonicalize(host.ctypes.data, None, None, 64, None, None) == -1
    assert b"pageable host memory" in lib.g2048_last_error()
    out = np.zeros((8 * 64, 16), np.uint8)
    assert lib.g2048_augment(host.ctypes.data, None, host_actions.ctypes.data, 64, out.ctypes.data, None,
                             np.zeros(8 * 64, np.uint8).ctypes.data, None) == -1


def test_engine_less_entry_points_accept_pinned_host_buffers_and_keep_the_device(torch_cuda):
    """g2048_augment / g2048_canonicalize need no engine: they launch on the device their buffers live on, accept
    pinned (device-mapped) host memory next to device memory, refuse pageable host memory with a message, and leave
    the caller's current device as it was."""
    import ctypes as C
    torch = torch_cuda
    from gym2048_amd import _lib
    lib = _lib.load()
    n = 512
    rs = np.random.default_rng(3)
    boards = rs.integers(0, 12, (n, 16)).astype(np.uint8)
    acts = rs.integers(0, 4, n).astype(np.uint8)
    dev_b, dev_a = torch.as_tensor(boards).cuda(), torch.as_tensor(acts).cuda()
    out_b = torch.zeros((8, n, 16), dtype=torch.uint8, device="cuda")
    out_a = torch.zeros((8, n), dtype=torch.uint8, device="cuda")
    _lib.check(lib.g2048_augment(dev_b.data_ptr(), None, dev_a.data_ptr(), n, out_b.data_ptr(), None, out_a.data_ptr(), None))
    torch.cuda.synchronize()
    pin_b, pin_a = torch.as_tensor(boards).pin_memory(), torch.as_tensor(acts).pin_memory()
    pout_b = torch.zeros((8, n, 16), dtype=torch.uint8).pin_memory()
    pout_a = torch.zeros((8, n), dtype=torch.uint8).pin_memory()
    before = torch.cuda.current_device()
    _lib.check(lib.g2048_augment(pin_b.data_ptr(), None, pin_a.data_ptr(), n, pout_b.data_ptr(), None, pout_a.data_ptr(), None))
    torch.cuda.synchronize()
    assert torch.cuda.current_device() == before
    assert torch.equal(pout_b, out_b.cpu()) and torch.equal(pout_a, out_a.cpu())
    # mixed: pinned input, device output
    out_b.zero_()
    _lib.check(lib.g2048_augment(pin_b.data_ptr(), None, dev_a.data_ptr(), n, out_b.data_ptr(), None, out_a.data_ptr(), None))
    torch.cuda.synchronize()
    assert torch.equal(pout_b, out_b.cpu())
    pageable = np.ascontiguousarray(boards)
    rc = lib.g2048_augment(pageable.ctypes.data, None, dev_a.data_ptr(), n, out_b.data_ptr(), None, out_a.data_ptr(), None)
    assert rc != 0 and b"pageable" in lib.g2048_last_error()
    canon = pin_b.clone().pin_memory()
    _lib.check(lib.g2048_canonicalize(canon.data_ptr(), None, None, n, None, None))
    torch.cuda.synchronize()
    dcanon = dev_b.clone()
    _lib.check(lib.g2048_canonicalize(dcanon.data_ptr(), None, None, n, None, None))
    torch.cuda.synchronize()
    assert torch.equal(canon, dcanon.cpu())


def test_collective_behind_the_c_abi_one_rank(torch_cuda):
    """g2048_comm_* / g2048_allgather_returns with RCCL loaded by the library itself: a one-rank communicator
    (the degenerate case a 1-GPU box can run) gathers the engine's returns; the single-process multi-engine
    form (persistent g2048_comm_local) with one engine does the same, repeatedly."""
    torch = torch_cuda
    from gym2048_amd import _lib
    from gym2048_amd.batched import Batched2048
    lib = _lib.load()
    n = 1 << 16
    eng = Batched2048(n, seed=3)
    eng.reset()
    eng.rollout(64)
    want = eng.get_last_scores()
    assert (want > 0).any()
    ident = (C.c_uint8 * _lib.COMM_ID_BYTES)()
    _lib.check(lib.g2048_comm_unique_id(ident))
    comm = C.c_void_p()
    _lib.check(lib.g2048_comm_create(1, 0, ident, 0, C.byref(comm)))
    out = torch.zeros(n, dtype=torch.int32, device=eng.device)
    stream = C.c_void_p(torch.cuda.current_stream(eng.device).cuda_stream)
    _lib.check(lib.g2048_allgather_returns(eng._h, comm, out.data_ptr(), stream))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)
    _lib.check(lib.g2048_comm_destroy(comm))
    # single-process form: the communicator set is built ONCE and reused by every rollout's all-gather
    local = C.c_void_p()
    devices = (C.c_int * 1)(0)
    _lib.check(lib.g2048_comm_local_create(devices, 1, C.byref(local)))
    engines = (C.c_void_p * 1)(eng._h)
    streams = (C.c_void_p * 1)(stream.value)
    for rep in range(3):
        out2 = torch.zeros(n, dtype=torch.int32, device=eng.device)
        outs = (C.c_void_p * 1)(out2.data_ptr())
        _lib.check(lib.g2048_allgather_returns_local(local, engines, outs, streams))   # enqueued, not synchronised
        torch.cuda.synchronize()
        assert np.array_equal(out2.cpu().numpy(), want)
        eng.rollout(8)
        want = eng.get_last_scores()
    _lib.check(lib.g2048_comm_local_destroy(local))
    dup = (C.c_int * 2)(0, 0)
    assert lib.g2048_comm_local_create(dup, 2, C.byref(local)) != 0         # one communicator per device
    assert lib.g2048_comm_create(2, 5, ident, 0, C.byref(comm)) != 0     # rank out of range


def test_local_shards_single_process_form(torch_cuda):
    """gym2048_amd.LocalShards (one process, several GPUs; here the one GPU there is): the shard plays the same
    games as a plain engine and the all-gather through the persistent communicator returns its last returns."""
    torch = torch_cuda
    from gym2048_amd import LocalShards
    from gym2048_amd.batched import Batched2048
    n, seed = 1 << 15, 12
    sh = LocalShards(n, devices=[0], seed=seed)
    ref = Batched2048(n, seed=seed)
    sh.reset()
    ref.reset()
    for _ in range(3):
        sh.rollout_random(40)
        ref.rollout_random(40)
        outs = sh.allgather_returns()
        sh.synchronize()
        assert len(outs) == 1 and np.array_equal(outs[0].cpu().numpy(), ref.get_last_scores())
    acts = ref.random_actions(6)
    rew = torch.zeros((6, n), dtype=torch.float32, device=ref.device)
    rew2 = torch.zeros_like(rew)
    sh.rollout([acts], reward=[rew2])
    ref.rollout(acts, reward=rew)
    assert torch.equal(rew, rew2) and np.array_equal(sh.engines[0].get_boards(), ref.get_boards())
    assert sh.n_global == n
    sh.close()


def test_local_shards_one_launch_thread_per_device(torch_cuda):
    """SURVEY 8e's scaling risk -- a serial host loop over the devices delays the last device's first kernel by all
    the launches issued before it.  LocalShards gives every engine its own launch thread: with two engines (both on
    the one GPU there is, each on its own stream) the two launch trains are ISSUED concurrently (their host windows
    overlap; in the serial form engine 1 starts when engine 0's train has been issued), and the games are the same."""
    torch = torch_cuda
    from gym2048_amd import LocalShards
    n, seed, k = 1 << 14, 9, 1500
    runs = {}
    for threaded in (False, True):
        streams = [torch.cuda.Stream(device=0), torch.cuda.Stream(device=0)]
        sh = LocalShards(n, devices=[0, 0], seed=seed, threads=threaded, streams=streams)
        sh.reset()
        acts = [e.random_actions(k) for e in sh.engines]
        rew = [torch.zeros((k, n), dtype=torch.float32, device="cuda:0") for _ in sh.engines]
        term = [torch.zeros((k, n), dtype=torch.uint8, device="cuda:0") for _ in sh.engines]
        torch.cuda.synchronize()
        plans = sh.prepare_rollout(acts, reward=rew, terminated=term)
        sh.run_plans(plans)                              # k launches per engine
        (a0, b0), (a1, b1) = sh.issue_windows
        sh.synchronize()
        runs[threaded] = dict(boards=[e.get_boards() for e in sh.engines], rew=[r.cpu() for r in rew],
                              term=[t.cpu() for t in term], stats=[e.episode_stats() for e in sh.engines],
                              overlap=min(b0, b1) - max(a0, a1), span=max(b0, b1) - min(a0, a1), w=(a0, b0, a1, b1))
        with pytest.raises(Exception):                   # one communicator per DEVICE: two engines on cuda:0 cannot gather
            sh.allgather_returns()
        sh.close()
    ser, thr = runs[False], runs[True]
    assert ser["overlap"] <= 0, ser["w"]                  # serial form: engine 1's train is issued after engine 0's
    assert thr["overlap"] > 0.5 * min(thr["w"][1] - thr["w"][0], thr["w"][3] - thr["w"][2]), thr["w"]   # concurrent issue
    print(f"issue span for 2 x {k} launches: serial {ser['span'] * 1e3:.2f} ms, one thread per engine {thr['span'] * 1e3:.2f} ms")
    for r in range(2):
        assert np.array_equal(ser["boards"][r], thr["boards"][r]) and ser["stats"][r] == thr["stats"][r]
        assert torch.equal(ser["rew"][r], thr["rew"][r]) and torch.equal(ser["term"][r], thr["term"][r])
    # and engine 1 (global boards n .. 2n-1) is not a copy of engine 0
    assert not np.array_equal(thr["boards"][0], thr["boards"][1])


def test_misaligned_buffers_are_refused(torch_cuda):
    torch = torch_cuda
    from gym2048_amd import _lib
    from gym2048_amd.batched import Batched2048
    e = Batched2048(256, seed=1)
    e.reset()
    buf = torch.zeros(256 * 4 + 64, dtype=torch.uint8, device=e.device)
    io = _lib.StepIO()
    io.action_dtype = _lib.ACT_RANDOM
    io.reward = buf.data_ptr() + 2                     # float32 output on a 2-byte boundary
    assert e._lib.g2048_step(e._h, C.byref(io), 1, None) == -1 and b"misaligned" in e._lib.g2048_last_error()
    io.reward, io.terminal_boards = None, buf.data_ptr() + 8
    assert e._lib.g2048_step(e._h, C.byref(io), 1, None) == -1
    io.terminal_boards, io.actions, io.action_dtype = None, buf.data_ptr() + 4, _lib.ACT_I64
    assert e._lib.g2048_step(e._h, C.byref(io), 1, None) == -1
    assert e.clock == 0                                 # nothing was stepped
    # device-side views of boards / scores use natural-width stores as well
    lib = e._lib
    assert lib.g2048_get_boards(e._h, C.c_void_p(buf.data_ptr() + 8), None) == -1 and b"aligned" in lib.g2048_last_error()
    assert lib.g2048_set_boards(e._h, C.c_void_p(buf.data_ptr() + 8), None) == -1
    for fn in (lib.g2048_get_scores, lib.g2048_set_scores, lib.g2048_get_last_scores):
        assert fn(e._h, C.c_void_p(buf.data_ptr() + 2), None) == -1 and b"aligned" in lib.g2048_last_error()
    assert lib.g2048_move(e._h, C.c_void_p(buf.data_ptr() + 4), _lib.ACT_I64, 1, None, None, None) == -1
    assert lib.g2048_move(e._h, C.c_void_p(buf.data_ptr()), _lib.ACT_U8, 1, C.c_void_p(buf.data_ptr() + 2), None, None) == -1


@pytest.mark.parametrize("args", [("5000", "40", "42", "0"), ("65536", "24", "7", str((1 << 32) - 65536))])
def test_pure_c_client_of_the_abi(torch_cuda, args):
    """tests/c_abi_client/client.cpp: includes only include/g2048.h (+ the oracle's header as checker), raw
    hipMalloc'ed buffers, g2048_create / reset / fill_random_actions / rollout / get_* / episode_stats, and
    compares everything with the C oracle itself -- the C ABI is usable without Python or torch."""
    import subprocess
    import __graft_entry__ as ge
    exe = ge.build_c_client()
    out = subprocess.run([exe, *args], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr


def test_host_side_record_decoding_and_render(torch_cuda):
    """decode_record (Python reading of the record format) == the device's export kernels; render() of one board
    of a big batch moves 16 bytes, and prints what the single-env renderer prints."""
    from gym2048_amd.batched import Batched2048, decode_record, exp_to_values
    from gym2048_amd.render import render_board
    n = 1 << 16
    eng = Batched2048(n, seed=12)
    eng.reset()
    eng.rollout(300)
    boards, scores, raw = eng.get_boards().reshape(n, 16), eng.get_scores(), eng.records().cpu().numpy()
    for i in list(range(0, n, 997)) + [n - 1]:
        cells, score = decode_record(raw[i])
        assert np.array_equal(cells, boards[i]) and score == scores[i]
    i = int(np.argmax(scores))
    want = render_board(exp_to_values(boards[i].reshape(4, 4)), int(scores[i]), "ansi").getvalue()
    assert eng.render(i, "ansi").getvalue() == want and eng.render(i, "rgb_array").shape == (280, 280, 3)
