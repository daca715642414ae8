"""GPU: the HIP path (through the C ABI, via gym2048_amd.Batched2048) against the oracle and the
golden vectors captured from the reference -- bit-exact boards / rewards / flags / scores."""
import numpy as np
import pytest

from conftest import TRAJECTORIES, load_golden, replay_trajectory

pytestmark = pytest.mark.gpu


class GpuBatch:
    """OracleBatch-shaped facade over Batched2048 so the same replay helper drives both."""

    def __init__(self, torch, n, seed, offset):
        from gym2048_amd.batched import Batched2048
        self.torch = torch
        self.eng = Batched2048(n, device=0, seed=seed, board_offset=offset)
        self.n = n
        self.illegal_move_reward = 0.0
        self.max_exp = 0

    def _sync_cfg(self):
        self.eng.set_illegal_move_reward(self.illegal_move_reward)
        self.eng.set_max_tile((1 << self.max_exp) if self.max_exp else None)

    def _pull(self):
        e = self.eng
        self.boards = e.get_boards().reshape(self.n, 16)
        self.score = e.get_scores()
        self.reward = e.reward.cpu().numpy()
        self.terminated = e.terminated.cpu().numpy()
        self.illegal = e.illegal.cpu().numpy()
        self.highest = e.highest.cpu().numpy()
        self.terminal_boards = e.terminal_boards.cpu().numpy()
        self.last_score = e.get_last_scores()

    def reset(self, **kw):
        self._sync_cfg()
        self.eng.reset(**kw)
        self._pull()

    def step(self, actions=None, auto_reset=True):
        if actions is not None:
            actions = self.torch.as_tensor(np.ascontiguousarray(actions, dtype=np.uint8))
        self.eng.step(actions, auto_reset=auto_reset)
        self._pull()


@pytest.mark.parametrize("name", TRAJECTORIES)
def test_golden_trajectories(torch_cuda, name):
    replay_trajectory(lambda n, seed, off: GpuBatch(torch_cuda, n, seed, off), load_golden(name))


def test_move_table(torch_cuda):
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048
    m = load_golden("move_table")
    n = len(m["boards"])
    eng = Batched2048(n)
    for d in range(4):
        for trial in (True, False):
            eng.set_boards(m["boards"])
            score, legal = eng.move(torch.full((n,), d, dtype=torch.int64), trial=trial)
            assert np.array_equal(score.cpu().numpy(), np.where(m["legal"][:, d], m["score"][:, d], 0))
            assert np.array_equal(legal.cpu().numpy(), m["legal"][:, d])
            want = m["boards"] if trial else m["new"][:, d]
            assert np.array_equal(eng.get_boards().reshape(n, 16), want)
    eng.set_boards(m["boards"])
    end, hi = eng.query()
    assert np.array_equal(end.cpu().numpy(), m["isend"]) and np.array_equal(hi.cpu().numpy(), m["highest"])
    # the four trial moves of isend (game2048_env.py:273-280) as one mask, against the reference's own legality flags
    want_mask = (m["legal"].astype(np.uint8) << np.arange(4, dtype=np.uint8)).sum(axis=1).astype(np.uint8)
    assert np.array_equal(eng.legal_actions().cpu().numpy(), want_mask)
    assert np.array_equal(eng.get_boards().reshape(n, 16), m["boards"])          # nothing moved


def test_shift_exhaustive(torch_cuda):
    """All 104 976 rows of shift() (exponents 0..17), four rows per board, as left moves."""
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048
    g = load_golden("shift_exhaustive")
    rows = np.array(np.meshgrid(*[np.arange(18)] * 4, indexing="ij")).reshape(4, -1).T.astype(np.uint8)
    n = len(rows) // 4
    boards = rows.reshape(n, 16)
    eng = Batched2048(n)
    eng.set_boards(boards)
    score, _ = eng.move(torch.full((n,), 3, dtype=torch.uint8))
    assert np.array_equal(eng.get_boards().reshape(-1, 4), g["out"])
    assert np.array_equal(score.cpu().numpy(), g["score"].reshape(n, 4).sum(axis=1))
    # the same rows as columns (up) and reversed (right / down)
    eng.set_boards(np.ascontiguousarray(boards.reshape(n, 4, 4).transpose(0, 2, 1)))
    eng.move(torch.full((n,), 0, dtype=torch.int32))
    assert np.array_equal(eng.get_boards().transpose(0, 2, 1).reshape(-1, 4), g["out"])
    eng.set_boards(np.ascontiguousarray(boards.reshape(n, 4, 4)[:, :, ::-1]))
    eng.move(torch.full((n,), 1, dtype=torch.int64))
    assert np.array_equal(eng.get_boards()[:, :, ::-1].reshape(-1, 4), g["out"])
    eng.set_boards(np.ascontiguousarray(boards.reshape(n, 4, 4).transpose(0, 2, 1)[:, ::-1, :]))
    eng.move(torch.full((n,), 2, dtype=torch.int64))
    assert np.array_equal(eng.get_boards()[:, ::-1, :].transpose(0, 2, 1).reshape(-1, 4), g["out"])


def test_isend_and_csv_fixtures(torch_cuda):
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048
    t = load_golden("isend_table")
    for mx in np.unique(t["max_exp"]):
        sel = t["max_exp"] == mx
        eng = Batched2048(int(sel.sum()), max_tile=(1 << int(mx)) if mx else None)
        eng.set_boards(t["boards"][sel])
        assert np.array_equal(eng.isend_numpy(), t["isend"][sel].astype(bool))
    c = load_golden("test_data_csv")
    eng = Batched2048(848)
    eng.set_boards(c["boards"])
    score, legal = eng.move(torch.as_tensor(c["actions"]))
    assert legal.all() and np.array_equal(score.cpu().numpy(), c["rewards"].astype(np.int32))
    assert np.array_equal(eng.get_boards().reshape(848, 16), c["moved"])
    eng.set_boards(c["next_boards"])
    eng.set_max_tile(2048)
    assert eng.isend_numpy()[-1] and not eng.isend_numpy()[0]


def test_onehot(torch_cuda):
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048
    s = load_golden("stack_table")
    eng = Batched2048(len(s["boards"]))
    eng.set_boards(s["boards"])
    for dt in (torch.uint8, torch.float16, torch.float32):
        got = eng.observe_onehot(dt).cpu().numpy()
        assert np.array_equal(got, s["onehot"].astype(got.dtype))
    assert np.array_equal(eng.tile_values().cpu().numpy().reshape(-1, 16),
                          np.where(s["boards"] > 0, 1 << s["boards"].astype(np.int64), 0))


@pytest.mark.parametrize("n,seed,offset,irw,max_tile,auto_reset", [
    (65536, 42, 0, 0.0, None, True),           # BASELINE config 2 size
    (5000, 7, (1 << 32) - 5000, -1.0, None, True),   # ragged size, top of the index space
    (4097, 3, 12345, -0.5, 64, True),
    (1024, 9, 0, 0.0, None, False),
    (1, 5, 0, 0.0, None, True),                # a single board
])
def test_random_rollout_vs_oracle(torch_cuda, n, seed, offset, irw, max_tile, auto_reset):
    from oracle import OracleBatch
    torch = torch_cuda
    g = GpuBatch(torch, n, seed, offset)
    o = OracleBatch(n, seed, offset, threads=0)
    for b in (g, o):
        b.illegal_move_reward = irw
        b.max_exp = int(np.log2(max_tile)) if max_tile else 0
        b.reset()
    assert np.array_equal(g.boards, o.boards)
    illegal_ends = 0
    for s in range(64):
        g.step(None, auto_reset=auto_reset)
        o.step(None, auto_reset=auto_reset)
        for f in ("boards", "score", "reward", "terminated", "illegal", "highest", "last_score"):
            assert np.array_equal(getattr(g, f), getattr(o, f)), (f, s)
        done = o.terminated.astype(bool)
        assert np.array_equal(g.terminal_boards[done], o.terminal_boards[done])
        illegal_ends += int((o.illegal.astype(bool) & done).sum())
    st = g.eng.episode_stats()
    had = o.ep_count > 0                                    # boards that finished at least one episode
    assert st["episodes"] == int(o.ep_count.sum()) and st["illegal_ends"] == illegal_ends
    # the exact return sum: by the device's conservation form, and -- with auto-reset -- as the sum of the final scores of
    # ALL finished episodes (SB3 Monitor's episode returns, ppo_train.py:123), which the oracle adds up episode by episode
    assert st["return_sum"] == o.return_sum
    if auto_reset:
        assert st["return_sum"] == o.finished_return_sum and st["mean_episode_score"] == o.finished_return_sum / max(1, st["episodes"])
    assert st["last_count"] == int(had.sum()) and st["last_score_sum"] == int(o.last_score[had].sum())
    assert st["last_score_max"] == (int(o.last_score[had].max()) if had.any() else 0)
    assert st["max_exp"] == int(o.boards.max())
    hist = np.bincount(o.boards.max(axis=1), minlength=32)
    assert st["highest_hist"] == hist.tolist()
    raw_last = g.eng.last_records().cpu().numpy()
    assert np.array_equal((raw_last & 0x1F)[had], o.terminal_boards[had]) and not raw_last[~had].any()


@pytest.mark.parametrize("n,dtype_name,irw,auto_reset", [
    (4096, "uint8", -2.0, True),       # whole blocks: the FULL + standard-configuration kernel
    (4096, "int64", 0.0, False),
    (1000, "int32", -1.0, True),       # ragged
    (63, "uint8", 0.0, True),
    (65536, "int64", -0.5, True),
])
def test_standard_configuration_kernel_vs_oracle(torch_cuda, n, dtype_name, irw, auto_reset):
    """step(want_info=False) -- reward + terminated only, no max_tile -- runs step_kernel's specialisation without
    the optional-output branches; the oracle decides."""
    from oracle import OracleBatch
    from oracle.cpu_ref import random_actions_np
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048
    seed, off = 77, 4242
    eng = Batched2048(n, seed=seed, board_offset=off, illegal_move_reward=irw)
    ora = OracleBatch(n, seed, off, threads=0)
    ora.illegal_move_reward = irw
    eng.reset()
    ora.reset()
    for s in range(40):
        a = random_actions_np(seed + 1, 1 + s, off, n)          # any action sequence will do
        r, t = eng.step(torch.as_tensor(a).to(getattr(torch, dtype_name)).to(eng.device), auto_reset=auto_reset,
                        want_info=False)
        ora.step(a, auto_reset=auto_reset)
        assert np.array_equal(r.cpu().numpy(), ora.reward) and np.array_equal(t.cpu().numpy(), ora.terminated), s
    assert np.array_equal(eng.get_boards().reshape(n, 16), ora.boards)
    assert np.array_equal(eng.get_scores(), ora.score) and np.array_equal(eng.get_last_scores(), ora.last_score)
    assert eng.episode_stats()["episodes"] == int(ora.ep_count.sum())


def test_action_dtypes_and_generated_actions(torch_cuda):
    """u8 / i32 / i64 action buffers and the synthetic policy give identical trajectories; the
    device-generated action tensor equals the CPU regeneration."""
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048
    from oracle.cpu_ref import random_actions_np
    n, k, seed, off = 3000, 40, 21, 1 << 20
    engs = [Batched2048(n, seed=seed, board_offset=off) for _ in range(4)]
    for e in engs:
        e.reset()
    acts = engs[0].random_actions(k)
    assert acts.shape == (k, n)
    a_np = acts.cpu().numpy()
    for j in range(0, k, 13):
        assert np.array_equal(a_np[j], random_actions_np(seed, 1 + j, off, n))
    for j in range(k):
        engs[0].step(None)
        engs[1].step(acts[j])
        engs[2].step(acts[j].to(torch.int32))
        engs[3].step(acts[j].to(torch.int64) + 4 * (j % 3))   # only the low two bits count
        ref = engs[0].get_boards()
        for e in engs[1:]:
            assert np.array_equal(e.get_boards(), ref)
            assert torch.equal(e.reward, engs[0].reward) and torch.equal(e.terminated, engs[0].terminated)


def test_rollout_paths_agree(torch_cuda):
    """k per-step launches (g2048_rollout with [k,n] buffers), the fused one-launch rollout and k
    separate step() calls end in the same state; rollout buffers hold the per-step outputs."""
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048
    n, k, seed = 20000, 50, 99
    a, b, c, d = (Batched2048(n, seed=seed) for _ in range(4))
    for e in (a, b, c, d):
        e.reset()
    acts = a.random_actions(k)
    rew = torch.zeros((k, n), dtype=torch.float32, device=a.device)
    term = torch.zeros((k, n), dtype=torch.uint8, device=a.device)
    a.rollout(acts, reward=rew, terminated=term)
    b.rollout_random(k)
    rew_f, term_f = torch.zeros_like(rew), torch.zeros_like(term)
    ill_f = torch.zeros_like(term)
    d.rollout(acts.to(torch.int64), reward=rew_f, terminated=term_f, illegal=ill_f, fused=True)  # one launch
    assert torch.equal(rew_f, rew) and torch.equal(term_f, term) and bool((ill_f <= term_f).all())
    rewards, terms = [], []
    for j in range(k):
        c.step(None)
        rewards.append(c.reward.clone())
        terms.append(c.terminated.clone())
    for e in (a, b, d):
        assert np.array_equal(e.get_boards(), c.get_boards())
        assert np.array_equal(e.get_scores(), c.get_scores())
        assert np.array_equal(e.get_last_scores(), c.get_last_scores())
        assert e.clock == c.clock == k
        assert e.episode_stats() == c.episode_stats()
    assert torch.equal(rew, torch.stack(rewards)) and torch.equal(term, torch.stack(terms))


@pytest.mark.parametrize("dtype_name", ["uint8", "int32", "int64"])
def test_fused_rollout_pipeline_every_tail_length(torch_cuda, dtype_name):
    """rollout_fused_kernel runs its action loads four steps ahead over two register sets, eight steps per loop trip:
    every rollout length from 1 to 19 (no full trip, exactly one, trips plus every tail length) and a ragged batch
    must give the per-step launches' outputs bit for bit, for every action dtype."""
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048
    n, seed = 777, 5
    dt = getattr(torch, dtype_name)
    for k in list(range(1, 20)) + [24, 33]:
        a, b = Batched2048(n, seed=seed, illegal_move_reward=-3.0), Batched2048(n, seed=seed, illegal_move_reward=-3.0)
        a.reset()
        b.reset()
        a.rollout_random(7)
        b.rollout_random(7)
        acts = a.random_actions(k).to(dt)
        outs = []
        for eng, fused in ((a, False), (b, True)):
            rew = torch.full((k, n), 7.0, dtype=torch.float32, device=eng.device)
            term = torch.full((k, n), 9, dtype=torch.uint8, device=eng.device)
            ill = torch.full((k, n), 9, dtype=torch.uint8, device=eng.device)
            hi = torch.full((k, n), 99, dtype=torch.uint8, device=eng.device)
            eng.rollout(acts, reward=rew, terminated=term, illegal=ill, highest=hi, fused=fused)
            outs.append((rew, term, ill, hi))
        for x, y in zip(*outs):
            assert torch.equal(x, y), (k, dtype_name)
        assert np.array_equal(a.get_boards(), b.get_boards()) and np.array_equal(a.get_scores(), b.get_scores()), k
        assert np.array_equal(a.get_last_scores(), b.get_last_scores()) and a.clock == b.clock, k
        assert a.episode_stats() == b.episode_stats(), k
        a.close()
        b.close()


def test_full_size_invariants(torch_cuda):
    """BASELINE's 2^20-board configuration: size-independent properties instead of a full oracle run.
    (1) sharding invariance: board i of a 2^20 batch == board i of the 2-shard run;
    (2) a 4096-board window of the big batch equals the oracle;
    (3) conservation: sum of tile values changes by exactly the spawned tiles on legal, non-reset steps."""
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048
    from oracle import OracleBatch
    n, seed, k = 1 << 20, 42, 24
    big = Batched2048(n, seed=seed)
    lo = Batched2048(n // 2, seed=seed, board_offset=0)
    hi = Batched2048(n // 2, seed=seed, board_offset=n // 2)
    win_off = n - 4096
    o = OracleBatch(4096, seed, win_off, threads=0)
    for e in (big, lo, hi, o):
        e.reset()
    for s in range(k):
        before = big.tile_values().sum(dim=(1, 2))
        for e in (big, lo, hi):
            e.step(None)
        o.step(None)
        after = big.tile_values().sum(dim=(1, 2))
        cont = (big.terminated == 0)
        delta = (after - before)[cont]
        assert bool(((delta == 2) | (delta == 4)).all()), "a legal step adds exactly one 2 or 4"
        assert torch.equal(big.boards()[: n // 2], lo.boards()) and torch.equal(big.boards()[n // 2:], hi.boards())
        assert np.array_equal(big.boards()[win_off:].cpu().numpy().reshape(-1, 16), o.boards)
        assert np.array_equal(big.reward[win_off:].cpu().numpy(), o.reward)
    assert torch.equal(big.scores()[: n // 2], lo.scores())
    # merge-score identity: score = sum((e-1) * 2^e) - 4 * (#spawned fours) >= 0 and <= that sum
    e = big.boards().to(torch.int64)
    pot = torch.where(e > 0, (e - 1) * (torch.ones_like(e) << e), torch.zeros_like(e)).sum(dim=(1, 2))
    sc = big.scores().to(torch.int64)
    assert bool((sc <= pot).all()) and bool(((pot - sc) % 4 == 0).all())


def test_masked_reset_set_state_and_views(torch_cuda):
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048
    n = 777
    e = Batched2048(n, seed=1)
    e.reset()
    for _ in range(10):
        e.step(None)
    before = e.get_boards().copy()
    scores_before = e.get_scores().copy()
    mask = torch.zeros(n, dtype=torch.uint8)
    mask[::3] = 1
    e.reset(mask=mask)
    after = e.get_boards()
    keep = mask.numpy() == 0
    assert np.array_equal(after[keep], before[keep]) and np.array_equal(e.get_scores()[keep], scores_before[keep])
    assert ((after[~keep] != 0).sum(axis=(1, 2)) == 2).all() and (e.get_scores()[~keep] == 0).all()
    # the zero-copy view aliases the engine's records: cells in the low five bits, the packed score
    # deficit only in the spare bits of bytes 8..15 (include/g2048.h)
    v = e.records()
    raw = v.cpu().numpy()
    assert v.data_ptr() == e._lib.g2048_records_ptr(e._h) and np.array_equal(raw & 0x1F, after.reshape(n, 16))
    assert not (raw[:, :8] & 0xE0).any()
    assert np.array_equal(e.boards().cpu().numpy(), after) and np.array_equal(e.scores().cpu().numpy(), e.get_scores())
    # checkpoint / resume reproduces the continuation exactly
    state = e.state_dict()
    for _ in range(5):
        e.step(None)
    ref_boards, ref_scores, ref_clock = e.get_boards(), e.get_scores(), e.clock
    f = Batched2048(n, seed=999)
    f.load_state_dict(state)
    for _ in range(5):
        f.step(None)
    assert np.array_equal(f.get_boards(), ref_boards) and np.array_equal(f.get_scores(), ref_scores)
    assert f.clock == ref_clock


def test_single_env_compat_matches_reference_trajectory(torch_cuda):
    """Game2048Env (N = 1 view) driven like the reference loop reproduces the golden trajectory of
    board 0 (seed 42), including the types the reference's tests check."""
    from gym2048_amd import Game2048Env
    d = load_golden("traj_random_seed42")
    env = Game2048Env()
    obs, info = env.reset(seed=42)
    assert obs.shape == (16, 4, 4) and info == {}
    vals = lambda e: np.where(e > 0, 1 << e.astype(np.int64), 0).reshape(4, 4)  # noqa: E731
    assert np.array_equal(env.get_board(), vals(d["initial_boards"][0]))
    for s in range(128):
        obs, reward, terminated, truncated, info = env.step(int(d["actions"][0, s]))
        assert isinstance(reward, float) and isinstance(terminated, bool) and truncated is False
        assert reward == d["reward"][0, s] and terminated == bool(d["terminated"][0, s])
        assert info["illegal_move"] == bool(d["illegal"][0, s]) and info["highest"] == 1 << int(d["highest"][0, s])
        assert np.array_equal(env.get_board(), vals(d["terminal_boards"][0, s]))
        assert obs.sum(axis=0).max() <= 1 and set(np.unique(obs)) <= {0, 1}
        if terminated:
            env.reset()
        assert np.array_equal(env.get_board(), vals(d["boards"][0, s]))
        assert env.score == d["score"][0, s]


def test_vec_env_adapter_matches_oracle_backed_adapter(torch_cuda):
    """Vec2048 over the real engine == Vec2048 over the oracle-backed fake (obs, rewards, dones, infos)."""
    from fake_engine import OracleEngine
    from gym2048_amd import Vec2048
    n, seed = 512, 17
    real = Vec2048(n, seed=seed, illegal_move_reward=-1.0)
    fake = Vec2048(n, seed=seed, illegal_move_reward=-1.0, engine=OracleEngine(n, seed))
    assert np.array_equal(real.reset(), fake.reset())
    rng = np.random.default_rng(5)
    for _ in range(40):
        a = rng.integers(0, 4, n)
        o1, r1, d1, i1 = real.step(a)
        o2, r2, d2, i2 = fake.step(a)
        assert np.array_equal(o1, o2) and np.array_equal(r1, r2) and np.array_equal(d1, d2)
        for x, y in zip(i1, i2):
            assert set(x) == set(y)
            if x:
                assert x["episode"]["r"] == y["episode"]["r"] and x["episode"]["l"] == y["episode"]["l"]
                assert x["highest"] == y["highest"] and x["illegal_move"] == y["illegal_move"]
                assert np.array_equal(x["terminal_observation"], y["terminal_observation"])
        assert np.array_equal(real.action_masks(), fake.action_masks())      # g2048_legal_actions vs the oracle's trial moves
    assert real.render().shape == (280, 280, 3)


# ---------------------------------------------------------------- numpy-compatible RNG mode

@pytest.mark.parametrize("name", ["traj_numpy_seed42", "traj_numpy_seed7_irw"])
def test_numpy_rng_mode_replays_the_reference_with_its_own_rng(torch_cuda, name):
    """rng='numpy': board i == the UNMODIFIED reference env after reset(seed = s + i), drawing from
    numpy's PCG64 exactly as gymnasium's np_random does (golden: tests/golden/traj_numpy_*.npz)."""
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048
    d = load_golden(name)
    seed, _, n, steps, _, _ = (int(x) for x in d["meta"])
    eng = Batched2048(n, seed=seed, rng="numpy", illegal_move_reward=float(d["illegal_move_reward"][0]))
    eng.reset()
    assert np.array_equal(eng.get_boards().reshape(n, 16), d["initial_boards"])
    for s in range(steps):
        eng.step(torch.as_tensor(d["actions"][:, s]))
        assert np.array_equal(eng.get_boards().reshape(n, 16), d["boards"][:, s]), s
        assert np.array_equal(eng.reward.cpu().numpy(), d["reward"][:, s])
        assert np.array_equal(eng.terminated.cpu().numpy(), d["terminated"][:, s])
        assert np.array_equal(eng.illegal.cpu().numpy(), d["illegal"][:, s])
        assert np.array_equal(eng.highest.cpu().numpy(), d["highest"][:, s])
        assert np.array_equal(eng.get_scores(), d["score"][:, s])


def test_numpy_rng_mode_vs_oracle_and_generator_state(torch_cuda):
    """4 096 boards x 48 steps against the C oracle's numpy mode, RNG states included; the final device
    RNG state loaded into numpy Generators continues like the oracle's."""
    from gym2048_amd.batched import Batched2048
    from gym2048_amd.seeding import planes_to_generators
    from oracle import OracleBatch
    n, seed = 4096, 123
    eng = Batched2048(n, seed=seed, rng="numpy")
    ob = OracleBatch(n, seed, threads=0)
    ob.seed_numpy(seed)
    assert np.array_equal(eng.get_numpy_rng().T, ob.rng)          # device seeding == numpy's SeedSequence/PCG64
    from gym2048_amd.seeding import pcg64_planes
    big = Batched2048(3, seed=2 ** 40 + 7, board_offset=100, rng="numpy")  # entropy >= 2^32: two words
    assert np.array_equal(big.get_numpy_rng(), pcg64_planes([2 ** 40 + 107, 2 ** 40 + 108, 2 ** 40 + 109]))
    eng.reset()
    ob.reset_numpy()
    for s in range(48):
        eng.step(None)
        ob.step_numpy(None)
        assert np.array_equal(eng.get_boards().reshape(n, 16), ob.boards), s
        assert np.array_equal(eng.reward.cpu().numpy(), ob.reward)
    assert np.array_equal(eng.get_numpy_rng().T, ob.rng)
    assert np.array_equal(eng.get_scores(), ob.score)
    gens = planes_to_generators(eng.get_numpy_rng()[:, :4])
    from oracle.numpy_rng import NumpyPCG64
    for g, col in zip(gens, ob.rng[:4]):
        mine = NumpyPCG64(int(col[0]) | (int(col[1]) << 64), int(col[2]) | (int(col[3]) << 64),
                          int(col[4]) >> 32, int(col[4]) & 0xFFFFFFFF)
        assert g.random() == mine.random()
    # checkpoint / resume carries the generators
    state = eng.state_dict()
    eng.step(None)
    ref = eng.get_boards()
    other = Batched2048(n, seed=1)
    other.load_state_dict(state)
    other.step(None)
    assert np.array_equal(other.get_boards(), ref) and other.rng_mode == "numpy"
    # g2048_rollout_random exists in this mode since round 6 (the fused kernel with the synthetic policy): the oracle,
    # stepped one step at a time with the same policy, ends on the same boards and generators
    ob.step_numpy(None)                                            # (the step taken above, after the checkpoint)
    eng.rollout_random(7)
    for _ in range(7):
        ob.step_numpy(None)
    assert np.array_equal(eng.get_boards().reshape(n, 16), ob.boards) and np.array_equal(eng.get_numpy_rng().T, ob.rng)


def test_single_env_numpy_mode_is_the_reference_env(torch_cuda):
    """Game2048Env(rng='numpy').reset(seed=42) gives the reference's own first board (SURVEY 8c:
    [[0,0,0,0],[0,0,2,0],[0,0,0,0],[0,2,0,0]] with numpy 2.x) and then its trajectory."""
    from gym2048_amd import Game2048Env
    d = load_golden("traj_numpy_seed42")
    env = Game2048Env(rng="numpy")
    env.reset(seed=42)
    assert env.get_board().tolist() == [[0, 0, 0, 0], [0, 0, 2, 0], [0, 0, 0, 0], [0, 2, 0, 0]]
    vals = lambda e: np.where(e > 0, 1 << e.astype(np.int64), 0).reshape(4, 4)  # noqa: E731
    for s in range(96):
        _, reward, term, _, info = env.step(int(d["actions"][0, s]))
        assert reward == d["reward"][0, s] and term == bool(d["terminated"][0, s])
        if term:
            env.reset()
        assert np.array_equal(env.get_board(), vals(d["boards"][0, s]))


def test_argument_validation_on_device(torch_cuda):
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048
    e = Batched2048(128, seed=1)
    e.reset()
    with pytest.raises(TypeError):
        e.step(torch.zeros(128, dtype=torch.float32))
    with pytest.raises(ValueError):
        e.step(torch.zeros(127, dtype=torch.int64))
    with pytest.raises(ValueError):
        e.rollout(torch.zeros((4, 128), dtype=torch.uint8), reward=torch.zeros((3, 128), device=e.device))
    with pytest.raises(ValueError):
        e.observe_onehot(out=torch.zeros((128, 16, 4, 4), dtype=torch.int32, device=e.device))
    with pytest.raises(ValueError):
        Batched2048(8, rng="mt19937")
    with pytest.raises(ValueError):
        e.load_state_dict({"blob": np.zeros(10, np.uint8)})
    # a blob written before ABI 14 ("\0G2048v3") was played under the previous spawn rule: refused, with the reason
    from gym2048_amd._lib import G2048Error
    old = e.state_dict()
    assert bytes(old["blob"][:8]) == b"\0G2048v5"
    old["blob"][7] = ord("3")
    with pytest.raises(G2048Error, match="before ABI 14"):
        e.load_state_dict(old)
    old["blob"][7] = ord("4")                                    # ABI 14's blob: the same games in another slab layout
    with pytest.raises(G2048Error, match="written by ABI 14"):
        e.load_state_dict(old)
    old["blob"][7] = ord("5")
    e.load_state_dict(old)                                       # ... and the untouched one loads
    boards = e.get_boards()
    e.step(torch.full((128,), 7, dtype=torch.int64))            # only the low two bits count: 7 == left
    f = Batched2048(128, seed=1)
    f.reset()
    f.step(torch.full((128,), 3, dtype=torch.uint8))
    assert np.array_equal(e.get_boards(), f.get_boards()) and not np.array_equal(e.get_boards(), boards)
    e.close()
    e.close()                                                    # idempotent


def test_full_size_conservation_over_a_rollout(torch_cuda):
    """2^20 boards x 200 steps through g2048_rollout: a checksum of checksums ties every output together --
    episodes == number of terminated flags; sum of all rewards == score_sum of finished episodes + the
    scores still running (illegal_move_reward = 0); illegal_ends <= episodes; and the whole run is
    reproduced bit for bit by a second engine (determinism at full size)."""
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048
    n, k = 1 << 20, 200
    a, b = Batched2048(n, seed=42), Batched2048(n, seed=42)
    for e in (a, b):
        e.reset()
    acts = a.random_actions(k)
    rew = torch.zeros((k, n), dtype=torch.float32, device=a.device)
    term = torch.zeros((k, n), dtype=torch.uint8, device=a.device)
    a.rollout(acts, reward=rew, terminated=term)
    b.rollout(acts)
    st = a.episode_stats()
    assert st["episodes"] == int(term.sum(dtype=torch.int64))
    # every reward is a merge score; each one belongs either to a finished episode or to a running one.
    # Sum of the finished ones, from the rollout buffers: episode return = sum of rewards up to its end.
    run = torch.zeros(n, dtype=torch.float64, device=a.device)
    finished = torch.zeros((), dtype=torch.float64, device=a.device)
    last = torch.zeros(n, dtype=torch.float64, device=a.device)
    for j in range(k):
        run += rew[j]
        done = term[j] != 0
        finished += run[done].sum()
        last = torch.where(done, run, last)
        run[done] = 0
    assert int(rew.sum(dtype=torch.float64)) == int(finished) + int(a.scores().sum(dtype=torch.int64))
    assert st["return_sum"] == int(finished)                           # g2048_stats.return_sum: ALL finished episodes, exact
    assert torch.equal(a.last_scores().to(torch.float64), last)       # the returns the all-gather ships
    assert st["last_score_sum"] == int(last.sum()) and st["last_score_max"] == int(last.max())
    assert 0 < st["illegal_ends"] <= st["episodes"] and st["last_score_max"] >= st["mean_last_score"] > 0
    assert sum(st["highest_hist"]) == n
    assert torch.equal(a.boards(), b.boards()) and torch.equal(a.scores(), b.scores())
    assert st == b.episode_stats() and a.clock == b.clock == k
    assert torch.equal(a.last_scores(), b.last_scores())


@pytest.mark.parametrize("max_tile,auto_reset,irw", [(64, True, -1.0), (None, False, 0.0)])
def test_numpy_rng_mode_config_variants_vs_oracle(torch_cuda, max_tile, auto_reset, irw):
    """numpy-RNG mode with max_tile / without auto-reset / with an illegal-move penalty, int32 actions,
    and a masked reset in the middle -- all against the C oracle's numpy mode."""
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048
    from oracle import OracleBatch
    n, seed = 1500, 77
    eng = Batched2048(n, seed=seed, rng="numpy", illegal_move_reward=irw, max_tile=max_tile)
    ob = OracleBatch(n, seed, threads=0)
    ob.illegal_move_reward = irw
    ob.max_exp = int(np.log2(max_tile)) if max_tile else 0
    ob.seed_numpy(seed)
    eng.reset()
    ob.reset_numpy()
    rng = np.random.default_rng(3)
    for s in range(60):
        acts = rng.integers(0, 4, n).astype(np.int32)
        eng.step(torch.as_tensor(acts), auto_reset=auto_reset)
        ob.step_numpy(acts.astype(np.uint8), auto_reset=auto_reset)
        assert np.array_equal(eng.get_boards().reshape(n, 16), ob.boards), s
        assert np.array_equal(eng.reward.cpu().numpy(), ob.reward)
        assert np.array_equal(eng.terminated.cpu().numpy(), ob.terminated)
        assert np.array_equal(eng.get_scores(), ob.score)
    assert np.array_equal(eng.get_numpy_rng().T, ob.rng)
    # masked reset consumes exactly two spawns of the selected boards' generators
    mask = (np.arange(n) % 5 == 0).astype(np.uint8)
    before = eng.get_numpy_rng().T.copy()
    eng.reset(mask=torch.as_tensor(mask))
    after = eng.get_numpy_rng().T
    assert np.array_equal(after[mask == 0], before[mask == 0]) and not np.array_equal(after[mask == 1], before[mask == 1])
    assert ((eng.get_boards()[mask == 1] != 0).sum(axis=(1, 2)) == 2).all()


def test_seeding_helpers_round_trip():
    from gym2048_amd.seeding import pcg64_planes, planes_to_generators
    planes = pcg64_planes([3, 4, 2 ** 35])
    gens = planes_to_generators(planes)
    for s, g in zip([3, 4, 2 ** 35], gens):
        want = np.random.Generator(np.random.PCG64(np.random.SeedSequence(s)))
        assert g.random() == want.random() and g.integers(0, 100) == want.integers(0, 100)


@pytest.mark.parametrize("tag,eps", [("eps0", 0.0), ("eps03", 0.3)])
def test_evaluate_model_on_the_gpu_equals_the_reference_evaluation_loop(torch_cuda, tag, eps, tmp_path):
    """gym2048_amd.evaluate.evaluate_model with the HIP engine (numpy-RNG mode, 24 episodes in lockstep) == what
    train.py:127-229 computed on the unmodified reference with the same policy (tests/golden/eval_table.npz)."""
    from test_host_logic import eval_fixture_policy
    from gym2048_amd.evaluate import evaluate_model, report_evaluation_results
    g = load_golden("eval_table")
    n = int(g["episodes"])
    res = evaluate_model(eval_fixture_policy(g["weights"]), n, eps)
    rows = res["Episodes"]
    assert [r["total_reward"] for r in rows] == g[f"{tag}_total_reward"].tolist()
    assert [r["highest"] for r in rows] == g[f"{tag}_highest"].tolist()
    assert [r["moves"] for r in rows] == g[f"{tag}_moves"].tolist()
    assert [r["illegal_moves"] for r in rows] == g[f"{tag}_illegal_moves"].tolist()
    name = report_evaluation_results(res, label=tag, path=str(tmp_path / f"scores_{tag}.csv"))
    assert open(name, "rb").read() == bytes(g[f"{tag}_csv"])
