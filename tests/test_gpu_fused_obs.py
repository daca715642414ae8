"""GPU: the observation fused into the step launch (g2048_step_io.obs) -- `return stack(self.Matrix), ...`
of Game2048Env.step (game2048_env.py:100, :17-32) without a second kernel -- bit-exact against the golden
trajectories captured from the reference, the reference's own stack() outputs, and the oracle at 2^20 boards."""
import numpy as np
import pytest

from conftest import TRAJECTORIES, load_golden

pytestmark = pytest.mark.gpu


def onehot_np(boards_exp, dtype=np.uint8):
    """stack() of exponent boards [n,16] -> [n,16,4,4] (channel c = exponent == c; exponents >= 16: all-zero)."""
    b = np.asarray(boards_exp).reshape(-1, 1, 4, 4)
    return (b == np.arange(16, dtype=np.uint8).reshape(1, 16, 1, 1)).astype(dtype)


@pytest.mark.parametrize("name", TRAJECTORIES)
def test_fused_obs_on_golden_trajectories(torch_cuda, name):
    """Every step of every golden trajectory with the uint8 observation written by the step launch: it must be
    stack() of the board the reference holds after that step (after its `if terminated: env.reset()`)."""
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048
    d = load_golden(name)
    seed, offset, n, steps, max_exp, auto_reset = (int(x) for x in d["meta"])
    eng = Batched2048(n, seed=seed, board_offset=offset, illegal_move_reward=float(d["illegal_move_reward"][0]),
                      max_tile=(1 << max_exp) if max_exp else None)
    eng.reset()
    obs = torch.zeros((n, 16, 4, 4), dtype=torch.uint8, device=eng.device)
    for s in range(steps):
        a = torch.as_tensor(np.ascontiguousarray(d["actions"][:, s], dtype=np.uint8))
        eng.step(a, auto_reset=bool(auto_reset), obs=obs)
        assert np.array_equal(eng.reward.cpu().numpy(), d["reward"][:, s]), s
        assert np.array_equal(eng.terminated.cpu().numpy(), d["terminated"][:, s]), s
        assert np.array_equal(obs.cpu().numpy(), onehot_np(d["boards"][:, s])), f"obs step {s}"
        assert np.array_equal(eng.get_boards().reshape(n, 16), d["boards"][:, s]), s


@pytest.mark.parametrize("dtype_name", ["uint8", "float16", "float32"])
def test_fused_obs_equals_reference_stack_outputs(torch_cuda, dtype_name):
    """The reference's own stack() outputs (tests/golden/stack_table.npz, incl. 2^16 / 2^17 tiles whose columns are
    all-zero): the boards are installed, a step that cannot change them is played (an illegal move keeps the
    board, game2048_env.py:91-95 -- direction chosen per board by trial moves), and the fused observation of every
    dtype must equal the table.  Boards that have no illegal direction are compared through the oracle instead."""
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048
    from oracle import OracleBatch
    dt = getattr(torch, dtype_name)
    t = load_golden("stack_table")
    n = len(t["boards"])
    eng = Batched2048(n, seed=3)
    eng.set_boards(t["boards"])
    mask = eng.legal_actions().cpu().numpy()
    illegal_dir = np.zeros(n, np.uint8)
    has_illegal = np.zeros(n, bool)
    for dirn in range(4):
        free = ((mask >> dirn) & 1) == 0
        illegal_dir[free & ~has_illegal] = dirn
        has_illegal |= free
    assert has_illegal.sum() >= 4
    obs = torch.full((n, 16, 4, 4), 7, dtype=dt, device=eng.device)
    ora = OracleBatch(n, 3)
    ora.boards[:] = t["boards"]
    eng.step(torch.as_tensor(illegal_dir), auto_reset=False, obs=obs)
    ora.step(illegal_dir, auto_reset=False)
    got = obs.cpu().numpy()
    assert np.array_equal(got[has_illegal], t["onehot"][has_illegal].astype(got.dtype))
    assert np.array_equal(got, ora.onehot().astype(got.dtype))
    assert np.array_equal(eng.get_boards().reshape(n, 16), ora.boards)


@pytest.mark.parametrize("n,dtype_name,auto_reset", [(1 << 16, "uint8", True), (65536 + 77, "float16", True),
                                                      (4099, "float32", True), (1000, "uint8", False),
                                                      (63, "float16", False), (1, "float32", True),
                                                      (257, "uint8", True)])
def test_fused_obs_ragged_sizes_and_dtypes_vs_oracle(torch_cuda, n, dtype_name, auto_reset):
    """Whole-block and ragged batches (the last wavefront writes only the boards that exist: the buffer carries a
    guard row that must stay untouched), every observation dtype, through g2048_rollout with [k,n,16,4,4] buffers."""
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048
    from oracle import OracleBatch
    dt = getattr(torch, dtype_name)
    k, seed = 12, 11
    eng, ora = Batched2048(n, seed=seed), OracleBatch(n, seed)
    eng.reset()
    ora.reset()
    acts = eng.random_actions(k)
    flat = torch.full((k * n + 1, 16, 4, 4), 9, dtype=dt, device=eng.device)      # + one guard board at the end
    obs = flat[: k * n].view(k, n, 16, 4, 4)
    reward = torch.zeros((k, n), dtype=torch.float32, device=eng.device)
    term = torch.zeros((k, n), dtype=torch.uint8, device=eng.device)
    eng.rollout(acts, reward=reward, terminated=term, auto_reset=auto_reset, obs=obs)
    got, a = obs.cpu().numpy(), acts.cpu().numpy()
    for j in range(k):
        ora.step(a[j], auto_reset=auto_reset)
        assert np.array_equal(got[j], ora.onehot().astype(got.dtype)), j
        assert np.array_equal(reward[j].cpu().numpy(), ora.reward), j
        assert np.array_equal(term[j].cpu().numpy(), ora.terminated), j
    assert bool((flat[k * n] == 9).all()), "the fused observation wrote past the last board"
    assert np.array_equal(eng.get_boards().reshape(n, 16), ora.boards)


def test_fused_obs_with_every_optional_output(torch_cuda):
    """The non-standard kernel flavour (illegal / highest / terminal_boards wanted, max_tile set) with a fused
    observation: every output against the oracle."""
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048
    from oracle import OracleBatch
    n, seed = 20000, 5
    eng, ora = Batched2048(n, seed=seed, illegal_move_reward=-2.0, max_tile=32), OracleBatch(n, seed)
    ora.illegal_move_reward, ora.max_exp = -2.0, 5
    eng.reset()
    ora.reset()
    obs = torch.zeros((n, 16, 4, 4), dtype=torch.float16, device=eng.device)
    for s in range(60):
        eng.step(None, obs=obs)
        ora.step(None)
        assert np.array_equal(obs.cpu().numpy(), ora.onehot().astype(np.float16)), s
        assert np.array_equal(eng.reward.cpu().numpy(), ora.reward), s
        assert np.array_equal(eng.illegal.cpu().numpy(), ora.illegal), s
        assert np.array_equal(eng.highest.cpu().numpy(), ora.highest), s
        done = ora.terminated.astype(bool)
        assert np.array_equal(eng.terminated.cpu().numpy(), ora.terminated), s
        assert np.array_equal(eng.terminal_boards.cpu().numpy()[done], ora.terminal_boards[done]), s


def test_fused_obs_2p20_full_batch_vs_oracle(torch_cuda):
    """BASELINE configs[2] at its real size with the observation: 2^20 boards x 40 steps, the fused uint8 one-hot
    of the WHOLE batch against the oracle's onehot() every step (and boards / rewards / flags as before)."""
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048
    from oracle import OracleBatch
    n, seed, steps = 1 << 20, 42, 40
    eng, ora = Batched2048(n, seed=seed), OracleBatch(n, seed, threads=0)
    eng.reset()
    ora.reset()
    obs = torch.zeros((n, 16, 4, 4), dtype=torch.uint8, device=eng.device)
    want_dev = torch.empty((n, 16), dtype=torch.uint8, device=eng.device)
    chan = torch.arange(16, dtype=torch.uint8, device=eng.device).view(1, 16, 1)
    for s in range(steps):
        eng.step(None, want_info=False, obs=obs)
        ora.step(None)
        assert np.array_equal(eng.reward.cpu().numpy(), ora.reward), s
        assert np.array_equal(eng.terminated.cpu().numpy(), ora.terminated), s
        if s % 8 == 7 or s == 0:      # the oracle's own onehot() (pinned to the reference's stack() outputs) ...
            assert np.array_equal(obs.cpu().numpy(), ora.onehot()), s
        else:                          # ... and in between the same comparison from the oracle's BOARDS, on the device
            want_dev.copy_(torch.from_numpy(ora.boards))
            assert bool(torch.equal(obs.view(n, 16, 16), (want_dev.view(n, 1, 16) == chan).to(torch.uint8))), s
    assert np.array_equal(eng.get_boards().reshape(n, 16), ora.boards)


def test_fused_obs_numpy_rng_mode(torch_cuda):
    """numpy-RNG mode: the observation follows the compacted resets (stand-alone kernel behind the step)."""
    torch = torch_cuda
    from gym2048_amd.batched import Batched2048
    from oracle import OracleBatch
    n, seed = 3000, 42
    eng, ora = Batched2048(n, seed=seed, rng="numpy"), OracleBatch(n, seed)
    ora.seed_numpy(seed)
    eng.reset()
    ora.reset_numpy()
    obs = torch.zeros((n, 16, 4, 4), dtype=torch.uint8, device=eng.device)
    rs = np.random.default_rng(1)
    for s in range(40):
        a = rs.integers(0, 4, n).astype(np.uint8)
        eng.step(torch.as_tensor(a), obs=obs)
        ora.step_numpy(a)
        assert np.array_equal(obs.cpu().numpy(), ora.onehot()), s


def test_fused_rollout_refuses_obs(torch_cuda):
    torch = torch_cuda
    from gym2048_amd import G2048Error
    from gym2048_amd.batched import Batched2048
    eng = Batched2048(256, seed=1)
    eng.reset()
    obs = torch.zeros((2, 256, 16, 4, 4), dtype=torch.uint8, device=eng.device)
    with pytest.raises(G2048Error, match="obs"):
        eng.rollout(eng.random_actions(2), fused=True, obs=obs)
    with pytest.raises(ValueError):
        eng.step(None, obs=obs[0].cpu())
    with pytest.raises(ValueError):
        eng.step(None, obs=obs[0].to(torch.int32))
