import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


_golden_cache = {}


def load_golden(name):
    """Golden fixture as a dict of arrays (eagerly decompressed once per session)."""
    if name not in _golden_cache:
        with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
            _golden_cache[name] = {k: z[k] for k in z.files}
    return _golden_cache[name]


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle
    oracle.build()
    return oracle.load()


@pytest.fixture(scope="session")
def host_check():
    """The device header g2048_device.h compiled for the host (tests/host_check)."""
    import ctypes as C
    import __graft_entry__ as ge
    lib = C.CDLL(ge.build_host_check())
    lib.hostcheck_shift.restype = C.c_uint32
    lib.hostcheck_move.restype = C.c_int
    lib.hostcheck_highest.restype = C.c_uint32
    lib.hostcheck_count_empty.restype = C.c_uint32
    lib.hostcheck_has_equal_neighbours.restype = C.c_int
    return lib


@pytest.fixture(scope="session")
def torch_cuda():
    """torch with a ROCm device and every native piece built.  On a box without a GPU the `gpu` tests
    are skipped (a plain `pytest` run stays green); on the GPU box they run and the product raises
    G2048Error if its HIP library is missing -- there is no fallback to skip into."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU (torch.cuda.is_available() is False)")
    import __graft_entry__ as ge
    ge.build()
    return torch


TRAJECTORIES = ["traj_random_seed42", "traj_random_offset", "traj_greedy_irw", "traj_greedy_max256",
                "traj_noautoreset", "traj_corner_deep"]


def replay_trajectory(make_batch, d, check_terminal=True):
    """Drive a batch object (OracleBatch-shaped) through a golden trajectory and compare everything the
    reference produced.  ``make_batch(n, seed, board_offset)`` -> object with reset()/step()/fields."""
    seed, offset, n, steps, max_exp, auto_reset = (int(x) for x in d["meta"])
    b = make_batch(n, seed, offset)
    b.illegal_move_reward = float(d["illegal_move_reward"][0])
    b.max_exp = max_exp
    b.reset()
    assert np.array_equal(b.boards, d["initial_boards"])
    for s in range(steps):
        b.step(d["actions"][:, s], auto_reset=bool(auto_reset))
        assert np.array_equal(b.reward, d["reward"][:, s]), f"reward step {s}"
        assert np.array_equal(b.terminated, d["terminated"][:, s]), f"terminated step {s}"
        assert np.array_equal(b.illegal, d["illegal"][:, s]), f"illegal step {s}"
        assert np.array_equal(b.highest, d["highest"][:, s]), f"highest step {s}"
        assert np.array_equal(b.boards, d["boards"][:, s]), f"boards step {s}"
        assert np.array_equal(b.score, d["score"][:, s]), f"score step {s}"
        if check_terminal:
            done = d["terminated"][:, s].astype(bool)
            assert np.array_equal(b.terminal_boards[done], d["terminal_boards"][done, s]), f"terminal step {s}"
            assert np.array_equal(b.last_score[done], d["terminal_score"][done, s]), f"terminal score step {s}"
    return b
