"""The data format next to the step path: the reference's 35-column transition CSV and augment().
CPU: format byte-identical to what the reference's training_data class writes; GPU: the device
augmentation kernel and rollout recording."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden
from gym2048_amd.transitions import Transitions, csv_header, unstack


def fixture_transitions():
    d = load_golden("training_data_fixture")
    return d, Transitions(d["x"], d["action"], d["reward"], d["next_x"], d["done"])


@pytest.mark.parametrize("name,add_returns", [("transitions_ref.csv", False), ("transitions_ref_returns.csv", True)])
def test_export_csv_is_byte_identical_to_the_reference(name, add_returns):
    _, t = fixture_transitions()
    want = open(os.path.join(GOLDEN, name)).read()
    assert t.to_csv_text(add_returns) == want
    assert want.splitlines()[0].split(",") == csv_header(add_returns) and len(csv_header()) == 35


def test_import_csv_round_trip_and_returns(tmp_path):
    d, t = fixture_transitions()
    path = tmp_path / "t.csv"
    t.export_csv(str(path))
    back = Transitions.import_csv(str(path))
    for f in ("x", "action", "reward", "next_x", "done"):
        assert np.array_equal(getattr(back, f), getattr(t, f)), f
    assert np.allclose(t.discounted_return(), d["returns"], rtol=0, atol=0)
    ref_file = Transitions.import_csv(os.path.join(GOLDEN, "transitions_ref.csv"))
    assert np.array_equal(ref_file.x, t.x) and np.array_equal(ref_file.done, t.done)


def test_unstack_inverts_stack():
    from gym2048_amd import stack
    s = load_golden("stack_table")
    for e in s["boards"][:16]:
        vals = np.where(e > 0, 1 << e.astype(np.int64), 0).reshape(4, 4)
        capped = np.where(vals > 32768, 0, vals)        # tiles above 2^15 have no layer
        assert np.array_equal(unstack(stack(vals)), capped)


@pytest.mark.gpu
def test_device_augment_matches_reference_augment(torch_cuda):
    d, t = fixture_transitions()
    a = t.augment()
    assert a.size() == 8 * t.size()
    assert np.array_equal(a.x, d["aug_x"]) and np.array_equal(a.action, d["aug_action"])
    assert np.array_equal(a.next_x, d["aug_next_x"]) and np.array_equal(a.reward, d["aug_reward"])
    assert np.array_equal(a.done, d["aug_done"])


@pytest.mark.gpu
def test_record_rollout_transitions_are_consistent_with_the_oracle(torch_cuda):
    import ctypes as C

    import oracle
    from gym2048_amd.batched import Batched2048
    lib = oracle.load()
    eng = Batched2048(64, seed=4)
    eng.reset()
    t = Transitions.record(eng, None, n_steps=40)
    assert t.size() == 64 * 40
    I64x16 = C.c_int64 * 16
    n_done = 0
    for i in range(t.size()):
        M, sc = I64x16(*t.x[i].reshape(16)), C.c_int64()
        legal = lib.g2048o_move(M, int(t.action[i, 0]), 0, C.byref(sc))
        moved = np.array(list(M)).reshape(4, 4)
        if legal:
            assert t.reward[i, 0] == sc.value
            diff = np.flatnonzero((moved != t.next_x[i]).reshape(16))
            assert len(diff) == 1 and moved.reshape(16)[diff[0]] == 0 and t.next_x[i].reshape(16)[diff[0]] in (2, 4)
        else:
            assert t.done[i, 0] and np.array_equal(t.next_x[i], t.x[i])
        n_done += int(t.done[i, 0])
        if (i + 1) % 40 and not t.done[i, 0]:
            assert np.array_equal(t.next_x[i], t.x[i + 1])    # env-major: consecutive rows chain
    assert n_done > 50
    text = t.to_csv_text()
    assert text.count("\n") == t.size() + 1


@pytest.mark.gpu
def test_recompute_rewards_matches_the_reference_add_rewards_script(torch_cuda):
    """Transitions.recompute_rewards == add_rewards_to_training_data.get_reward_for_state_action run by the
    reference env (tests/golden/rewards_table.npz)."""
    t = load_golden("rewards_table")
    vals = np.where(t["boards"] > 0, 1 << t["boards"].astype(np.int64), 0)
    tr = Transitions(vals, t["actions"], np.zeros(len(vals)), vals, np.zeros(len(vals), bool))
    assert np.array_equal(tr.recompute_rewards().reward.reshape(-1), t["rewards_default"])
    assert np.array_equal(tr.recompute_rewards(illegal_move_reward=-1.0).reward.reshape(-1), t["rewards_minus1"])
