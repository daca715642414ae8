"""GPU: the reference-side ctypes stub printed in INTEGRATION.md really works -- the code block is
extracted from the document, pointed at the in-tree library and compared with Batched2048."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_integration_md_stub_runs_and_matches_engine(torch_cuda):
    torch = torch_cuda
    import __graft_entry__ as ge
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(import ctypes as C.*?)```", text, re.S).group(1)
    code = code.replace('C.CDLL("libg2048_hip.so")', f'C.CDLL({ge.HIP_LIB!r})')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    n, seed = 2048, 42
    stub = ns["HipBoards"](n, device=0, seed=seed)
    from gym2048_amd.batched import Batched2048
    eng = Batched2048(n, seed=seed)
    obs = stub.reset(seed)
    eng.reset(seed=seed)
    assert torch.equal(obs, eng.observe_onehot(torch.uint8))
    rng = np.random.default_rng(0)
    for _ in range(30):
        a = torch.as_tensor(rng.integers(0, 4, n)).to("cuda:0")
        obs, reward, terminated = stub.step(a)
        eng.step(a)
        assert torch.equal(obs, eng.observe_onehot(torch.uint8))
        assert torch.equal(reward, eng.reward) and torch.equal(terminated, eng.terminated)
    mean, episodes = stub.mean_episode_return()
    st = eng.episode_stats()
    assert episodes == st["episodes"] > 0 and mean == st["mean_episode_return"] == st["return_sum"] / st["episodes"]


def test_integration_md_mentions_every_public_entry_point_group():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name in ("g2048_create", "g2048_reset", "g2048_step", "g2048_onehot", "g2048_set_numpy_rng", "Vec2048",
                 "register()", "allgather_returns"):
        assert name in text, name
