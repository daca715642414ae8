"""CPU, world_size = 2 over gloo: the multi-GPU layout (one process per shard, global-index RNG, one
all-gather of episodic returns per rollout) with the oracle standing in for the per-rank engine.
On the GPU box the same sharding helpers run over RCCL (bench.py --gpus N)."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_global, steps, seed, out_dir):
    sys.path.insert(0, ROOT)
    from gym2048_amd._lib import Stats
    from gym2048_amd.sharding import allgather_returns, allgather_stats, merge_stats, shard_range
    from oracle import OracleBatch
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        shard = shard_range(n_global, rank, world)
        ob = OracleBatch(shard.n_local, seed, board_offset=shard.offset)
        ob.reset()
        for _ in range(steps):
            ob.step(None)
        returns = allgather_returns(torch.from_numpy(ob.last_score.copy()), shard)
        boards = allgather_returns(torch.from_numpy(ob.boards.astype(np.int32).sum(axis=1).astype(np.int32)), shard)
        # the per-rank summary struct the engine's stats kernel would leave on the device (g2048_stats)
        had = ob.ep_count > 0
        st = Stats(episodes=int(ob.ep_count.sum()), illegal_ends=0, last_count=int(had.sum()),
                   last_score_sum=int(ob.last_score[had].sum()), last_score_max=int(ob.last_score.max()),
                   max_exp=int(ob.boards.max()), return_sum=ob.return_sum)
        for k, v in enumerate(np.bincount(ob.boards.max(axis=1), minlength=32)):
            st.highest_hist[k] = int(v)
        rows = allgather_stats(torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8))
        assert rows.shape == (world, C.sizeof(Stats))
        # a shard description that does not belong to this process group: a one-rank shard has nothing to exchange
        # (no collective is entered, so no rank is left waiting); any other mismatch is refused before the collective
        from gym2048_amd.sharding import Shard
        mine = torch.arange(5, dtype=torch.int32) + rank
        alone = allgather_returns(mine, Shard(0, 1, 5, 0, 5))
        assert torch.equal(alone, mine) and alone.data_ptr() != mine.data_ptr()
        with pytest.raises(ValueError, match="process group has"):
            allgather_returns(mine, shard_range(5 * (world + 1), 0, world + 1))
        summ = merge_stats(rows)
        if rank == 0:
            np.savez(os.path.join(out_dir, "gathered.npz"), returns=returns.numpy(), boards=boards.numpy(),
                     episodes=summ["episodes"], max_score=summ["last_score_max"], last_count=summ["last_count"],
                     last_sum=summ["last_score_sum"], hist=np.array(summ["highest_hist"]), return_sum=summ["return_sum"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_global,world", [(4096, 2), (1001, 2), (1003, 4)])   # equal and ragged shards
def test_sharded_ranks_equal_single_process(tmp_path, n_global, world):
    from oracle import OracleBatch
    steps, seed = 40, 42
    mp.spawn(_worker, args=(world, _free_port(), n_global, steps, seed, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "gathered.npz")
    ref = OracleBatch(n_global, seed)
    ref.reset()
    for _ in range(steps):
        ref.step(None)
    assert np.array_equal(got["returns"], ref.last_score)                      # global order, bit-exact
    assert np.array_equal(got["boards"], ref.boards.astype(np.int32).sum(axis=1))
    assert int(got["episodes"]) == int(ref.ep_count.sum()) and int(got["max_score"]) == int(ref.last_score.max())
    assert int(got["last_count"]) == int((ref.ep_count > 0).sum()) and int(got["last_sum"]) == int(ref.last_score.sum())
    assert np.array_equal(got["hist"], np.bincount(ref.boards.max(axis=1), minlength=32))
    assert int(got["return_sum"]) == ref.return_sum == ref.finished_return_sum > 0      # the exact all-episode return sum


def test_allgather_single_process_is_identity():
    from gym2048_amd.sharding import allgather_returns, shard_range
    x = torch.arange(10, dtype=torch.int32)
    y = allgather_returns(x, shard_range(10, 0, 1))
    assert torch.equal(x, y) and y.data_ptr() != x.data_ptr()
