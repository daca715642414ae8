"""Worker of tests/test_gpu_configs.py::test_two_rank_hip_shards_allgather: one rank of a 2-process gloo
group; every rank drives its own HIP engine shard on cuda:0 and the episodic returns are all-gathered.
Usage (env: RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT): dist_hip_worker.py n seed k out.npz"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n, seed, k, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gym2048_amd.batched import Batched2048
    from gym2048_amd.sharding import shard_range, allgather_returns, allgather_stats, merge_stats
    shard = shard_range(n, rank, world)
    eng = Batched2048(shard.n_local, device=0, seed=seed, board_offset=shard.offset)
    eng.reset()
    eng.rollout(k)                                     # k launches of the synthetic policy on this shard
    returns = allgather_returns(eng.last_scores().cpu(), shard)        # gloo: host tensors
    boards = [torch.empty((shard_range(n, r, world).n_local, 16), dtype=torch.uint8) for r in range(world)]
    dist.all_gather(boards, torch.from_numpy(eng.get_boards().reshape(-1, 16)))
    summ = merge_stats(allgather_stats(eng.episode_stats_device().cpu()))    # per-rank summary structs
    if rank == 0:
        np.savez(out, returns=returns.numpy(), boards=torch.cat(boards).numpy(), episodes=summ["episodes"],
                 last_sum=summ["last_score_sum"], last_max=summ["last_score_max"], hist=np.array(summ["highest_hist"]),
                 return_sum=summ["return_sum"])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
