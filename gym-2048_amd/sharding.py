"""Multi-GPU layout: one process per GPU, contiguous shards of the global batch, no data-path
collective.  The only exchange is the episodic-return all-gather once per rollout (RCCL over xGMI
through ``torch.distributed``; backend "nccl" is RCCL on ROCm, "gloo" in the CPU tests).

The reference has no distributed code at all; this follows BASELINE.json's
north_star: shard the batch over the GPUs of one node, all-gather only the episodic returns.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class Shard:
    rank: int
    world_size: int
    n_global: int
    offset: int   # global index of this rank's first board
    n_local: int

    @property
    def stop(self) -> int:
        return self.offset + self.n_local


def shard_range(n_global: int, rank: int, world_size: int) -> Shard:
    """Contiguous split; the first ``n_global % world_size`` ranks get one extra board."""
    if not 0 <= rank < world_size:
        raise ValueError("rank out of range")
    base, extra = divmod(n_global, world_size)
    n_local = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return Shard(rank, world_size, n_global, offset, n_local)


def weak_shard(n_per_gpu: int, rank: int, world_size: int) -> Shard:
    """Weak scaling: every rank owns ``n_per_gpu`` boards of a ``world_size * n_per_gpu`` batch."""
    return Shard(rank, world_size, n_per_gpu * world_size, rank * n_per_gpu, n_per_gpu)


def allgather_returns(local_returns: torch.Tensor, shard: Shard) -> torch.Tensor:
    """All-gather per-board episodic returns into the global order (``[n_global]`` on every rank).

    ``local_returns``: ``[n_local]`` tensor on this rank's device (e.g. ``Batched2048.last_scores()``).
    Equal shards use one ``all_gather_into_tensor`` (one RCCL all-gather over xGMI); ragged shards
    pad to the largest shard first.
    """
    if not dist.is_initialized():               # no process group: nothing to exchange
        return local_returns.clone()
    if dist.get_world_size() != shard.world_size:
        # the collective is over the WHOLE default group: a shard description for another world size would make the
        # ranks disagree about the output size (or leave the others waiting).  A one-rank shard inside a larger job
        # has nothing to exchange; anything else is a caller error.
        if shard.world_size == 1:
            return local_returns.clone()
        raise ValueError(f"shard.world_size = {shard.world_size} but the process group has {dist.get_world_size()} ranks")
    # (a ONE-rank group with a one-rank shard still runs the collective: bench.py's forced-dist mode)
    base, extra = divmod(shard.n_global, shard.world_size)
    n_max = base + (1 if extra else 0)
    send = local_returns
    if send.numel() != n_max:
        send = torch.zeros(n_max, dtype=local_returns.dtype, device=local_returns.device)
        send[: local_returns.numel()] = local_returns
    out = torch.empty(n_max * shard.world_size, dtype=send.dtype, device=send.device)
    dist.all_gather_into_tensor(out, send.contiguous())
    if extra == 0:
        return out
    parts = [out[r * n_max: r * n_max + shard_range(shard.n_global, r, shard.world_size).n_local]
             for r in range(shard.world_size)]
    return torch.cat(parts)


def allgather_stats(local_stats: torch.Tensor) -> torch.Tensor:
    """All-gather the per-rank episodic-return SUMMARY (``Batched2048.episode_stats_device()``: the C struct
    g2048_stats as ``uint8 [sizeof]``) -> ``uint8 [world, sizeof]`` on every rank.  176 bytes per rank:
    one latency-bound RCCL all-gather per rollout (SURVEY 8e's first option)."""
    if not dist.is_initialized():
        return local_stats.reshape(1, -1).clone()
    flat = local_stats.contiguous().reshape(-1)
    out = torch.empty(dist.get_world_size() * flat.numel(), dtype=flat.dtype, device=flat.device)
    dist.all_gather_into_tensor(out, flat)            # flat in, flat out: accepted by RCCL and gloo alike
    return out.reshape(dist.get_world_size(), flat.numel())


class SummaryExchange:
    """The once-per-rollout exchange through the LIBRARY's communicator instead of the framework's process group:
    ``g2048_allgather_summary`` = the returns-only summary kernel writing straight into this rank's row of the gathered
    array + ONE in-place ``ncclAllGather`` (RCCL over xGMI), both enqueued on the CURRENT stream right behind the rollout's
    last step launch -- no hop to a communication stream and back (``all_gather_into_tensor`` records an event on the
    launch stream, waits for it on ProcessGroupNCCL's own stream, and the launch stream then waits for that one).

    The communicator is created once: rank 0's ``ncclUniqueId`` reaches the others through ``torch.distributed``
    (any backend; ``broadcast_object_list``), then ``ncclCommInitRank`` on every rank.  Without a process group it is a
    one-rank communicator.  ``gather()`` returns the ``uint8 [world, sizeof(g2048_stats)]`` tensor (``merge_stats`` reads
    it after the stream has been synchronised)."""

    def __init__(self, engine):
        import ctypes as C
        from . import _lib
        self._lib = _lib.load()
        self._engine = engine
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        if world > 1 and torch.cuda.device_count() < 2:
            # ncclCommInitRank refuses two ranks on one device ("Duplicate GPU detected") -- and may leave the other ranks
            # waiting inside the rendezvous: say so before anybody enters it (one-GPU smoke tests use allgather_stats)
            raise ValueError(f"SummaryExchange needs one GPU per rank: {world} ranks, {torch.cuda.device_count()} visible device(s)")
        ident = (C.c_uint8 * _lib.COMM_ID_BYTES)()
        if rank == 0:
            _lib.check(self._lib.g2048_comm_unique_id(ident))
        if world > 1:
            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=0)
            ident = (C.c_uint8 * _lib.COMM_ID_BYTES).from_buffer_copy(box[0])
        comm = C.c_void_p()
        _lib.check(self._lib.g2048_comm_create(world, rank, ident, engine.device_index, C.byref(comm)))
        self._comm = comm
        self.world, self.rank = world, rank
        self._row = C.sizeof(_lib.Stats)
        self.out = torch.zeros((world, self._row), dtype=torch.uint8, device=engine.device)

    def gather(self) -> torch.Tensor:
        from . import _lib
        e = self._engine
        _lib.check(self._lib.g2048_allgather_summary(e._h, self._comm, self.out.data_ptr(), e._stream()))
        return self.out

    def close(self):
        if getattr(self, "_comm", None):
            self._lib.g2048_comm_destroy(self._comm)
            self._comm = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def merge_stats(rows) -> dict:
    """Global summary from the gathered per-rank structs (host side, after the stream has been synchronised)."""
    from .batched import parse_stats
    parts = [parse_stats(r) for r in rows]
    tot = dict(episodes=sum(p["episodes"] for p in parts), illegal_ends=sum(p["illegal_ends"] for p in parts),
               max_exp=max(p["max_exp"] for p in parts),
               highest_hist=[sum(col) for col in zip(*(p["highest_hist"] for p in parts))],
               return_sum=sum(p["return_sum"] for p in parts))
    # the last_* numbers exist only where the terminal records were read (parse_stats: None for a returns-only summary
    # or an engine that does not keep them): one such row and the global figure is unknown, not a smaller sum
    if all(p["last_count"] is not None for p in parts):
        tot.update(last_count=sum(p["last_count"] for p in parts), last_score_sum=sum(p["last_score_sum"] for p in parts),
                   last_score_max=max(p["last_score_max"] for p in parts))
        tot["mean_last_score"] = tot["last_score_sum"] / tot["last_count"] if tot["last_count"] else 0.0
    else:
        tot.update(last_count=None, last_score_sum=None, last_score_max=None, mean_last_score=None)
    tot["mean_episode_score"] = tot["return_sum"] / tot["episodes"] if tot["episodes"] else 0.0
    return tot


class LocalShards:
    """ONE process driving several GPUs of a node -- SURVEY 8e's single-process form of the same layout: engine r on
    ``devices[r]`` owns global boards ``[r * n_per_gpu, (r + 1) * n_per_gpu)`` (global-index spawn stream: the games
    do not depend on the split) and the once-per-rollout all-gather of episodic returns goes through the library's
    persistent communicator set (``g2048_comm_local_create``: ncclCommInitAll once; ``g2048_allgather_returns_local``:
    export kernels + one grouped ncclAllGather, enqueued).

    ONE LAUNCH THREAD PER DEVICE (SURVEY 8e "scaling risk"): a ``rollout`` of k steps is k host launches of ~3 us
    each, so a serial loop over 8 engines would start the last device's first kernel ~8 x k x 3 us after the first
    device's.  Every engine therefore has its own worker thread (a one-thread executor: calls on one engine stay in
    order); the C ABI calls are made through ctypes, which releases the GIL, so the launch trains of the devices are
    issued concurrently.  Each worker enqueues on ITS device's stream ``streams[r]`` (by default the stream that was
    current on the constructing thread), which is what ``synchronize()`` and ``allgather_returns()`` use as well.
    The one-process-per-GPU form (``weak_shard`` + ``torch.distributed``) is what ``bench.py`` uses; this class is for
    callers that do not want a launcher."""

    def __init__(self, n_per_gpu: int, devices=None, seed: int = 0, threads: bool = True, streams=None, **engine_kwargs):
        from concurrent.futures import ThreadPoolExecutor
        from . import _lib
        from .batched import Batched2048
        if devices is None:
            devices = list(range(torch.cuda.device_count()))
        self.devices = [int(d) for d in devices]
        self.n_per_gpu = int(n_per_gpu)
        self.engines = [Batched2048(self.n_per_gpu, device=d, seed=seed, board_offset=r * self.n_per_gpu, **engine_kwargs)
                        for r, d in enumerate(self.devices)]
        # torch's "current stream" is per thread: the workers must launch on the streams the caller sees
        self.streams = list(streams) if streams is not None else [torch.cuda.current_stream(e.device) for e in self.engines]
        if len(self.streams) != len(self.engines) or any(s.device != e.device for s, e in zip(self.streams, self.engines)):
            raise ValueError("streams: one torch.cuda.Stream per engine, on that engine's device")
        self._pools = [ThreadPoolExecutor(1, thread_name_prefix=f"g2048-dev{d}") for d in self.devices] if threads else None
        self.issue_windows = [(0.0, 0.0)] * len(self.engines)    # host (start, end) of each engine's part of the last call
        self._lib = _lib.load()
        self._comm = None                              # the communicator set is built by the first all-gather

    @property
    def n_global(self) -> int:
        return self.n_per_gpu * len(self.engines)

    def _each(self, fn):
        """``fn(r, engine)`` for every engine, each on its own device thread and stream; returns when all calls have
        RETURNED (= everything is enqueued; the devices keep running).  The first exception is re-raised."""
        import time

        def on_device(r):
            t0 = time.perf_counter()
            with torch.cuda.stream(self.streams[r]):
                out = fn(r, self.engines[r])
            self.issue_windows[r] = (t0, time.perf_counter())
            return out
        if self._pools is None:                       # threads=False: the serial form (tests compare the two)
            return [on_device(r) for r in range(len(self.engines))]
        futures = [pool.submit(on_device, r) for r, pool in enumerate(self._pools)]
        return [f.result() for f in futures]

    def reset(self, seed=None):
        self._each(lambda r, e: e.reset(seed=seed))

    def rollout_random(self, k_steps: int):
        self._each(lambda r, e: e.rollout_random(k_steps))

    def rollout(self, actions, **buffers):
        """``actions`` (and every optional ``[k, n]`` buffer): one tensor per device, on that device."""
        self._each(lambda r, e: e.rollout(actions[r], **{name: buf[r] for name, buf in buffers.items() if buf is not None}))

    def prepare_rollout(self, actions, **buffers):
        """Validated per-device launch descriptors (``Batched2048.prepare_rollout``); ``run_plans`` issues them."""
        return [e.prepare_rollout(actions[r], **{name: buf[r] for name, buf in buffers.items() if buf is not None})
                for r, e in enumerate(self.engines)]

    def run_plans(self, plans):
        self._each(lambda r, e: plans[r].run())

    def allgather_returns(self):
        """Every device's copy of all ``n_global`` last episodic returns (int32), in global board order.  Enqueued on
        the devices' streams (``self.streams``) from the caller's thread -- one grouped RCCL call must come from one
        thread; ``synchronize()`` (or using the tensors on those streams) waits for it."""
        import ctypes as C
        from . import _lib
        g = len(self.engines)
        if self._comm is None:                         # ncclCommInitAll: hundreds of milliseconds, ONCE
            comm = C.c_void_p()
            _lib.check(self._lib.g2048_comm_local_create((C.c_int * g)(*self.devices), g, C.byref(comm)))
            self._comm = comm
        outs = []
        for e, s in zip(self.engines, self.streams):
            with torch.cuda.stream(s):                 # allocated ON the stream that writes it: torch's caching allocator
                outs.append(torch.empty(self.n_global, dtype=torch.int32, device=e.device))  # orders reuse per stream
        eng = (C.c_void_p * g)(*[e._h for e in self.engines])
        ptrs = (C.c_void_p * g)(*[o.data_ptr() for o in outs])
        streams = (C.c_void_p * g)(*[s.cuda_stream for s in self.streams])
        _lib.check(self._lib.g2048_allgather_returns_local(self._comm, eng, ptrs, streams))
        return outs

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def close(self):
        if getattr(self, "_pools", None):
            for pool in self._pools:
                pool.shutdown(wait=True)
            self._pools = None
        if getattr(self, "_comm", None):
            self._lib.g2048_comm_local_destroy(self._comm)
            self._comm = None
        for e in self.engines:
            e.close()
