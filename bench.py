#!/usr/bin/env python
"""bench.py -- env-steps/s of the batched 2048 step path on N MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--boards B]

A "step" is one pass of the hot path over the whole batch: step_kernel advances every board by one action
(move + score + spawn + done + auto-reset), reading the action from, and writing reward/terminated to, [K][B]
rollout buffers that are resident in HBM before the timed region starts.  The launch form is chosen by K ALONE, the same
for every N: below 200 steps the pass is ONE launch on one stream (--chains 1); from 200 steps it is TWO launches per rank,
one per half of the batch, run as two chains on two streams from two host threads (g2048_set_chains(2): bit-identical to
one chain; the head of one half-batch kernel overlaps the tail of the other's); --chains overrides, `config.chains` says
what the timed rollout did.  Workload = BASELINE configs[2]: B = 2^20 boards per GPU, synthetic uniform random policy
(actions pre-generated on the device by g2048_fill_random_actions), seed 42.

N > 1, one rank per GPU: either under a launcher (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`:
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment) or as plain `python bench.py --gpus N ...`, which
starts the N ranks itself (self_launch: same environment, rank 0's JSON line relayed as the LAST line of stdout, worst
exit code returned; fewer than N visible GPUs -> one clear line on stderr, exit code 2).  Rank r owns global boards
[r*B, (r+1)*B) -- no data-path collective; one RCCL all-gather of the episodic returns at the end of the
rollout, inside the timed region and enqueued ON THE LAUNCH STREAM right behind the K step launches: ONE summary
launch writing into this rank's row + ONE in-place ncclAllGather through the library's own communicator
(g2048_allgather_summary; G2048_BENCH_COLLECTIVE=torch selects torch.distributed's all_gather_into_tensor instead:
11.5-12.3 against 22-27 us with a one-rank group, profiles/r06_j_forced_dist_runs.txt; `config.collective_path`).  The
all-gather is also the closing barrier of the timed region: no rank's copy completes before every rank has
contributed, so each rank only synchronises its device afterwards and the elapsed times are MAX-reduced
over ranks.  `timing` in the JSON splits the region: launch_train_us (K launches, HIP events),
collective_us (statistics kernel + all-gather, HIP events), host_tail_us (what the wall clock saw on top).
scaling = "weak".  G2048_BENCH_FORCE_DIST=1 runs exactly this N > 1 code path with a one-rank RCCL process
group on a one-GPU box (init_process_group("nccl"), the device-tensor all-gather, the NCCL barrier).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     -- algorithmic bytes (38 B/env-step) / HIP-event time per launch vs the 8 TB/s HBM peak; `resident` says
                  what the 2^20-board figure is a fraction of (the 38 MiB a launch touches sit in the Infinity Cache),
                  `streaming` = the same kernel at 2^24 boards (HBM), `small_batch` = BASELINE configs[1] (65 536 boards);
                  `traffic` = HBM bytes per launch from the committed PMC profile, printed only while the
                  profile was taken from the kernel sources that are being run (hash of csrc/), else null
  host         -- gc / scheduling priority of the timed regions (--nice N to change it), CPU model, physical / visible cores
  cpu_baseline -- the C oracle (oracle/, "port") timed on this box's host cores on a bounded sample, OpenMP team calibrated
                  (threads_tried); reference_estimate =
                  the Python port's rate here / the port-to-reference ratio measured in the build container
  timing       -- the split of the region, and k_region_repeats_us: five further regions with the same brackets
                  (`value` is the FIRST region)
  extras       -- the other launch form (two_chains_k200 / single_chain), BASELINE configs[1] (batch_65536), a 2^24-board
                  run that really streams HBM (best of 4), the step that writes its one-hot observation (one launch), the
                  policy-in-the-loop figure (BASELINE configs[4]), fused rollout kernels, numpy-RNG mode, the single-env host
                  path, Python port
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# dmabuf IPC: what RCCL needs across processes on this driver.  Read when the HIP runtime starts, so it is set before
# torch is imported -- also for ranks started by somebody else's launcher.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ALGO_BYTES_PER_STEP = 38          # board in 16 + action 1 + board out 16 + reward 4 + terminated 1
HBM_PEAK_GBS = 8000.0             # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SEED = 42
REF_PORT_RATIO = 3.53            # oracle.cpu_ref.RefEnv steps/s over the reference's Game2048Env.step steps/s
AGE_STEPS = 64                    # fused steps that bring a freshly reset batch to the steady-state mix of episode ages


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--boards", type=int, default=1 << 20, help="boards per GPU")
    ap.add_argument("--no-extras", action="store_true", help="skip fused / streaming / cpu legs")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--device-warmup", type=float, default=0.25,
                    help="seconds of GPU work on a SCRATCH engine before anything is measured (clocks at their "
                         "sustained level, as in a long-running job); 0 disables")
    ap.add_argument("--chains", type=int, choices=[1, 2], default=None,
                    help="launch chains per rollout (g2048_set_chains): 2 = the batch is cut in two and the halves run as two "
                         "chains of launches on two streams from two host threads (bit-identical results; the head of one "
                         "half-batch kernel overlaps the tail of the other's), 1 = one launch per step on one stream.  Default: 2, "
                         "except in a multi-rank run of fewer than 200 steps (see main())")
    ap.add_argument("--no-two-chain-extra", action="store_true",
                    help="skip extras.two_chains_k200 (under a profiler that serialises kernels across queues -- rocprofv3 --pmc -- "
                         "the two chains' ticket kernels wait for each other until their bound runs out)")
    ap.add_argument("--nice", type=int, default=None, metavar="N",
                    help="run the timed regions at scheduling priority N (e.g. -10; needs the privilege).  Default: the "
                         "priority the process was started with -- the headline number must not depend on who runs it")
    ap.add_argument("--gather", choices=["summary", "full"], default="summary",
                    help="N > 1: what the once-per-rollout all-gather ships -- the per-rank return summary "
                         "(g2048_stats) or every board's last episodic return (int32[B])")
    return ap.parse_args()


def host_cpu_info() -> dict:
    """What the CPU legs ran on: the CPUs this process may use, how many PHYSICAL cores they are (distinct
    thread_siblings_list entries in sysfs; SMT siblings share one), and the model string."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(os.cpu_count() or 1))
    cores = set()
    for c in cpus:
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as f:
                cores.add(f.read().strip())
        except OSError:
            cores.add(str(c))
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"visible_cpus": len(cpus), "physical_cores": max(1, len(cores)), "model": model}


def cpu_baseline(target_seconds: float) -> dict:
    """The C oracle (oracle/libg2048_oracle.so, OpenMP over boards) on a bounded sample of the same
    workload: B_cpu boards, synthetic random policy, auto-reset, seed 42.

    The OpenMP team is CALIBRATED, not assumed: every candidate size up to the number of PHYSICAL cores (an oversubscribed
    team on SMT siblings or under a cgroup CPU quota is slower, and round 5 picked one from a two-step sample: 3.6e7 on 64
    threads in one run, 1.9e7 on 128 in the next) runs for >= 0.3 s after a warm-up step, and the fastest is used for the
    sample.  `cores` = the threads used; `threads_tried` = every calibration result; `per_core_value` = value / cores."""
    from oracle import OracleBatch
    info = host_cpu_info()
    cap = min(info["visible_cpus"], info["physical_cores"])
    n = 1 << 18
    tried = {}
    for threads in sorted({1, 8, 16, 32, 64, 96, 128, cap}):
        if threads > cap:
            continue
        ob = OracleBatch(n, SEED, threads=threads)
        ob.reset()
        ob.step(None)                                      # team start-up, first touch
        steps, t0 = 0, time.perf_counter()
        while steps < 2 or time.perf_counter() - t0 < 0.3:
            ob.step(None)
            steps += 1
        tried[threads] = steps * n / (time.perf_counter() - t0)
    best_threads = max(tried, key=tried.get)
    ob = OracleBatch(n, SEED, threads=best_threads)
    ob.reset()
    steps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < target_seconds:      # bounded by wall time, not by a step count
        ob.step(None)
        steps += 1
    dt = time.perf_counter() - t0
    one = OracleBatch(1 << 16, SEED, threads=1)
    one.reset()
    one.step(None)
    t1 = time.perf_counter()
    for _ in range(16):
        one.step(None)
    dt1 = time.perf_counter() - t1
    value = n * steps / dt
    return {"value": value, "unit": "env-steps/s", "cores": best_threads, "kind": "port",
            "sample": f"C oracle (oracle/g2048_oracle.c, OpenMP x{best_threads}; {info['physical_cores']} physical cores / "
                      f"{info['visible_cpus']} visible CPUs), {n} boards x {steps} steps, random policy, auto-reset, seed {SEED}; {dt:.1f} s",
            "per_core_value": value / best_threads,
            "threads_tried": {str(k): v for k, v in sorted(tried.items())},
            "cpu_model": info["model"], "physical_cores": info["physical_cores"], "visible_cpus": info["visible_cpus"],
            "single_core_value": (1 << 16) * 16 / dt1}


def python_port_rate() -> float:
    """oracle.cpu_ref.RefEnv, BASELINE configs[0] shape: one env, seed 42, random actions, 1 core."""
    from oracle.cpu_ref import RefEnv, random_action
    n = 20000
    acts = [random_action(SEED, t + 1, 0) for t in range(n)]
    env = RefEnv(SEED, 0)
    env.reset()
    t0 = time.perf_counter()
    for a in acts:
        if env.step(a)[1]:
            env.reset()
    return n / (time.perf_counter() - t0)


CSRC_FILES = ("gym-2048_amd/csrc/g2048_device.h", "gym-2048_amd/csrc/g2048_kernels.hip",
              "gym-2048_amd/csrc/g2048_kernels.h", "gym-2048_amd/csrc/g2048_api.hip", "gym-2048_amd/csrc/g2048_pcg64.h",
              "gym-2048_amd/csrc/g2048_side_launcher.h",
              "include/g2048.h")


def csrc_hash() -> str:
    """sha256[:16] over the kernel / ABI sources: stamps a PMC profile with the code it was taken from."""
    h = hashlib.sha256()
    for rel in CSRC_FILES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def load_traffic():
    """The committed PMC profile (profiles/traffic_latest.json: HBM bytes per launch of step_kernel at
    2^20 boards, the tag of the profile and the hash of the sources it was taken from), or None."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def traffic_for_current_sources(boards: int):
    """(bytes_per_launch | None, provenance): the number is reported only for the profiled batch size and
    only while the kernel sources still hash to what was profiled -- a stale profile prints null."""
    t = load_traffic()
    if not t:
        return None, {"profile": None}
    prov = {"profile": t.get("profile"), "profiled_csrc": t.get("csrc_sha16"), "current_csrc": csrc_hash()}
    if t.get("csrc_sha16") == prov["current_csrc"] and t.get("kernel_trace_avg_us"):
        # the rocprofv3 --kernel-trace --stats average of the committed profile of THESE sources, taken from the ONE-CHAIN
        # form (`bench.py --chains 1`: one whole-batch launch per step): the cross-check of extras.single_chain.launch_us
        prov["kernel_trace_avg_us"] = t["kernel_trace_avg_us"]
        prov["kernel_trace_dispatches"] = t.get("kernel_trace_dispatches")
    if boards != (1 << 20) or t.get("csrc_sha16") != prov["current_csrc"]:
        return None, prov
    return t.get("bytes_per_launch"), prov


def free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n: int, cmd, env=None, grace: float = 15.0, timeout: float = None) -> int:
    """Run ``cmd`` as ``n`` ranks of one node (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in
    the environment, as torch.distributed.run would set them), relay what they print and return the worst exit code.

    Output contract: every line rank 0 writes to stdout is passed through in order, EXCEPT that its last JSON line
    (the bench line) is held back and printed as the very LAST line of this process's stdout; the other ranks'
    stdout goes to stderr.  If a rank dies, the others get ``grace`` seconds to finish (they are usually stuck in a
    collective waiting for it) and are then terminated -- by their own PIDs."""
    import subprocess
    import threading
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs across processes on this driver
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0")
        procs.append(subprocess.Popen(list(cmd), env=e, stdout=subprocess.PIPE if r == 0 else sys.stderr, text=r == 0))
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout), daemon=True)
    reader.start()
    t0 = time.monotonic()
    first_failure = None
    while True:
        states = [p.poll() for p in procs]               # poll EVERY rank (any() would stop at the first live one)
        if all(c is not None for c in states):
            break
        time.sleep(0.05)
        now = time.monotonic()
        if first_failure is None and any(c not in (None, 0) for c in states):
            first_failure = now
        overdue = timeout is not None and now - t0 > timeout
        if overdue or (first_failure is not None and now - first_failure > grace):
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(5)
                except subprocess.TimeoutExpired:
                    p.kill()
            if overdue:
                print(f"bench.py: the {n}-rank run exceeded {timeout:.0f} s and was stopped", file=sys.stderr)
    reader.join(10)
    codes = [p.wait() for p in procs]
    last_json = None
    for idx in range(len(lines) - 1, -1, -1):
        if lines[idx].lstrip().startswith("{"):
            last_json = lines.pop(idx)
            break
    for ln in lines:
        sys.stdout.write(ln)
    if last_json is not None:
        sys.stdout.write(last_json if last_json.endswith("\n") else last_json + "\n")
    sys.stdout.flush()
    worst = 0
    for r, c in enumerate(codes):
        if c != 0:
            print(f"bench.py: rank {r} exited with {c}", file=sys.stderr)
            worst = max(worst, c if c > 0 else 128 - c)      # a rank killed by signal s counts as 128 + s
    if worst == 0 and last_json is None:
        print("bench.py: rank 0 printed no JSON line", file=sys.stderr)
        worst = 1
    return worst


def self_launch(n: int, argv) -> int:
    """``python bench.py --gpus N`` (N > 1) started WITHOUT torch.distributed.run: start the N ranks ourselves, one per
    GPU, with exactly the environment the launcher would have given them (the torchrun form keeps working: it sets
    WORLD_SIZE and this function is never reached).  Fails with one clear line when the box has fewer than N GPUs."""
    if os.environ.get("G2048_BENCH_SAME_DEVICE") != "1":          # (that switch puts every rank on cuda:0: gloo smoke test)
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            print(f"bench.py: --gpus {n} needs {n} visible GPUs and this box has {have}; nothing was run", file=sys.stderr)
            return 2
    limit = float(os.environ.get("G2048_BENCH_LAUNCH_TIMEOUT", "3000"))
    return launch_ranks(n, [sys.executable, os.path.abspath(__file__)] + list(argv), timeout=limit)


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: no launcher around us, so this process becomes the launcher
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    if world != args.gpus:
        args.gpus = world                                # under a launcher the launcher's world size is the truth
    if os.environ.get("G2048_BENCH_SAME_DEVICE") != "1" and torch.cuda.is_available() and torch.cuda.device_count() <= local_rank:
        sys.exit(f"bench.py: rank {rank} wants cuda:{local_rank} but this box has {torch.cuda.device_count()} visible GPU(s)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a ROCm GPU; the product has no CPU path")
    # G2048_BENCH_BACKEND=gloo + G2048_BENCH_SAME_DEVICE=1: smoke-test the N > 1 code path on a
    # one-GPU box (all ranks share cuda:0, collectives over gloo through host copies).
    # G2048_BENCH_FORCE_DIST=1: the N > 1 code path (process group, device all-gather, barriers) with ONE rank.
    backend = os.environ.get("G2048_BENCH_BACKEND", "nccl")
    force_dist = os.environ.get("G2048_BENCH_FORCE_DIST") == "1"
    if os.environ.get("G2048_BENCH_SAME_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist_on = world > 1 or force_dist
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build_hip()
    if dist_on:
        dist.barrier()
    from gym2048_amd.batched import Batched2048
    from gym2048_amd.sharding import weak_shard, allgather_returns, allgather_stats, merge_stats, SummaryExchange

    B, K, W = args.boards, args.steps, args.warmup
    shard = weak_shard(B, rank, world)
    # Episode bookkeeping of the benchmarked engine: what the once-per-rollout exchange needs.  --gather summary ships
    # episodes / illegal ends / the exact return sum, none of which needs the per-board terminal records, so they are
    # off (g2048_set_last_records: one sparse 16-byte store per finished episode less; extras.with_last_records has the
    # same launch train with them on); --gather full ships every board's last return and keeps them.
    if args.chains is None:
        # The launch form is decided by K ALONE, the same for every N, so that the points of a scaling curve are the same
        # thing measured at different N: one chain (one whole-batch launch per step) below 200 steps, two chains from 200.
        # Below 200 steps two chains buy nothing measurable (the driver's K = 20: 9.34 vs 9.36 us per step, BENCH_r04) and
        # behind an RCCL communicator they delay the once-per-rollout exchange (profiles/r04_ab_forced_dist_chains.txt);
        # at K = 1 000 they are 13 % faster per step.
        args.chains = 1 if K < 200 else 2
    if args.chains == 2:
        os.environ.setdefault("G2048_SIDE_SPIN_US", "2000")   # opt into the long spin window of the side launch thread (default 200 us)
    keep_last = args.gather == "full"
    eng = Batched2048(B, device=local_rank, seed=SEED, board_offset=shard.offset, last_records=keep_last, chains=args.chains)
    eng.reset()
    # State preparation (not timed, not counted as warm-up steps): SURVEY 8d defines the workload as ">= 1 000 steps
    # after >= 50 warm-up steps".  Right after a reset every board is two tiles old and a random move is illegal far
    # more often than in the steady state (more episode ends per launch, 10-30 % slower launches for ~30 steps), so
    # the boards are first advanced AGE_STEPS steps with the fused kernel: the batch then has the steady-state mix of
    # episode ages whatever --warmup says.
    eng.rollout_random(AGE_STEPS)

    def barrier():
        if dist_on:
            dist.barrier()
            dist.barrier()      # twice: ranks ENTER the second one within microseconds of each other, so they also leave
                                # it together -- the exit skew of a barrier would otherwise be counted by whoever left first
        torch.cuda.synchronize()

    host_gather = backend != "nccl" and dist_on        # gloo smoke test: collectives on host copies
    import ctypes
    from gym2048_amd._lib import Stats
    stats_bytes = ctypes.sizeof(Stats)                                   # sizeof(g2048_stats)
    stats_buf = torch.empty(stats_bytes, dtype=torch.uint8, device=dev)
    returns_buf = torch.empty(B, dtype=torch.int32, device=dev) if args.gather == "full" else None

    # How the summaries travel (config.collective_path): "direct" = the library's own communicator,
    # g2048_allgather_summary -- ONE summary launch writing into this rank's row + ONE in-place ncclAllGather, both on the
    # launch stream; "torch" = g2048_returns_summary_async + torch.distributed.all_gather_into_tensor (ProcessGroupNCCL
    # hops to its own stream and back).  A/B on one GPU under G2048_BENCH_FORCE_DIST: profiles/r06_*_forced_dist_ab.txt.
    collective_path = os.environ.get("G2048_BENCH_COLLECTIVE", "direct")
    if collective_path not in ("direct", "torch"):
        sys.exit("G2048_BENCH_COLLECTIVE must be 'direct' or 'torch'")
    if not dist_on or backend != "nccl" or args.gather != "summary":
        collective_path = "torch"                        # gloo smoke tests (host copies) and --gather full
    exchange, exchange_error = None, None
    if dist_on and collective_path == "direct":
        # the library's communicator is a SECOND RCCL communicator next to the process group's.  Should any rank fail to build
        # it, every rank falls back to the process-group all-gather (the ranks agree through a MIN all-reduce) and the line
        # says so -- a slower exchange is better than no scaling point
        try:
            exchange = SummaryExchange(eng)
        except Exception as exc:  # pragma: no cover - needs a multi-GPU failure
            exchange_error = f"{type(exc).__name__}: {exc}"
        ok = torch.tensor([1 if exchange is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if exchange is not None:
                exchange.close()
                exchange = None
            collective_path = "torch"
            exchange_error = exchange_error or "another rank could not build the library's communicator"

    def gather_returns():
        """The path's only exchange, once per rollout: every rank reduces the episodic returns of its shard on
        the device (ONE launch, no host sync: episodes, illegal ends and the EXACT sum of the final scores of all
        finished episodes, g2048_stats.return_sum) and the per-rank summaries are all-gathered (RCCL over xGMI,
        latency-bound); --gather full ships every board's last return instead (4 MiB per rank at 2^20).
        Everything is enqueued on the current stream into buffers allocated beforehand."""
        if args.gather == "full":
            local = eng.last_scores(out=returns_buf)     # int32[B] returns, from the terminal records (one kernel)
            return allgather_returns(local.cpu() if host_gather else local, shard)
        if exchange is not None:
            return exchange.gather()
        local = eng.episode_stats_device(out=stats_buf, returns_only=True)
        return allgather_stats(local.cpu() if host_gather else local)

    # ---- everything the timed region and its warm-up steps need is allocated, generated and TOUCHED first (a fresh
    #      box's first touch of a page must not land in the timed region; allocations and frees idle or synchronise the
    #      device for milliseconds), so that NOTHING but launches stands between the device warm-up and the timed region
    actions = eng.random_actions(K, t_first=eng.clock + 1 + W)   # [K][B] u8, resident in HBM
    reward = torch.zeros((K, B), dtype=torch.float32, device=dev)
    terminated = torch.zeros((K, B), dtype=torch.uint8, device=dev)
    if dist_on:                                          # warm the collective once (communicator setup)
        gather_returns()
    plan = eng.prepare_rollout(actions, reward=reward, terminated=terminated)   # argument checks: not timed
    ev0, ev1, ev2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    close_by_poll = os.environ.get("G2048_BENCH_CLOSE", "poll") != "sync"
    if close_by_poll:
        eng.stream_wait(eng.stream_signal())             # first use allocates the completion word: not in the timed region
    wplans = []
    if W > 0:
        wa = eng.random_actions(W)
        wr = torch.zeros((min(W, 8), B), dtype=torch.float32, device=dev)
        wt = torch.zeros((min(W, 8), B), dtype=torch.uint8, device=dev)
        for j0 in range(0, W, 8):                        # same kernel, same outputs as the timed steps
            kk = min(8, W - j0)
            wplans.append(eng.prepare_rollout(wa[j0:j0 + kk], reward=wr[:kk], terminated=wt[:kk]))

    # ---- device warm-up on a scratch engine (every rank): not steps of the benchmarked engine, whose own warm-up is
    #      exactly the W steps below.  SUSTAINED 128-launch trains of the same kernel in the same configuration: shader
    #      and memory clocks at the level of the long-running job the metric describes.  It runs LAST -- the device
    #      loses that state within a few hundred microseconds of idling (tools/ubench/gap_probe.hip: a 20-launch train
    #      takes 199.8 us right after a synchronize, 207.7 us after 0.5 ms of idle, 210.8 us after 2 ms) -- and the
    #      scratch engine is only freed after the timed region (hipFree synchronises the device).
    # Host hygiene for a 200 us region that the HOST feeds (20 launches of ~3 us each, then a poll): no cyclic garbage
    # collection inside it (what timeit does) -- collected NOW, before the device warm-up, because a full collection takes
    # milliseconds and nothing but launches may stand between the warm-up and the timed region.  The scheduling priority is
    # left alone unless --nice asks (one of eleven driver-style runs of round 5 had every region stretched to 230-470 us by
    # a host that issued a launch every 15 us instead of every 3, profiles/r05_h_bench_k20_outlier.json: a busy host shows
    # in `timing`, it is not papered over).  Both facts are in the line's top-level `host` object.
    import gc
    gc.collect()
    gc.disable()
    renice_error = None
    if args.nice is not None:
        try:
            os.setpriority(os.PRIO_PROCESS, 0, args.nice)
        except (OSError, AttributeError) as exc:
            renice_error = str(exc)
    try:
        host_priority = os.getpriority(os.PRIO_PROCESS, 0)
    except (OSError, AttributeError):
        host_priority = None
    scratch = None
    if args.device_warmup > 0:
        scratch = Batched2048(B, device=local_rank, seed=SEED + 1, last_records=keep_last, chains=args.chains)   # same configuration
        scratch.reset()
        kw = 128                                         # launches per warm-up train (~1.2 ms)
        sa = scratch.random_actions(kw)
        sr = torch.zeros((kw, B), dtype=torch.float32, device=dev)
        st_ = torch.zeros((kw, B), dtype=torch.uint8, device=dev)
        wplan = scratch.prepare_rollout(sa, reward=sr, terminated=st_)
        wshort = scratch.prepare_rollout(sa[:16], reward=sr[:16], terminated=st_[:16])
        scratch.rollout_random(128)                      # shader clocks
        if dist_on:
            dist.barrier()                               # the ranks START the warm-up together, so that they also end it
        t_end = time.perf_counter() + args.device_warmup  # together: a rank that waited for the others at the opening barrier
        while True:                                      # for a millisecond would start the timed region cooled down
            left = t_end - time.perf_counter()
            if left <= 0:
                break
            (wplan if left > 4e-3 else wshort).run()     # memory clocks: per-step launches that stream; short trains at the
            torch.cuda.synchronize()                     # end, so that every rank stops within ~0.2 ms of the deadline
    def timed_region():
        """ONE timed region with the contract's brackets: barrier + synchronize | exactly K steps [+ the exchange] | the
        host learns that the device is done.  Returns (wall seconds, launch-train ms, collective ms, gathered rows, cost
        of the synchronize() after the poll in us)."""
        barrier()                                            # opening bracket: barrier + synchronize
        ev0.record()                                         # on the (idle) launch stream: start of the launch train
        t0 = time.perf_counter()
        plan.run()                                           # g2048_rollout: EXACTLY K step launches
        ev1.record()
        # the path's only exchange: once per rollout, N > 1 only.  Enqueued behind the last step launch; at N > 1 it is
        # also the closing barrier (an all-gather completes on no rank before every rank has contributed)
        rows = gather_returns() if dist_on else None
        if dist_on:
            ev2.record()                                     # (N = 1 has no exchange to bracket: one marker packet less behind the train)
        # closing bracket: [collective +] the host learns that the device is done.  By default through the library's
        # completion word (g2048_stream_signal / g2048_stream_wait: a one-wave kernel behind everything above publishes a
        # ticket to pinned host memory with a system-scope release, the host polls it) -- the same guarantee as a stream
        # synchronisation, a few us sooner (tools/ubench/tail_probe.hip); the contract's synchronize() follows and finds
        # nothing left to wait for (its cost is reported, not timed).  G2048_BENCH_CLOSE=sync times synchronize() itself.
        if close_by_poll:
            eng.stream_wait(eng.stream_signal())
            wall = time.perf_counter() - t0
            torch.cuda.synchronize()
            after_us = (time.perf_counter() - t0 - wall) * 1e6
        else:
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            after_us = 0.0
        return wall, ev0.elapsed_time(ev1), (ev1.elapsed_time(ev2) if dist_on else 0.0), rows, after_us

    for wp in wplans:                                    # the W untimed warm-up steps of the benchmarked engine
        wp.run()
    elapsed, kernel_region_ms, collective_ms, gathered, sync_after_us = timed_region()   # THE timed region: `value`
    if gathered is not None:
        gathered = gathered.clone()          # (the direct exchange gathers into ONE buffer, which the repeat regions below overwrite)
    timed_chains = eng.chains_used
    # sanity inside the bench: the rollout really happened (episodes finished, rewards written) -- the engine's books as
    # they stand after THE timed region (what the gathered summaries describe)
    stats = eng.episode_stats()
    # five further regions of the same K steps with the same brackets (not `value`: how much one 200 us sample moves)
    repeats = []
    for _ in range(5 if K <= 2000 else 0):
        r_wall, r_train, r_coll, _, _ = timed_region()
        repeats.append((r_wall, r_train, r_coll))
    gc.enable()
    if scratch is not None:
        scratch.close()
        del scratch, sa, sr, st_, wplan, wshort

    tmax = torch.tensor([elapsed, kernel_region_ms, collective_ms] + [x for r in repeats for x in r], dtype=torch.float64,
                        device=dev if backend == "nccl" else "cpu")
    if dist_on:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed, kernel_region_ms, collective_ms = float(tmax[0]), float(tmax[1]), float(tmax[2])
    repeats = [tuple(float(x) for x in tmax[3 + 3 * i: 6 + 3 * i]) for i in range(len(repeats))]
    gathered_rows = None
    if gathered is not None:
        assert gathered.numel() == (B * world if args.gather == "full" else stats_bytes * world)
        gathered_rows = int(gathered.shape[0]) if args.gather == "summary" else int(gathered.numel() // B)

    total_steps = K * B * world
    value = total_steps / elapsed
    launch_us = kernel_region_ms * 1e3 / K
    achieved = ALGO_BYTES_PER_STEP * B / (launch_us * 1e-6) / 1e9
    traffic, traffic_prov = traffic_for_current_sources(B)
    working_set_mib = (B * 16 * 2 + B * 6) / 2**20

    eff_chains = timed_chains    # what g2048_rollout did in the timed region (a two-chain engine splits only rollouts that pay)
    wall_us_per_step = elapsed * 1e6 / K
    out = {
        "metric": ("env-steps/sec at batch=2^20 per MI355X" if B == (1 << 20) else f"env-steps/sec at batch={B} per MI355X"), "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": elapsed * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"batch={B} envs per GPU, random-policy rollout, int8 boards (BASELINE configs[2])",
                   "boards_per_gpu": B, "global_boards": B * world, "seed": SEED,
                   "state": f"reset + {AGE_STEPS} fused steps (steady-state episode ages) + the W warm-up steps",
                   "device_warmup_s": args.device_warmup,
                   "path": (("two step_kernel launches per env-step (half batches, two chains on two streams, g2048_set_chains(2)), "
                             if eff_chains == 2 else "one step_kernel launch per env-step (g2048_rollout), ") +
                            "actions/reward/terminated in [K][B] HBM rollout buffers, auto-reset fused"),
                   "chains": eff_chains, "chains_requested": eng.chains,
                   "chains_rule": "by K alone, for every N: 1 below 200 steps, 2 from 200 (DESIGN.md 5.1a); --chains overrides",
                   "episode_bookkeeping": "per-wavefront counters + exact return sum; per-board terminal records " + ("on" if keep_last else "off"),
                   # what the process group really was: a reader of a scaling curve can see that N ranks met
                   "backend": (backend if dist_on else None),
                   "pg_world_size": (dist.get_world_size() if dist_on else 1),
                   "gathered_rows": gathered_rows,
                   "collective_path": ((("g2048_allgather_summary: one summary launch + one in-place ncclAllGather on the launch stream"
                                         if collective_path == "direct" else
                                         "g2048_returns_summary_async + torch.distributed.all_gather_into_tensor") if args.gather == "summary"
                                        else "g2048_get_last_scores + torch.distributed.all_gather_into_tensor") if dist_on else None),
                   "collective_path_fallback": exchange_error,
                   "collective": ((f"one all-gather per rollout of the per-rank return summaries (g2048_stats, {stats_bytes} B each)"
                                   if args.gather == "summary" else "one all-gather per rollout of the per-board episodic returns (int32[B] each)")
                                  if dist_on else "none")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     # the same fraction from the one WALL-CLOCK number of the line (ms_per_step: host tail included)
                     "frac_wall": ALGO_BYTES_PER_STEP * B / (wall_us_per_step * 1e-6) / 1e9 / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_provenance": traffic_prov,
                     "kernel": "g2048::step_kernel<1, true, true, false>", "algorithmic_bytes_per_env_step": ALGO_BYTES_PER_STEP,
                     "launch_us": launch_us, "launches_per_step": eff_chains,
                     "basis": "achieved = 38 B x boards / (HIP-event time of the K-step region / K); frac_wall uses ms_per_step",
                     "notes": "DESIGN.md 5.1: cache-resident at 2^20 (extras.streaming_2p24 streams HBM); issue-bound before HBM-bound"},
        "timing": {"launch_train_us": kernel_region_ms * 1e3, "collective_us": collective_ms * 1e3,
                   "host_tail_us": elapsed * 1e6 - (kernel_region_ms + collective_ms) * 1e3,
                   "closing": "poll" if close_by_poll else "synchronize",
                   "synchronize_after_poll_us": sync_after_us,
                   # five further K-step regions with the same brackets, right behind the timed one (max over ranks):
                   # wall us per region, and their launch trains -- `value` stays the FIRST region
                   "k_region_repeats_us": [r[0] * 1e6 for r in repeats],
                   "k_region_repeats_launch_train_us": [r[1] * 1e3 for r in repeats],
                   "k_region_repeats_collective_us": [r[2] * 1e3 for r in repeats],
                   "k_region_first_us": elapsed * 1e6},
        # host conditions of the timed regions (top level: `value` is a ~200 us region fed by one host thread)
        "host": {"gc": "disabled inside the timed regions (collected before the device warm-up)", "nice": host_priority,
                 "nice_requested": args.nice, "renice_error": renice_error, **host_cpu_info()},
        "episodes_finished": int(stats["episodes"]), "return_sum": int(stats["return_sum"]),
        "mean_episode_score": stats["mean_episode_score"],        # exact: over ALL finished episodes of this rank's shard
    }
    if keep_last:
        out["mean_last_episode_score"] = stats["mean_last_score"]   # over each board's most recent finished episode
    if force_dist and world == 1:
        out["config"]["forced_dist"] = f"one-rank {backend} process group: the N > 1 code path on one GPU"
        out["config"]["collective"] = "one-rank all-gather of the episodic-return summary (forced)" 
    if gathered is not None and args.gather == "summary":
        g = merge_stats(gathered)
        out["global_returns"] = {"episodes": g["episodes"], "illegal_ends": g["illegal_ends"], "return_sum": g["return_sum"],
                                 "mean_episode_score": g["mean_episode_score"]}

    if rank == 0 and world == 1 and not args.no_extras and not force_dist:
        extras = {}
        # (a) fused K-step rollout kernel: boards stay in registers; NOT HBM-bound, 38 B model n/a
        eng.rollout_random(8)
        kf, runs = 256, []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            eng.rollout_random(kf)
            e1.record()
            torch.cuda.synchronize()
            runs.append(e0.elapsed_time(e1))
        extras["fused_rollout_steps_per_s"] = kf * B / (min(runs) * 1e-3)
        # (a2) per-step I/O as in the timed region (action[j][i] in, reward[j][i] / terminated[j][i] out) but as ONE
        #      fused launch of 256 steps with the boards in registers (g2048_rollout_fused; own [256][B] buffers)
        try:
            kf2 = 256
            fa = eng.random_actions(kf2)
            fr = torch.zeros((kf2, B), dtype=torch.float32, device=dev)
            ft = torch.zeros((kf2, B), dtype=torch.uint8, device=dev)
            fplan = eng.prepare_rollout(fa, reward=fr, terminated=ft, fused=True)
            fplan.run()
            torch.cuda.synchronize()
            runs = []
            for _ in range(3):                           # best of 3, each on fresh actions of the engine's next 256 steps
                fa = eng.random_actions(kf2, out=fa)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                fplan.run()
                e1.record()
                torch.cuda.synchronize()
                runs.append(e0.elapsed_time(e1))
            extras["fused_rollout_with_io_steps_per_s"] = kf2 * B / (min(runs) * 1e-3)
            extras["fused_rollout_with_io_ms_runs"] = runs
            del fa, fr, ft, fplan
        except Exception as exc:  # pragma: no cover
            extras["fused_rollout_with_io_steps_per_s"] = f"error: {exc}"
        # (a2a) the same engine run as ONE chain (one whole-batch launch per step): the kernel-level figure that a
        #       rocprofv3 kernel trace of `bench.py --chains 1` shows (profiles/*_kernel_stats.csv)
        if eng.chains == 2:
            try:
                eng.set_chains(1)
                kk = min(K, 200)
                splan = eng.prepare_rollout(actions[:kk], reward=reward[:kk], terminated=terminated[:kk])
                splan.run()
                best = None
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    splan.run()
                    e1.record()
                    torch.cuda.synchronize()
                    us = e0.elapsed_time(e1) * 1e3 / kk
                    best = us if best is None else min(best, us)
                extras["single_chain"] = {"launch_us": best, "steps_per_s": B / (best * 1e-6), "launches": kk,
                                          "frac_of_hbm_peak": ALGO_BYTES_PER_STEP * B / (best * 1e-6) / 1e9 / HBM_PEAK_GBS}
                del splan
            except Exception as exc:  # pragma: no cover
                extras["single_chain"] = {"error": str(exc)}
            finally:
                eng.set_chains(2)
        # (a2a') ... and when the timed engine ran as ONE chain (K < 200): the same engine as TWO chains over a 200-step
        #        rollout (own buffers), the form `python bench.py` (K = 1 000) times -- best of 3
        if eng.chains == 1 and not args.no_two_chain_extra:
            tc = None
            try:
                os.environ.setdefault("G2048_SIDE_SPIN_US", "2000")
                # (its own engine: a ticket wait that runs out -- under a kernel-serialising profiler -- makes an engine refuse
                #  every later call, and the extras below still need `eng`)
                tc = Batched2048(B, device=local_rank, seed=SEED, last_records=keep_last, chains=2)
                tc.reset()
                tc.rollout_random(AGE_STEPS)
                kk = 200
                a2 = tc.random_actions(kk)
                r2 = torch.zeros((kk, B), dtype=torch.float32, device=dev)
                t2 = torch.zeros((kk, B), dtype=torch.uint8, device=dev)
                tplan = tc.prepare_rollout(a2, reward=r2, terminated=t2)
                tplan.run()
                best = None
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    tplan.run()
                    e1.record()
                    torch.cuda.synchronize()
                    us = e0.elapsed_time(e1) * 1e3 / kk
                    best = us if best is None else min(best, us)
                extras["two_chains_k200"] = {"us_per_step": best, "steps_per_s": B / (best * 1e-6), "steps": kk,
                                             "chains_used": tc.chains_used,
                                             "frac_of_hbm_peak": ALGO_BYTES_PER_STEP * B / (best * 1e-6) / 1e9 / HBM_PEAK_GBS}
                del a2, r2, t2, tplan
            except Exception as exc:  # pragma: no cover
                extras["two_chains_k200"] = {"error": str(exc)}
            finally:
                if tc is not None:
                    tc.close()
        # (a2b) the timed launch train on an engine that KEEPS the per-board terminal records (the library's default):
        #       what g2048_get_last_scores / --gather full cost the step
        try:
            other = Batched2048(B, device=local_rank, seed=SEED, last_records=not keep_last, chains=args.chains)
            other.reset()
            other.rollout_random(AGE_STEPS)
            kk = min(K, 100)
            oplan = other.prepare_rollout(actions[:kk], reward=reward[:kk], terminated=terminated[:kk])
            oplan.run()
            best = None
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                oplan.run()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / kk
                best = us if best is None else min(best, us)
            extras["with_last_records" if not keep_last else "without_last_records"] = {
                "launch_us": best, "steps_per_s": B / (best * 1e-6), "launches": kk,
                "frac_of_hbm_peak": ALGO_BYTES_PER_STEP * B / (best * 1e-6) / 1e9 / HBM_PEAK_GBS}
            other.close()
            del other, oplan
        except Exception as exc:  # pragma: no cover
            extras["with_last_records"] = {"error": str(exc)}
        # (a3) the env-step INCLUDING the observation the reference's step() returns (stack(), game2048_env.py:100):
        #      ONE launch per step -- step_kernel<.., HAS_OBS> writes the [B,16,4,4] one-hot of the record it leaves
        #      behind -- through g2048_rollout over [Ko,B,16,4,4] observation buffers, best of 3 x Ko launches.  uint8
        #      (+256 B per env-step, 294 B in all), and the float16 / float32 forms the consumer asks for
        #      (ppo_train.py:62 `observations.float()`: +512 B = 550 B, +1 024 B = 1 062 B per env-step)
        for name, odt, per_board in (("u8", torch.uint8, 256), ("f16", torch.float16, 512), ("f32", torch.float32, 1024)):
            try:
                ko = min(K, 20 if odt == torch.uint8 else 8)
                obs = torch.zeros((ko, B, 16, 4, 4), dtype=odt, device=dev)
                oplan = eng.prepare_rollout(actions[:ko], reward=reward[:ko], terminated=terminated[:ko], obs=obs)
                oplan.run()
                best = None
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    oplan.run()
                    e1.record()
                    torch.cuda.synchronize()
                    us = e0.elapsed_time(e1) * 1e3 / ko
                    best = us if best is None else min(best, us)
                obs_bytes = ALGO_BYTES_PER_STEP + per_board
                extras["step_with_obs_" + name] = {"us_per_step": best, "steps_per_s": B / (best * 1e-6), "launches": ko,
                                                   "launches_per_step": 1, "algorithmic_bytes_per_env_step": obs_bytes,
                                                   "achieved_GBs": obs_bytes * B / (best * 1e-6) / 1e9,
                                                   "frac_of_hbm_peak": obs_bytes * B / (best * 1e-6) / 1e9 / HBM_PEAK_GBS}
                del obs, oplan
            except Exception as exc:  # pragma: no cover
                extras["step_with_obs_" + name] = {"error": str(exc)}
        # (a4) BASELINE configs[1]: 65 536 boards on one GPU, the same kernel, one chain; 200-step rollouts over [200][65536]
        #      buffers, best of 3 (SURVEY 8d "Config 2").  256 workgroups = one per CU: the launch is one latency chain, and
        #      one host thread issues a launch every 3-4.6 us -- this size is launch-bound, not HBM-bound -- so the rollout is
        #      replayed from a cached hipGraph of its launch train
        try:
            ns, ks = 1 << 16, 200
            small = Batched2048(ns, device=local_rank, seed=SEED, last_records=keep_last, chains=1)
            small.reset()
            small.rollout_random(AGE_STEPS)
            sa_ = small.random_actions(ks)
            sr_ = torch.zeros((ks, ns), dtype=torch.float32, device=dev)
            st2 = torch.zeros((ks, ns), dtype=torch.uint8, device=dev)
            splan = small.prepare_rollout(sa_, reward=sr_, terminated=st2).prepare_graph()   # <= 2^17 boards: a cached hipGraph
            splan.run()
            runs = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                splan.run()
                e1.record()
                torch.cuda.synchronize()
                runs.append(e0.elapsed_time(e1) * 1e3 / ks)
            us = min(runs)
            fplan = small.prepare_rollout(sa_, reward=sr_, terminated=st2, fused=True)
            fplan.run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            fplan.run()
            e1.record()
            torch.cuda.synchronize()
            extras["batch_65536"] = {"boards": ns, "launches": ks, "launch_us": us, "launch_us_runs": runs, "steps_per_s": ns / (us * 1e-6),
                                     "form": ("the launch train replayed from the engine's cached hipGraph (g2048_rollout_prepare)"
                                              if small.graph_replays else "stream launches"), "graph_replays": small.graph_replays,
                                     "frac_of_hbm_peak": ALGO_BYTES_PER_STEP * ns / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                     "fused_rollout_with_io_steps_per_s": ks * ns / (e0.elapsed_time(e1) * 1e-3)}
            small.close()
            del sa_, sr_, st2, splan, fplan
        except Exception as exc:  # pragma: no cover
            extras["batch_65536"] = {"error": str(exc)}
        # (b) a batch that does not fit L2 + Infinity Cache: 2^24 boards (256 MiB of records).  Every buffer
        #     is written once before timing (first touch), 40 ms of untimed warm-up rollouts, best of 4 timed ones.
        del reward, terminated, actions
        try:
            nb, kb = 1 << 24, 24
            big = Batched2048(nb, device=local_rank, seed=SEED, last_records=keep_last, chains=args.chains)   # as the headline engine
            big.reset()
            big.rollout_random(AGE_STEPS)
            ab = big.random_actions(kb)
            rb = torch.zeros((kb, nb), dtype=torch.float32, device=dev)
            tb = torch.zeros((kb, nb), dtype=torch.uint8, device=dev)
            # the memory system needs tens of milliseconds of sustained streaming to reach its steady state after
            # an idle or compute-only phase (tools/stream_probe.py: 145 -> 119 us per launch over the first ~30 ms),
            # so the same rollout runs untimed until 40 ms have been spent, then 4 timed repeats
            t_w = time.perf_counter()
            while time.perf_counter() - t_w < 0.04:
                big.rollout(ab, reward=rb, terminated=tb)
                torch.cuda.synchronize()
            runs = []
            for _ in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                big.rollout(ab, reward=rb, terminated=tb)
                e1.record()
                torch.cuda.synchronize()
                runs.append(e0.elapsed_time(e1) * 1e3 / kb)
            us = min(runs)
            extras["streaming_2p24"] = {"boards": nb, "steps": kb, "launch_us": us, "launch_us_runs": runs,
                                        "steps_per_s": nb / (us * 1e-6),
                                        "achieved_GBs": ALGO_BYTES_PER_STEP * nb / (us * 1e-6) / 1e9,
                                        "frac_of_hbm_peak": ALGO_BYTES_PER_STEP * nb / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}
            big.close()
            del ab, rb, tb
        except Exception as exc:  # pragma: no cover - memory pressure on a shared box
            extras["streaming_2p24"] = {"error": str(exc)}
        # (b1) BASELINE configs[4]: the env driven by a ppo_train.py-shaped policy on the same GPU (bench_policy.py)
        try:
            import bench_policy
            extras["policy_loop"] = bench_policy.run(boards=B, steps=10, warmup=2)
        except Exception as exc:  # pragma: no cover
            extras["policy_loop"] = {"error": str(exc)}
        # (b2) numpy-compatible RNG mode (the reference's own PCG64 per board, seeded on the device; board i == the unmodified
        #      reference env after reset(seed = 42 + i)): one launch per step (what a policy in the loop uses) and the fused
        #      k-step form over the same [k][B] actions (record and generator in registers, lanes drifting in time)
        try:
            nn, kn = 1 << 20, 32
            npe = Batched2048(nn, device=local_rank, seed=SEED, rng="numpy")
            npe.reset()
            an = npe.random_actions(kn)
            rn = torch.zeros((kn, nn), dtype=torch.float32, device=dev)
            tn = torch.zeros((kn, nn), dtype=torch.uint8, device=dev)
            res = {}
            for form, fused in (("per_step", False), ("fused", True)):
                nplan = npe.prepare_rollout(an, reward=rn, terminated=tn, fused=fused)
                nplan.run()
                torch.cuda.synchronize()
                best = None
                for _ in range(2):
                    t1 = time.perf_counter()
                    nplan.run()
                    torch.cuda.synchronize()
                    dtn = time.perf_counter() - t1
                    best = dtn if best is None else min(best, dtn)
                res[form] = {"steps_per_s": kn * nn / best, "us_per_step": best / kn * 1e6, "launches_per_step": 0 if fused else 1}
            res["per_step"]["algorithmic_bytes_per_env_step"] = 102      # record 16 + 16, generator 40 in + 24 out, action 1, reward 4, terminated 1
            res["per_step"]["frac_of_hbm_peak"] = 102 * nn / (res["per_step"]["us_per_step"] * 1e-6) / 1e9 / HBM_PEAK_GBS
            extras["numpy_rng_mode"] = res
            extras["numpy_rng_mode_steps_per_s"] = res["per_step"]["steps_per_s"]
            npe.close()
            del an, rn, tn, nplan
        except Exception as exc:  # pragma: no cover
            extras["numpy_rng_mode_steps_per_s"] = f"error: {exc}"
        # (b3) the drop-in single env (host arrays, one library call per step through the engine's pinned host block):
        #      Game2048Env.step + reset on termination, random actions -- what train.py / gather_training_data.py would see
        try:
            import numpy as _np
            from gym2048_amd import Game2048Env
            env = Game2048Env(device=local_rank)
            env.reset(seed=SEED)
            acts = _np.random.default_rng(0).integers(0, 4, 4000)
            for a in acts[:500]:
                if env.step(int(a))[2]:
                    env.reset()
            t1 = time.perf_counter()
            for a in acts:
                if env.step(int(a))[2]:
                    env.reset()
            extras["single_env_host_steps_per_s"] = len(acts) / (time.perf_counter() - t1)
        except Exception as exc:  # pragma: no cover
            extras["single_env_host_steps_per_s"] = f"error: {exc}"
        out["extras"] = extras
        # What the headline fraction is a fraction OF: at 2^20 boards the 38 MiB a launch touches stay in the 256 MiB
        # Infinity Cache, so `frac` is a cache-resident figure quoted against the HBM peak; the HBM figure proper is the
        # 2^24-board run (256 MiB of records: every launch streams), and BASELINE configs[1] (65 536 boards) is a latency
        # chain of 256 workgroups.  All three from HIP events in this process; kernel traces per size: profiles/r06_*_2p*.csv
        out["roofline"]["resident"] = "infinity-cache"
        st24 = extras.get("streaming_2p24")
        if isinstance(st24, dict) and "launch_us" in st24:
            out["roofline"]["streaming"] = {"boards": st24["boards"], "launch_us": st24["launch_us"], "achieved": st24["achieved_GBs"],
                                            "frac": st24["frac_of_hbm_peak"], "resident": "hbm", "launches": st24["steps"]}
        b16 = extras.get("batch_65536")
        if isinstance(b16, dict) and "launch_us" in b16:
            out["roofline"]["small_batch"] = {"boards": b16["boards"], "launch_us": b16["launch_us"],
                                              "achieved": ALGO_BYTES_PER_STEP * b16["boards"] / (b16["launch_us"] * 1e-6) / 1e9,
                                              "frac": b16["frac_of_hbm_peak"], "resident": "L2", "form": b16["form"]}
        sc = extras.get("single_chain")
        if isinstance(sc, dict) and "launch_us" in sc:
            # the KERNEL's own figure, next to the per-step one above: one whole-batch launch per step, so the HIP-event
            # time per launch is a kernel duration -- the number a kernel trace of `bench.py --chains 1` averages
            out["roofline"]["one_launch_per_step"] = {
                "launch_us": sc["launch_us"], "achieved": ALGO_BYTES_PER_STEP * B / (sc["launch_us"] * 1e-6) / 1e9,
                "frac": sc["frac_of_hbm_peak"], "launches": sc["launches"],
                "kernel_trace_avg_us": (out["roofline"].get("traffic_provenance") or {}).get("kernel_trace_avg_us")}
        elif eff_chains == 1:
            out["roofline"]["one_launch_per_step"] = {
                "launch_us": launch_us, "achieved": achieved, "frac": achieved / HBM_PEAK_GBS, "launches": K,
                "kernel_trace_avg_us": (out["roofline"].get("traffic_provenance") or {}).get("kernel_trace_avg_us")}
        # (c) CPU legs (rank 0, N = 1 only per the contract; cheap enough to always show at N = 1)
        if world == 1:
            ge.build()
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
            port = python_port_rate()
            extras["python_port_steps_per_s_1core"] = port
            # the reference itself cannot travel to this box: its rate is estimated from the Python port's, measured here, by
            # the port / reference ratio measured in the build container (tests/golden/VALIDATION.txt: 3.4-3.7)
            out["cpu_baseline"]["reference_estimate"] = {"value": port / REF_PORT_RATIO, "unit": "env-steps/s", "cores": 1,
                                                         "basis": f"oracle.cpu_ref.RefEnv on 1 core of this box / {REF_PORT_RATIO} "
                                                                  "(RefEnv / reference Game2048Env.step, both 1 core, build container)"}

    # The JSON line must be the LAST thing on stdout.  RCCL prints a version banner through C stdio, which a pipe only
    # flushes at exit -- after Python's own buffer -- and under torch.distributed.run every rank shares one stdout: so
    # every rank flushes its C stdio now, the ranks meet, and only then rank 0 prints.
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:  # pragma: no cover
        pass
    sys.stdout.flush()
    if dist_on:
        dist.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if exchange is not None:
        exchange.close()
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
