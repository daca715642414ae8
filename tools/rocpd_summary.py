#!/usr/bin/env python
"""Summarise rocprofv3 (ROCm 7.2, rocpd SQLite output) runs as text for profiles/.

    python tools/rocpd_summary.py <results.db> [more.db ...] [--kernel step_kernel]

For a --kernel-trace --stats run it prints the per-kernel table (calls, total, average, %) and the
duration / back-to-back gap statistics of the kernels matching --kernel; for --pmc runs it prints the
per-dispatch average of every collected counter for those kernels.
"""
import argparse
import sqlite3
import statistics


def concurrency(k, name, bytes_per_item):
    """Kernels of one name that run CONCURRENTLY (two launch chains on two streams): a per-kernel average duration says
    nothing about throughput then.  Per grid size: dispatches, mean duration; over all of them, inside the trains (gaps
    longer than 50 us split trains): the time with 0 / 1 / 2+ of them in flight, and the aggregate rate = work items
    (grid_x lanes = boards) of the kernels of a train / the train's span."""
    by_grid = {}
    for st, en, du, gx, *_ in k:
        by_grid.setdefault(gx, []).append(du / 1e3)
    for gx, d in sorted(by_grid.items()):
        print(f"   grid {gx:>9d}: {len(d):6d} dispatches, mean duration {statistics.mean(d):7.3f} us")
    events = sorted([(st, 1) for st, *_ in k] + [(en, -1) for _, en, *_ in k])
    inflight, last, hist = 0, events[0][0], {}
    for t, delta in events:
        if inflight > 0 or t - last < 50_000:          # (idle gaps between trains are not "0 in flight inside a train")
            hist[min(inflight, 2)] = hist.get(min(inflight, 2), 0) + (t - last)
        last, inflight = t, inflight + delta
    tot = sum(hist.values()) or 1
    print("   time with n kernels in flight (inside trains): " + ", ".join(f"{n}{'+' if n == 2 else ''}: {100 * v / tot:.1f} %" for n, v in sorted(hist.items())))
    # trains: maximal runs of dispatches whose start is < 50 us after the previous end of ANY of them
    trains, cur_start, cur_end, cur_items, cur_n = [], None, None, 0, 0
    for st, en, du, gx, *_ in sorted(k):
        if cur_start is not None and st - cur_end > 50_000:
            trains.append((cur_start, cur_end, cur_items, cur_n))
            cur_start = None
        if cur_start is None:
            cur_start, cur_end, cur_items, cur_n = st, en, 0, 0
        cur_end, cur_items, cur_n = max(cur_end, en), cur_items + gx, cur_n + 1
    trains.append((cur_start, cur_end, cur_items, cur_n))
    big = [t for t in trains if t[3] >= 100]
    if big:
        rate = [t[2] / ((t[1] - t[0]) / 1e9) for t in big]
        items, span = sum(t[2] for t in big), sum(t[1] - t[0] for t in big)
        print(f"   trains of >= 100 dispatches: {len(big)}; aggregate {items / (span / 1e9):.4e} boards/s "
              f"= {bytes_per_item * items / (span / 1e9) / 1e9:.0f} GB/s algorithmic ({bytes_per_item} B per board-step), "
              f"i.e. {span / 1e3 / (items / (1 << 20)):.3f} us per 2^20 board-steps; best train {max(rate):.4e} boards/s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dbs", nargs="+")
    ap.add_argument("--kernel", default="step_kernel")
    ap.add_argument("--traffic-json", default=None,
                    help="write HBM bytes per launch of --kernel from FETCH_SIZE/WRITE_SIZE (+ DRAM_32B cross-check)")
    ap.add_argument("--profile-tag", default=None, help="name of the profile the traffic file belongs to")
    ap.add_argument("--csrc-hash", default=None, help="bench.csrc_hash() of the sources that were profiled")
    ap.add_argument("--concurrency", action="store_true", help="kernels in flight + aggregate throughput of --kernel (two-chain runs)")
    ap.add_argument("--bytes-per-item", type=int, default=38, help="algorithmic bytes per board-step for the aggregate rate")
    args = ap.parse_args()
    collected = {}
    for path in args.dbs:
        db = sqlite3.connect(path)
        cur = db.cursor()
        print(f"== {path}")
        rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
        if rows:
            print(f"{'kernel':90s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'%':>7s}")
            for name, calls, total, avg, pct in rows:
                print(f"{name[:90]:90s} {calls:7d} {total:12.1f} {avg:10.3f} {pct:7.2f}")
        k = list(cur.execute("select start,end,duration,grid_x,workgroup_x,vgpr_count,sgpr_count from kernels "
                             "where name like ? order by start", (f"%{args.kernel}%",)))
        if k:
            d = [r[2] / 1e3 for r in k]
            gaps = [(k[i + 1][0] - k[i][1]) / 1e3 for i in range(len(k) - 1)]
            print(f"-- {args.kernel}: {len(d)} dispatches, grid {k[0][3]} x wg {k[0][4]}, vgpr {k[0][5]}, sgpr {k[0][6]}")
            print(f"   duration us: mean {statistics.mean(d):.3f} median {statistics.median(d):.3f} "
                  f"min {min(d):.3f} max {max(d):.3f}")
            if gaps:
                print(f"   gap to next dispatch us: mean {statistics.mean(gaps):.3f} median {statistics.median(gaps):.3f}")
            if args.concurrency:
                concurrency(k, args.kernel, args.bytes_per_item)
        try:
            pmc = list(cur.execute("select counter_name, avg(value), count(*) from counters_collection "
                                   "where kernel_name like ? group by counter_name", (f"%{args.kernel}%",)))
        except sqlite3.Error:
            pmc = []
        if pmc:
            print(f"-- PMC, average per dispatch of {args.kernel}")
            for name, avg, cnt in pmc:
                print(f"   {name:28s} {avg:18.1f}   (n={cnt})")
                collected[name] = avg
    if args.traffic_json and "FETCH_SIZE" in collected and "WRITE_SIZE" in collected:
        import json
        # profiles/r01_pmc_calibration.txt: FETCH_SIZE (KB) under-reports coalesced reads by exactly 2x on
        # gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE (KB) is exact; *_DRAM_32B x 32 B are exact.
        rd = collected["FETCH_SIZE"] * 1024 * 2
        wr = collected["WRITE_SIZE"] * 1024
        out = {"kernel": args.kernel, "profile": args.profile_tag, "csrc_sha16": args.csrc_hash,
               "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
               "bytes_per_launch": rd + wr,
               "method": "rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE, separate passes"}
        if "TCC_EA0_RDREQ_DRAM_32B" in collected and "TCC_EA0_WRREQ_WRITE_DRAM_32B" in collected:
            out["crosscheck_dram_32b_bytes_per_launch"] = 32 * (collected["TCC_EA0_RDREQ_DRAM_32B"] +
                                                                collected["TCC_EA0_WRREQ_WRITE_DRAM_32B"])
        with open(args.traffic_json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
