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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dbs", nargs="+")
    ap.add_argument("--kernel", default="step_kernel")
    ap.add_argument("--traffic-json", default=None,
                    help="write HBM bytes per launch of --kernel from FETCH_SIZE/WRITE_SIZE (+ DRAM_32B cross-check)")
    ap.add_argument("--profile-tag", default=None, help="name of the profile the traffic file belongs to")
    ap.add_argument("--csrc-hash", default=None, help="bench.csrc_hash() of the sources that were profiled")
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
