#!/usr/bin/env python
"""What runs, and when, around the once-per-rollout exchange of a multi-rank bench run.

    rocprofv3 --kernel-trace --output-format csv -d DIR -o r -- \
        env G2048_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-extras --chains 2
    python tools/timeline_tail.py DIR/*/r_kernel_trace.csv

Prints the kernels around the last two returns-only summary kernels (`stats_kernel<false>`: the warm-up's and the timed
region's) with their start times relative to the first printed kernel, durations and hardware queue -- the gaps between
the tiny kernels of the exchange are host latency (profiles/r04_ab_forced_dist_chains.txt)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
summaries = [i for i, r in enumerate(rows) if "stats_kernel<false>" in r["Kernel_Name"]]
for at in summaries[-2:]:
    first = max(at - 6, 0)
    base = int(rows[first]["Start_Timestamp"])
    print("----")
    for r in rows[first:at + 8]:
        start, end = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(start - base) / 1e3:9.1f} us  dur {(end - start) / 1e3:7.1f}  q{r['Queue_Id']}  "
              f"{r['Kernel_Name'][:70]}  grid {r['Grid_Size_X']}")
