#!/usr/bin/env python
"""Timeline of the step kernels in a rocprofv3 kernel trace (tools/ubench/overlap.hip): per queue, when each
kernel started and ended, and how much of the time two kernels were resident at once.
    python tools/overlap_timeline.py <r_kernel_trace.csv>"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "step_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-64:]                                    # the last repetition
t0 = int(rows[0]["Start_Timestamp"])
queues = sorted({r["Queue_Id"] for r in rows})
print("queues:", queues)
for r in rows[:24]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"queue {queues.index(r['Queue_Id'])}  start {s / 1e3:8.2f} us  end {e / 1e3:8.2f} us  dur {(e - s) / 1e3:6.2f} us")
# overlap: sweep line
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1))
    ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
depth, last, hist = 0, ev[0][0], {}
for t, d in ev:
    hist[depth] = hist.get(depth, 0) + (t - last)
    depth += d
    last = t
tot = sum(hist.values())
print("time with k kernels in flight:", {k: f"{v / tot:.0%}" for k, v in sorted(hist.items())},
      f"; span {tot / 1e3:.1f} us for {len(rows)} kernels")
