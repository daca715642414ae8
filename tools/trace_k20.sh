export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tk -o r -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > /tmp/tk.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/tk/**/r_kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "step_kernel" in r["Kernel_Name"]]
# last 25 step_kernel dispatches = 5 warm-up + 20 timed
sel=idx[-25:]
prev_end=None
for i in sel:
    r=rows[i]; s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    gap = (s-prev_end)/1e3 if prev_end else 0
    print(f"dur {(e-s)/1e3:7.2f} us  gap_before {gap:9.2f} us")
    prev_end=e
PY
