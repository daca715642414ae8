#!/bin/bash
# Calibrate rocprofv3 memory counters against kernels with known byte counts (run via gpurun).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/pmc_calib; mkdir -p $O; export TMPDIR=/tmp; cd $R
for c in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_DRAM_32B TCC_EA0_WRREQ_WRITE_DRAM_32B "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  d=$O/$(echo $c | tr ' ' '_'); rocprofv3 --kernel-trace --pmc $c -d $d -o r -- tools/ubench/pmc_calib > $d.log 2>&1
  python - <<PY
import sqlite3
db=sqlite3.connect("$d/r_results.db")
for r in db.execute("select kernel_name, counter_name, avg(value) from counters_collection where kernel_name like '%copy_k%' group by kernel_name, counter_name"): print("   ", r[0][:40], r[1], r[2])
PY
  rm -rf $d
done
