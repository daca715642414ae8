#!/bin/bash
# Copy the summaries of a tools/gpu_profile.sh run (merged back into gpurun_out/prof_<tag>/) into profiles/
# and make its traffic file the one bench.py reports.  Usage: tools/collect_profile.sh <tag>
set -e
T=$1
cd "$(dirname "$0")/.."
S=gpurun_out/prof_$T
cp $S/summary.txt profiles/${T}_rocprofv3_summary.txt
cp $S/kernel_stats.csv profiles/${T}_kernel_stats.csv
# stamp the traffic file with the kernel-trace average of the same run (what bench.py prints as roofline.profiled_launch_us)
python - "$S" <<'PY'
import csv, json, sys
s = sys.argv[1]
t = json.load(open(f"{s}/traffic.json"))
for row in csv.DictReader(open(f"{s}/kernel_stats.csv")):
    if "step_kernel<1, true, true, false>" in row["Name"]:
        t["kernel_trace_avg_us"] = float(row["AverageNs"]) / 1e3
        t["kernel_trace_dispatches"] = int(row["Calls"])
        break
json.dump(t, open(f"{s}/traffic.json", "w"), indent=1)
PY
cp $S/traffic.json profiles/${T}_traffic.json
cp $S/traffic.json profiles/traffic_latest.json
cp $S/bench_under_rocprof.json profiles/${T}_bench_under_rocprof.json
[ -f $S/summary_two_chains.txt ] && cp $S/summary_two_chains.txt profiles/${T}_two_chains_summary.txt
[ -f $S/kernel_stats_two_chains.csv ] && cp $S/kernel_stats_two_chains.csv profiles/${T}_kernel_stats_two_chains.csv
[ -f $S/bench_two_chains_under_rocprof.json ] && cp $S/bench_two_chains_under_rocprof.json profiles/${T}_bench_two_chains_under_rocprof.json
[ -f $S/kernel_stats_extras.csv ] && cp $S/kernel_stats_extras.csv profiles/${T}_kernel_stats_extras.csv
[ -f $S/summary_obs_kernel.txt ] && cp $S/summary_obs_kernel.txt profiles/${T}_obs_kernel_summary.txt
[ -f $S/bench_extras_under_rocprof.json ] && cp $S/bench_extras_under_rocprof.json profiles/${T}_bench_extras_under_rocprof.json
for L in 2p16 2p24; do    # the step kernel per batch size (round 6): kernel trace, FETCH / WRITE summary, the line under the profiler
  [ -f $S/kernel_stats_$L.csv ] && cp $S/kernel_stats_$L.csv profiles/${T}_kernel_stats_$L.csv
  [ -f $S/summary_$L.txt ] && cp $S/summary_$L.txt profiles/${T}_summary_$L.txt
  [ -f $S/bench_${L}_under_rocprof.json ] && cp $S/bench_${L}_under_rocprof.json profiles/${T}_bench_${L}_under_rocprof.json
done
cp $S/kernel_stats.csv profiles/${T}_kernel_stats_2p20.csv    # (the main set IS the 2^20 one: same file under the per-size name)
[ -f $S/kernel_stats_policy.csv ] && cp $S/kernel_stats_policy.csv profiles/${T}_kernel_stats_policy.csv
[ -f $S/policy_naive_conv_check.txt ] && cp $S/policy_naive_conv_check.txt profiles/${T}_policy_naive_conv_check.txt
[ -f $S/policy_under_rocprof.json ] && cp $S/policy_under_rocprof.json profiles/${T}_policy_under_rocprof.json
python -c "import bench, json; t = json.load(open('profiles/traffic_latest.json')); print('profile', t['profile'], 'csrc', t['csrc_sha16'], 'current', bench.csrc_hash())"
