#!/bin/bash
# Copy the summaries of a tools/gpu_profile.sh run (merged back into gpurun_out/prof_<tag>/) into profiles/
# and make its traffic file the one bench.py reports.  Usage: tools/collect_profile.sh <tag>
set -e
T=$1
cd "$(dirname "$0")/.."
S=gpurun_out/prof_$T
cp $S/summary.txt profiles/${T}_rocprofv3_summary.txt
cp $S/kernel_stats.csv profiles/${T}_kernel_stats.csv
cp $S/traffic.json profiles/${T}_traffic.json
cp $S/traffic.json profiles/traffic_latest.json
cp $S/bench_under_rocprof.json profiles/${T}_bench_under_rocprof.json
[ -f $S/kernel_stats_extras.csv ] && cp $S/kernel_stats_extras.csv profiles/${T}_kernel_stats_extras.csv
[ -f $S/summary_obs_kernel.txt ] && cp $S/summary_obs_kernel.txt profiles/${T}_obs_kernel_summary.txt
[ -f $S/bench_extras_under_rocprof.json ] && cp $S/bench_extras_under_rocprof.json profiles/${T}_bench_extras_under_rocprof.json
python -c "import bench, json; t = json.load(open('profiles/traffic_latest.json')); print('profile', t['profile'], 'csrc', t['csrc_sha16'], 'current', bench.csrc_hash())"
