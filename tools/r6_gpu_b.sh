#!/bin/bash
# round 6: summary kernel (RMW merge) -- correctness, kernel durations by rocprofv3, the exchange A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06b; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q > $O/pytest_round6.log 2>&1; echo "round6 rc $?" >> $O/pytest_round6.log
tail -3 $O/pytest_round6.log
timeout 300 python tools/stats_probe.py > $O/stats_probe_one_launch.txt 2>&1
timeout 300 python tools/stats_probe.py > $O/stats_probe_two_stage.txt 2>&1
cat $O/stats_probe_one_launch.txt $O/stats_probe_two_stage.txt | grep "2^"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt1 -o r -- python tools/stats_probe.py > $O/kt1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt2 -o r -- python tools/stats_probe.py > $O/kt2.log 2>&1
grep -h "summary_kernel\|stats_kernel\|stats_merge" $O/kt1/*kernel_stats.csv $O/kt2/*kernel_stats.csv | cut -c1-220
cp $O/kt1/*kernel_stats.csv $O/kernel_stats_one_launch.csv; cp $O/kt2/*kernel_stats.csv $O/kernel_stats_two_stage.csv
rm -rf $O/kt1 $O/kt2
timeout 300 python tools/dist_probe.py 9 > $O/dist_probe.txt 2>&1; grep "wall" $O/dist_probe.txt
for path in direct torch direct torch direct direct; do
  G2048_BENCH_COLLECTIVE=$path G2048_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras 2>/dev/null | grep '"metric"' > $O/fd_tmp.json
  python - $path $O/fd_tmp.json <<'PY' | tee -a $O/forced_dist_ab.txt
import json,sys
d=json.loads(open(sys.argv[2]).read()); t=d["timing"]
print(f"{sys.argv[1]:6s} {d['value']:.4g} train {t['launch_train_us']:.1f} coll {t['collective_us']:.1f} tail {t['host_tail_us']:.1f} repeats_coll {[round(x,1) for x in t['k_region_repeats_collective_us']]} global {d.get('global_returns',{}).get('return_sum')} local {d['return_sum']}")
PY
done
