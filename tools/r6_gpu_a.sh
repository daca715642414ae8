#!/bin/bash
# round 6, first GPU pass: new tests, the summary kernel A/B, the exchange A/B under the forced one-rank process group
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q > $O/pytest_round6.log 2>&1; echo "round6 rc $?" >> $O/pytest_round6.log
tail -5 $O/pytest_round6.log
timeout 120 tests/c_abi_client/client > $O/c_client.log 2>&1; echo "client rc $?" >> $O/c_client.log; tail -2 $O/c_client.log
timeout 300 python tools/stats_probe.py > $O/stats_probe_one_launch.txt 2>&1
timeout 300 python tools/stats_probe.py > $O/stats_probe_two_stage.txt 2>&1
cat $O/stats_probe_one_launch.txt $O/stats_probe_two_stage.txt
timeout 300 python tools/dist_probe.py 9 > $O/dist_probe.txt 2>&1; cat $O/dist_probe.txt
for path in direct torch direct torch direct torch; do
  G2048_BENCH_COLLECTIVE=$path G2048_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras 2>/dev/null | grep '"metric"' > $O/fd_tmp.json
  python - $path $O/fd_tmp.json <<'PY' | tee -a $O/forced_dist_ab.txt
import json,sys
d=json.loads(open(sys.argv[2]).read()); t=d["timing"]
print(f"{sys.argv[1]:6s} {d['value']:.4g} train {t['launch_train_us']:.1f} coll {t['collective_us']:.1f} tail {t['host_tail_us']:.1f} repeats_coll {[round(x,1) for x in t['k_region_repeats_collective_us']]} global {d.get('global_returns',{}).get('return_sum')} local {d['return_sum']}")
PY
done
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "gpu rc $?" >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
