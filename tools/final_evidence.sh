#!/bin/bash
# final evidence of the committed sources (run on the GPU box: gpurun -- bash tools/final_evidence.sh <tag>; then tools/collect_profile.sh <tag>): profile set (per size + policy), bench lines, forced-dist runs, GPU suite
set -u
TAG=${1:-r06_j}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/final_$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "gpu rc $?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
for r in 1 2 3; do
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '"metric"' > $O/bench_k20_$r.json
done
timeout 900 python bench.py 2>/dev/null | grep '"metric"' > $O/bench_default.json
for r in 1 2 3 4 5 6 7 8 9 10 11 12; do
  G2048_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras 2>/dev/null | grep '"metric"' > $O/fd_tmp.json
  python - direct $O/fd_tmp.json <<'PY' | tee -a $O/forced_dist_runs.txt
import json,sys
d=json.loads(open(sys.argv[2]).read()); t=d["timing"]
print(f"{sys.argv[1]:6s} {d['value']:.4g} train {t['launch_train_us']:.1f} coll {t['collective_us']:.1f} tail {t['host_tail_us']:.1f} repeats_coll {[round(x,1) for x in t['k_region_repeats_collective_us']]} global==local {d.get('global_returns',{}).get('return_sum') == d['return_sum']}")
PY
done
for r in 1 2 3 4; do
  G2048_BENCH_COLLECTIVE=torch G2048_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras 2>/dev/null | grep '"metric"' > $O/fd_tmp.json
  python - torch $O/fd_tmp.json <<'PY' | tee -a $O/forced_dist_runs.txt
import json,sys
d=json.loads(open(sys.argv[2]).read()); t=d["timing"]
print(f"{sys.argv[1]:6s} {d['value']:.4g} train {t['launch_train_us']:.1f} coll {t['collective_us']:.1f} tail {t['host_tail_us']:.1f} repeats_coll {[round(x,1) for x in t['k_region_repeats_collective_us']]} global==local {d.get('global_returns',{}).get('return_sum') == d['return_sum']}")
PY
done
cp $O/fd_tmp.json $O/forced_dist_line.json
G2048_PROFILE_POLICY=1 bash tools/gpu_profile.sh $TAG > $O/gpu_profile.log 2>&1
tail -30 $O/gpu_profile.log
