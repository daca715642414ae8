#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06l; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_configs.py -x -q -m gpu > $O/pytest_configs.log 2>&1; echo "rc $?" >> $O/pytest_configs.log; tail -3 $O/pytest_configs.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench.err | grep '"metric"' > $O/bench_k20.json
python - $O/bench_k20.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]; t=d["timing"]
print("value %.4g frac %.3f wall %.3f traffic %s tail %.1f prov %s" % (d["value"], r["frac"], r["frac_wall"], r["traffic"], t["host_tail_us"], r["traffic_provenance"]))
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
