#!/bin/bash
# round 6: the whole GPU suite, then the driver's command line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06e; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "gpu rc $?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_k20.out 2> $O/bench_k20.err; echo "bench rc $?"
grep '"metric"' $O/bench_k20.out > $O/bench_k20.json
python - $O/bench_k20.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("value %.4g frac %.3f wall %.3f" % (d["value"], d["roofline"]["frac"], d["roofline"]["frac_wall"]))
print("roofline.streaming", d["roofline"].get("streaming")); print("small", d["roofline"].get("small_batch"))
print("cpu", {k:d["cpu_baseline"][k] for k in ("value","cores","per_core_value","threads_tried","cpu_model","single_core_value")})
ex=d["extras"]
for k in ("step_with_obs_u8","step_with_obs_f16","step_with_obs_f32","numpy_rng_mode_steps_per_s","batch_65536","streaming_2p24"): print(k, ex.get(k))
print("policy", json.dumps(ex.get("policy_loop"))[:1500])
print("host", d["host"])
PY
