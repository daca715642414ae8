#!/bin/bash
# round 6: the rest of the GPU suite from where it stopped + the new fused numpy tests + numpy timing
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06f; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_round6.py tests/test_gpu_parity.py -x -q -m gpu -k "numpy or strict or argument_validation" > $O/pytest_a.log 2>&1; echo "a rc $?" >> $O/pytest_a.log; tail -4 $O/pytest_a.log
timeout 300 python - > $O/numpy_fused_probe.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, ".")
import torch
from gym2048_amd.batched import Batched2048
n, k = 1 << 20, 128
e = Batched2048(n, seed=42, rng="numpy"); e.reset()
acts = e.random_actions(k)
rew = torch.zeros((k, n), dtype=torch.float32, device=e.device); term = torch.zeros((k, n), dtype=torch.uint8, device=e.device)
for fused in (False, True, False, True):
    plan = e.prepare_rollout(acts, reward=rew, terminated=term, fused=fused)
    plan.run(); torch.cuda.synchronize()
    t0 = time.perf_counter(); plan.run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"numpy-RNG mode 2^20 x {k} steps, {'ONE fused launch' if fused else 'one launch per step'}: {k*n/dt:.3e} env-steps/s, {dt/k*1e6:.1f} us per step")
t0 = time.perf_counter(); e.rollout_random(k); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"g2048_rollout_random (numpy mode): {k*n/dt:.3e} env-steps/s")
PY
cat $O/numpy_fused_probe.txt | grep -v amdgpu
timeout 2400 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_round6.py > $O/pytest_gpu.log 2>&1; echo "gpu rc $?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
