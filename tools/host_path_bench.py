"""Tool: rates of the host-buffer drop-in paths (numpy in / numpy out) on the MI355X box.
These exist for compatibility; bench.py measures the device-resident path."""
import sys
import time

sys.path.insert(0, '.')
import numpy as np

from gym2048_amd import Game2048Env, Vec2048

env = Game2048Env()
env.reset(seed=42)
rng = np.random.default_rng(0)
acts = rng.integers(0, 4, 3000)
t0 = time.perf_counter()
for a in acts:
    obs, r, term, trunc, info = env.step(int(a))
    if term:
        env.reset()
dt = time.perf_counter() - t0
print("Game2048Env (N=1 drop-in) steps/s:", round(len(acts) / dt))
for dtype in (np.uint8, np.int64):
    for n in (8, 1024, 65536):
        ve = Vec2048(n, seed=1, obs_dtype=dtype)
        ve.reset()
        k = 30 if n > 1024 else 300
        a = rng.integers(0, 4, (k, n))
        ve.step(a[0])
        t0 = time.perf_counter()
        for j in range(k):
            ve.step(a[j])
        dt = time.perf_counter() - t0
        print(f"Vec2048 n={n} (numpy in/out, {np.dtype(dtype).name} obs) env-steps/s:", round(k * n / dt))
        ve.close()
