"""Seeding helpers for the numpy-compatible RNG mode.

gymnasium seeds an env with ``numpy.random.Generator(numpy.random.PCG64(numpy.random.SeedSequence(seed)))``
(the reference relies on it: game2048_env.py:103), and SB3's vector envs seed env ``i`` with
``seed + i``.  The SeedSequence hashing is left to numpy itself; this module only lays the resulting
PCG64 states out as the five uint64 planes ``g2048_set_numpy_rng`` expects.
"""
from __future__ import annotations

import numpy as np

_M64 = (1 << 64) - 1


def pcg64_planes(seeds) -> np.ndarray:
    """uint64 ``[5, n]``: state_lo, state_hi, inc_lo, inc_hi, buf of ``PCG64(SeedSequence(seed))``."""
    seeds = [int(s) for s in seeds]
    planes = np.zeros((5, len(seeds)), np.uint64)
    for i, s in enumerate(seeds):
        st = np.random.PCG64(np.random.SeedSequence(s)).state
        state, inc = st["state"]["state"], st["state"]["inc"]
        planes[:, i] = (state & _M64, state >> 64, inc & _M64, inc >> 64, st["uinteger"] | (st["has_uint32"] << 32))
    return planes


def planes_to_generators(planes: np.ndarray):
    """The inverse, for inspection: one ``numpy.random.Generator`` per column of ``planes``."""
    gens = []
    for col in np.asarray(planes, dtype=np.uint64).T:
        bg = np.random.PCG64()
        bg.state = {"bit_generator": "PCG64",
                    "state": {"state": int(col[0]) | (int(col[1]) << 64), "inc": int(col[2]) | (int(col[3]) << 64)},
                    "has_uint32": int(col[4]) >> 32, "uinteger": int(col[4]) & 0xFFFFFFFF}
        gens.append(np.random.Generator(bg))
    return gens
