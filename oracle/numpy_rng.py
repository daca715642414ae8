"""CPU ORACLE, numpy-compatible RNG mode -- TEST INFRASTRUCTURE ONLY (see cpu_ref.py for the rules).

The reference draws from ``self.np_random`` (game2048_env.py:168,170), which gymnasium creates as
``numpy.random.Generator(numpy.random.PCG64(numpy.random.SeedSequence(seed)))`` [third-party: gymnasium
``utils/seeding.py``; numpy ``random/_pcg64.pyx``, ``_generator.pyx``, ``src/distributions/distributions.c``,
``src/pcg64/pcg64.h``; versions unpinned by the reference's requirements.txt, numpy 2.2.6 here].
The reference itself is absent from none of this -- these are numpy's published algorithms, restated:

* PCG64 = PCG XSL-RR 128/64 (O'Neill 2014): ``state = state * MULT + inc`` (mod 2^128), then
  ``out = rotr64(hi ^ lo, state >> 122)``;
* ``next_uint32`` serves the LOW half of a fresh 64-bit output first and buffers the HIGH half;
* ``Generator.random()`` = ``(next_uint64 >> 11) * 2**-53`` (does not touch the 32-bit buffer);
* ``Generator.shuffle(list)`` = ``for i in n-1 .. 1: j = random_interval(i); swap(x[i], x[j])`` with
  ``random_interval(max)`` = masked rejection on ``next_uint32`` (max <= 0xffffffff).

``tests/test_numpy_rng.py`` pins every one of these against numpy itself.
"""
from __future__ import annotations

from fractions import Fraction

MASK64 = (1 << 64) - 1
MASK128 = (1 << 128) - 1
PCG_MULT = 0x2360ED051FC65DA44385DF649FCCF645  # PCG_DEFAULT_MULTIPLIER_128

# random() < 0.9  <=>  (next_uint64 >> 11) < TWO_THRESHOLD_53, with 0.9 the IEEE double the reference
# compares against (game2048_env.py:168) and k * 2**-53 exact for k < 2**53
_frac = Fraction(0.9) * (1 << 53)
TWO_THRESHOLD_53 = int(_frac) if _frac.denominator == 1 else int(_frac) + 1


class NumpyPCG64:
    """The (state, inc, has_uint32, uinteger) machine behind numpy's Generator(PCG64)."""

    def __init__(self, state: int, inc: int, has_uint32: int = 0, uinteger: int = 0):
        self.state, self.inc, self.has_uint32, self.uinteger = state, inc, has_uint32, uinteger

    @classmethod
    def from_numpy_state(cls, st: dict) -> "NumpyPCG64":
        return cls(st["state"]["state"], st["state"]["inc"], st["has_uint32"], st["uinteger"])

    @classmethod
    def from_seed(cls, seed: int) -> "NumpyPCG64":
        """gymnasium seeding: PCG64(SeedSequence(seed)) -- the hashing is left to numpy itself."""
        import numpy as np
        return cls.from_numpy_state(np.random.PCG64(np.random.SeedSequence(seed)).state)

    def next64(self) -> int:
        self.state = (self.state * PCG_MULT + self.inc) & MASK128
        hi, lo = self.state >> 64, self.state & MASK64
        x, rot = hi ^ lo, self.state >> 122
        return ((x >> rot) | (x << ((-rot) & 63))) & MASK64

    def next32(self) -> int:
        if self.has_uint32:
            self.has_uint32 = 0
            return self.uinteger
        n = self.next64()
        self.has_uint32, self.uinteger = 1, n >> 32
        return n & 0xFFFFFFFF

    def random(self) -> float:
        return (self.next64() >> 11) * (1.0 / 9007199254740992.0)

    def interval(self, mx: int) -> int:
        if mx == 0:
            return 0
        mask = (1 << mx.bit_length()) - 1
        while True:
            v = self.next32() & mask
            if v <= mx:
                return v

    def shuffle(self, x: list) -> None:
        for i in range(len(x) - 1, 0, -1):
            j = self.interval(i)
            x[i], x[j] = x[j], x[i]


def add_tile_numpy(M: list, rng: NumpyPCG64) -> None:
    """game2048_env.py:166-176 on a flat row-major list of tile values, with numpy's draws."""
    val = 2 if (rng.next64() >> 11) < TWO_THRESHOLD_53 else 4        # :168
    positions = list(range(16))                                        # :169 (flat index r*4+c)
    rng.shuffle(positions)                                             # :170
    for p in positions:                                                # :171-175
        if M[p] == 0:
            M[p] = val
            return
    raise AssertionError("No empty cell found")                        # :176
