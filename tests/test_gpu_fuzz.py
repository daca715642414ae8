"""GPU: a short run of the differential fuzz (tests/fuzz_parity.py: random sizes, configurations and call mixes of the
C ABI, both RNG modes, every output of every call against the oracle).  The long runs are in profiles/r03_fuzz.txt."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_fuzz_call_mixes_vs_oracle(torch_cuda, seed):
    import fuzz_parity
    line = fuzz_parity.run(budget=12.0, seed=seed)
    assert line.startswith("fuzz ok")
    print(line)
