"""GPU: a short run of the differential fuzz (tests/fuzz_parity.py: random sizes, configurations and call mixes of the
C ABI, both RNG modes, every output of every call against the oracle).  The long runs are in profiles/r03_fuzz.txt."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,own_streams", [(11, False), (12, True)])
def test_fuzz_call_mixes_vs_oracle(torch_cuda, monkeypatch, seed, own_streams):
    import fuzz_parity
    if own_streams:                     # every case on a fresh non-default (non-blocking) stream
        monkeypatch.setenv("G2048_FUZZ_STREAMS", "1")
    line = fuzz_parity.run(budget=12.0, seed=seed)
    assert line.startswith("fuzz ok")
    print(line)


@pytest.mark.timeout(120)
def test_two_chain_stress_threads_and_streams(torch_cuda):
    """A short run of tests/stress_two_chains.py: three threads, each on its own non-default stream, create, use and
    destroy two-chain engines that share the device's side chain, with pauses that cross the launcher's sleep / wake
    transitions; every rollout equals its one-chain twin."""
    import stress_two_chains
    line = stress_two_chains.run(seconds=10.0, seed=3, n_threads=3)
    assert line.startswith("stress ok"), line
    print(line)
