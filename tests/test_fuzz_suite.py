"""The offline fuzzers of tools/ at a bounded, seeded count inside the suite (they found the one real difference of
round 2 in the headline kernel only when run by hand).  Every chunk is a test of its own, so pytest-xdist spreads them:
  * tools/fuzz_index_sim.py — the indexed quality-5 parse on the simulator against the oracle: 300 cases, one wave
    layout / search mode per case (the offline tool runs all six);
  * tools/fuzz_abi_sim.py — random call sequences through the BrotliEncoder* boundary next to the reference: 200;
  * tools/fuzz_plan_sim.py — random partition plans (with dictionaries) next to the reference driven shard by shard: 200;
  * tools/fuzz_tiles_sim.py — the tiled chain: tests/test_sim_tiles.py;
  * `-m gpu`: 50 cases of the index fuzzer through the real kernels."""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))

INDEX_BASE, ABI_BASE, PLAN_BASE = 100000, 200000, 300000


@pytest.fixture(scope="module")
def sim():
    from simharness import Sim
    return Sim()


@pytest.mark.parametrize("chunk", range(60))
def test_index_fuzz(sim, oracle, chunk):
    from fuzz_index_sim import make_case
    from test_sim_kernels import IX_LAYOUTS, _oracle_plan
    variants = [(l, e) for l in IX_LAYOUTS.values() for e in (0, 4)]
    for seed in range(INDEX_BASE + 5 * chunk, INDEX_BASE + 5 * chunk + 5):
        data, shard, hint, rev = make_case(seed)
        flags, extra = variants[seed % len(variants)]
        got = sim.encode(data, size_hint=hint, shard_size=shard, reverse=rev, flags=flags | extra)
        assert got == _oracle_plan(oracle, data, hint, shard), (seed, flags, extra)


@pytest.mark.parametrize("chunk", range(10))
def test_abi_fuzz(ref, chunk):
    import fuzz_abi_sim
    bad = [s for s in range(ABI_BASE + 20 * chunk, ABI_BASE + 20 * chunk + 20) if not fuzz_abi_sim.one(s)]
    assert not bad


@pytest.mark.parametrize("chunk", range(40))
def test_plan_fuzz(ref, chunk):
    import fuzz_plan_sim
    bad = [s for s in range(PLAN_BASE + 5 * chunk, PLAN_BASE + 5 * chunk + 5) if not fuzz_plan_sim.one(s)]
    assert not bad


@pytest.mark.gpu
def test_index_fuzz_on_the_gpu(oracle):
    """The same generator through the real kernels (the default wave layout and the forced exact search)."""
    from brotli_amd import hip
    from fuzz_index_sim import make_case
    from test_sim_kernels import _oracle_plan
    ctx = hip.Context(0)
    try:
        for seed in range(INDEX_BASE, INDEX_BASE + 50):
            data, shard, hint, _ = make_case(seed)
            want = _oracle_plan(oracle, data, hint, shard)
            for flags in (0, 4):
                got, _ = ctx.encode_host(data, hip.make_params(5, 22, shard, hint or min(len(data), 1 << 30), flags=flags))
                assert got == want, (seed, flags)
    finally:
        ctx.close()
