"""The TILED quality-5 chain (JOB_FLAG_TILED: k_chain.h tiles / sweeps, k_tile.h) on the host SIMT simulator against
the oracle: the tiles of a long shard are parsed at once from speculated states, verified, and repaired by sweeps —
the bytes must be the oracle's whatever the tiles guessed.  (The `-m gpu` tests run it at full size through the C ABI.)"""
import os
import sys

import pytest

import gen_inputs as G
from simharness import Sim

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
from test_sim_kernels import _oracle_plan  # noqa: E402

ALICE = open(os.path.join(HERE, "golden", "alice29.txt"), "rb").read()
IX = 2 | 64          # JOB_FLAG_QUAD | JOB_FLAG_INDEXED


@pytest.fixture(scope="module")
def sim():
    return Sim()


def _tiled(monkeypatch, sim, data, shard, tile_kb=64, warm=2048, reverse=0, hint=1 << 30, flags=IX):
    monkeypatch.setenv("SIM_TILE_KB", str(tile_kb))
    monkeypatch.setenv("SIM_TILE_WARM", str(warm))
    return sim.encode(data, 5, 22, hint, shard, reverse=reverse, flags=flags)


@pytest.mark.parametrize("reverse", [0, 1])
def test_text_one_shard_of_several_tiles(sim, oracle, monkeypatch, reverse):
    """Workgroups in either order: a tile never sees what a tile scheduled after it has decided."""
    data = G.enwik_text(300000, seed=3)
    assert _tiled(monkeypatch, sim, data, 0, reverse=reverse) == _oracle_plan(oracle, data, 1 << 30, 0)


def test_plan_with_stream_offsets_and_a_tiny_last_tile(sim, oracle, monkeypatch):
    """Shards behind a stream offset start their blocks at byte 2; 131079 bytes = two tiles and seven bytes."""
    data = G.enwik_text(131079 + 100000, seed=9, vocab=3000)
    assert _tiled(monkeypatch, sim, data, 131079, warm=256) == _oracle_plan(oracle, data, 1 << 30, 131079)


def test_two_blocks_per_tile(sim, oracle, monkeypatch):
    data = G.enwik_text(400000, seed=21)
    assert _tiled(monkeypatch, sim, data, 0, tile_kb=128) == _oracle_plan(oracle, data, 1 << 30, 0)


def test_mixed_data_needs_sweeps(sim, oracle, monkeypatch):
    """Floats / noise / sparse zeros: thousands of unstored positions, joins that fail, several sweeps."""
    data = G.mixed_corpus(300000, seed=7)
    assert _tiled(monkeypatch, sim, data, 0, reverse=1) == _oracle_plan(oracle, data, 1 << 30, 0)


def test_english_keeps_the_dictionary_gate_open_and_leaves_the_tiled_path(sim, oracle, monkeypatch):
    data = ALICE + ALICE[:30000]
    assert _tiled(monkeypatch, sim, data, 0) == _oracle_plan(oracle, data, 1 << 30, 0)


def test_wave_layouts(sim, oracle, monkeypatch):
    data = G.enwik_text(200000, seed=5, vocab=300)
    want = _oracle_plan(oracle, data, 1 << 30, 0)
    for groups in (1, 2):
        assert _tiled(monkeypatch, sim, data, 0, flags=IX | (groups << 8)) == want, groups


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_slice(sim, oracle, seed):
    """A slice of tools/fuzz_tiles_sim.py (run offline over hundreds of seeds)."""
    from fuzz_tiles_sim import one
    assert one(seed, sim, oracle, verbose=False)


def test_shard_of_several_meta_blocks_leaves_the_tiles_and_goes_on(sim, oracle, monkeypatch):
    """More literals than a meta-block holds (encode.c:1141-1166: a cut at max_literals = 64 Ki at lgwin 18): the tiles do
    not know about cuts — the shard takes the plain chain, whose later rounds must follow (it once stopped there)."""
    import numpy as np
    rng = np.random.default_rng(3)
    text = G.enwik_text(200000, seed=13, vocab=3000)
    noise = rng.integers(0, 256, 120000, dtype=np.uint8).tobytes()
    data = b"".join(text[i:i + 5] + noise[3 * (i // 5):3 * (i // 5) + 3] for i in range(0, 200000, 5))[:262000]
    monkeypatch.setenv("SIM_TILE_KB", "64")
    got = sim.encode(data, 5, 18, 1 << 30, 0, flags=IX)
    parts = oracle.encode_shard(data, 5, 18, 1 << 30, 0, True)
    assert got == parts


def test_counter_wrap_leaves_the_tiles_at_once(sim, oracle, monkeypatch):
    """A key run longer than the reference's 16-bit store counter (300 000 zeros): the tile that meets it stops, the shard
    goes the plain way (k_chain.h: c_search_exact counts the wraps) — same bytes."""
    data = G.enwik_text(70000, seed=2) + bytes(300000) + G.enwik_text(70000, seed=3)
    assert _tiled(monkeypatch, sim, data, 0) == _oracle_plan(oracle, data, 1 << 30, 0)
