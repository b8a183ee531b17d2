"""A long meta-block built and written by many waves (brotli_amd/csrc/k_wide.h: parts of 4096 commands / literals,
scans over the parts' totals, the three block splitters side by side, prefix-code jobs spread over the waves, the
command stream of every part at its bit offset) on the host SIMT simulator: the bytes must be the oracle's / the
reference library's — the same as the one-wave kernels give (k_build / k_store; the two share their pieces).  SIM_WIDE=K
sends EVERY meta-block of the simulator's drivers through the many-wave kernels: plans, single streams fed in pieces,
tiled streams, through the BrotliEncoder* boundary."""
import os
import sys

import numpy as np
import pytest

import gen_inputs as G
from simharness import Sim

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def sim():
    return Sim()


@pytest.fixture(autouse=True, params=[3, 64])
def wide(request, monkeypatch):
    """Waves per meta-block: 3 (every wave walks several parts) and the most there are (most waves find no part)."""
    monkeypatch.setenv("SIM_WIDE", str(request.param))


def alice():
    return open(os.path.join(HERE, "golden", "alice29.txt"), "rb").read()


def cases():
    rng = np.random.default_rng(11)
    text = bytes(G.enwik_text(700000, seed=4))
    noise = bytes(rng.integers(0, 256, 60000, dtype=np.uint8))
    return {
        "alice": alice(),                                              # 5 command parts, 13 literal contexts
        "text700k": text,                                              # 20 parts; several block types per category
        "mix": bytes(G.mixed_corpus(900000, seed=5)),                  # literal-heavy members: dozens of literal parts
        "noise_in_text": text[:200000] + noise + text[200000:400000],  # one command with 60000 literals: the step leaves the LDS window
        "noise": noise,                                                # ShouldCompress says no: the raw path of k_wide_tail
        "tiny": alice()[:70],
        "exactly_one_part": bytes(G.enwik_text(33000, seed=9)),
        "zeros": bytes(300000),                                        # one command per block
    }


@pytest.mark.parametrize("name", list(cases()))
@pytest.mark.parametrize("reverse", [0, 1])
def test_one_shard_equals_the_oracle(sim, oracle, name, reverse):
    data = cases()[name]
    assert sim.encode(data, 5, 22, 0, 0, reverse=reverse) == oracle.encode_plan(data, 5, 22, 0)


@pytest.mark.parametrize("quality,lgwin,shard", [(5, 22, 65536), (5, 18, 0), (6, 22, 100000), (7, 20, 0), (9, 24, 0), (9, 16, 0),
                                                  (4, 22, 0), (3, 20, 50000), (2, 18, 0)])
def test_other_qualities_and_plans(sim, oracle, quality, lgwin, shard):
    """Every parse kernel in front of the many-wave build / store: plans (each shard a meta-block of its own), the deep
    hashers, the forgetful chain, the quick family with its count-only / static codes and single-block splitters."""
    data = cases()["text700k"][:400000]
    assert sim.encode(data, quality, lgwin, 0, shard) == oracle.encode_plan(data, quality, lgwin, shard)


def test_shard_of_several_meta_blocks(sim, oracle):
    """lgwin 17: the meta-block limit is 256 KiB, a 700 kB shard has three — the rounds loop runs the wide kernels once
    per meta-block, each continuing the bit stream where the one before ended (carried bits, out_bytes not dword aligned)."""
    data = cases()["text700k"]
    assert sim.encode(data, 5, 17, 0, 0) == oracle.encode_plan(data, 5, 17, 0)


@pytest.mark.parametrize("reverse", [0, 1])
def test_tiled_stream(sim, ref, reverse):
    """A stock one-shot call longer than the window (the tiled stream, k_tile.h): its meta-blocks are written as if each
    began at bit 0 and moved to their bit offsets afterwards — by the wide kernels here."""
    data = bytes(G.enwik_text(430000, seed=2))
    got, info = sim.encode_stream(data, lgwin=17, reverse=reverse)
    assert got is not None and got == ref.compress(data, 5, 17)
    assert info[2] >= 2


def test_tiled_stream_with_a_raw_meta_block(sim, ref):
    rng = np.random.default_rng(3)
    text = bytes(G.enwik_text(1500000, seed=12))
    data = text[:700000] + bytes(rng.integers(0, 256, 600000, dtype=np.uint8)) + text[700000:]
    got, info = sim.encode_stream(data, lgwin=17)
    assert got is not None and got == ref.compress(data, 5, 17)


def test_single_stream_fed_in_pieces(sim, ref):
    """The serial device stream (k_parse.h) with PROCESS / FLUSH / FINISH: meta-blocks that start in the middle of an
    output byte and dword; next to the reference library driven with the same calls."""
    from test_gpu_abi import _bind, drive
    data = cases()["text700k"][:300000]
    calls = [(70000, 0), (50000, 1), (1, 1), (100000, 0), (79999, 2)]
    stock = _bind(os.path.join(ROOT, "oracle", "_ref", "libbrotli_ref.so"))
    want, fin = drive(stock, data, calls, ((1, 5), (2, 22)))
    assert fin
    assert sim.stream(data, calls, quality=5, lgwin=22) == bytes(want)


def test_through_the_boundary(ref, monkeypatch):
    """encode_abi.c over the simulator-backed HIP layer (tests/simt/sim_hip_layer.cc): the stock one-shot call (a
    tiled shard inside the window, a tiled stream beyond it) and a partition plan, next to the reference library."""
    import ctypes as C
    import subprocess
    from refharness import TABLES
    from test_gpu_abi import _bind, drive
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "simt")], check=True)
    monkeypatch.setenv("BROTLI_AMD_TABLES", TABLES)
    L = _bind(os.path.join(ROOT, "tests", "simt", "libbrotlienc_sim.so"))
    data = cases()["text700k"][:350000]

    def one_shot(lgwin):
        cap = L.BrotliEncoderMaxCompressedSize(len(data))
        out = C.create_string_buffer(cap)
        n = C.c_size_t(cap)
        assert L.BrotliEncoderCompress(5, lgwin, 0, len(data), data, C.byref(n), out)
        return out.raw[:n.value]
    assert one_shot(22) == ref.compress(data, 5, 22)
    assert one_shot(17) == ref.compress(data, 5, 17)
    got, fin = drive(L, data, [(len(data), 2)], params=((0x4D490001, 64 << 10),))
    assert fin and bytes(got) == ref.encode_plan(data, 5, 22, 64 << 10)
