"""The tiled quality-5 chain (JOB_FLAG_TILED; k_chain.h tiles / sweeps, k_tile.h) on the GPU box through the HIP
C ABI: long shards whose chain tiles all parse at once.  The reference itself (oracle/_ref, one encoder instance per
shard on the host cores) encodes the same input with the same plan; the sha256 of its concatenated output must be
ours.  BROTLI_AMD_TILE_KB selects the tile size (read at context creation and by hip.refresh_env())."""
import hashlib
import os

import pytest

import gen_inputs as G
from test_gpu_zy_full_size import ROOT, _reference, _threads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch
    torch.cuda.init()
    from brotli_amd import hip
    c = hip.Context(0)
    yield c
    c.close()


def _encode(ctx, data, shard, tile_kb, monkeypatch, lgwin=22):
    import torch
    from brotli_amd import hip
    monkeypatch.setenv("BROTLI_AMD_TILE_KB", str(tile_kb))
    hip.refresh_env()             # (the library reads its knobs when a context is created: this one exists already)
    try:
        n = len(data)
        hint = min(n, 1 << 30)
        params = hip.make_params(5, lgwin, shard, hint)
        d_in = hip.to_device(data, 0)
        d_out = torch.empty(ctx.max_output(n, params), dtype=torch.uint8, device="cuda:0")
        nbytes, info = ctx.encode_device(d_in, n, params, d_out)
        comp = d_out[:nbytes].cpu().numpy().tobytes()
    finally:
        monkeypatch.delenv("BROTLI_AMD_TILE_KB")
        hip.refresh_env()
    return comp, info


@pytest.mark.parametrize("shard_kb,tile_kb", [(1024, 128), (1024, 64), (4095, 128), (300, 64)])
def test_text_long_shards_equal_the_reference(ctx, monkeypatch, shard_kb, tile_kb):
    data = G.enwik_text(256 << 20, seed=G.SEED + 3)
    shard = shard_kb << 10
    comp, info = _encode(ctx, data, shard, tile_kb, monkeypatch)
    ref = _reference(data, 5, 22, shard, min(len(data), 1 << 30), _threads())
    assert len(comp) == ref["out_bytes"] and hashlib.sha256(comp).hexdigest() == ref["sha256"]
    assert info["tile_sweeps"] >= 1                # at least one sweep ran: the tiles were in use
    assert info["tile_fallback_shards"] == 0       # and no shard of this text had to leave the tiled path


def test_tiles_and_plain_chain_agree_on_mixed_data(ctx, monkeypatch):
    """Floats, noise, sparse zeros: unstored positions by the million, shards that give the tiles up (too many
    events, the dictionary gate, a counter wrap) and take the plain chain — the bytes are the reference's either way."""
    data = G.mixed_corpus(192 << 20, seed=G.SEED + 5)
    shard = 1 << 20
    tiled, info = _encode(ctx, data, shard, 128, monkeypatch)
    plain, _ = _encode(ctx, data, shard, 0, monkeypatch)
    assert tiled == plain
    ref = _reference(data, 5, 22, shard, min(len(data), 1 << 30), _threads())
    assert len(tiled) == ref["out_bytes"] and hashlib.sha256(tiled).hexdigest() == ref["sha256"]


def test_english_shards_keep_the_dictionary_gate_open(ctx, monkeypatch):
    """alice29.txt over and over (with synthetic text in between) in 1 MiB shards: behind the first tile of most shards
    the static dictionary's gate is still open — the shard's other tiles start over with it taken as open (k_tile.h:
    k_tile_restart, gate_walk) instead of the shard leaving the tiled path; where it closes on the way, the tile in
    which it does is parsed from the exact counters."""
    alice = open(os.path.join(ROOT, "tests", "golden", "alice29.txt"), "rb").read()
    parts = []
    for k in range(400):
        parts.append(alice[(k * 7919) % 60000:])
        if k % 3 == 0:
            parts.append(bytes(G.enwik_text(90000, seed=500 + k)))
    data = b"".join(parts)[:24 << 20]
    shard = 1 << 20
    comp, info = _encode(ctx, data, shard, 64, monkeypatch)
    ref = _reference(data, 5, 22, shard, min(len(data), 1 << 30), _threads())
    assert len(comp) == ref["out_bytes"] and hashlib.sha256(comp).hexdigest() == ref["sha256"]
    assert info["tile_sweeps"] >= 1


def test_lgwin_18_shards_of_a_window(ctx, monkeypatch):
    """A smaller window: shards of at most (1 << 18) - 16 bytes, four tiles of 64 KiB each."""
    data = G.enwik_text(64 << 20, seed=G.SEED + 7)
    shard = (1 << 18) - 16
    comp, info = _encode(ctx, data, shard, 64, monkeypatch, lgwin=18)
    ref = _reference(data, 5, 18, shard, min(len(data), 1 << 30), _threads())
    assert len(comp) == ref["out_bytes"] and hashlib.sha256(comp).hexdigest() == ref["sha256"]
