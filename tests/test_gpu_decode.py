"""The device decoder (k_decode.h) on the GPU, through the C ABI (brotli_amd_decode_device / _host):
reference streams at every quality, the shards of a plan as independent pieces, a round trip of the
device encoder at 64 MiB that never leaves HBM, damaged input."""
import pytest

import gen_inputs as G

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900, method="thread")]


@pytest.fixture(scope="module")
def ctx():
    import torch
    torch.cuda.init()       # same initialisation order as bench.py and test_gpu_parity.py: torch first
    from brotli_amd import hip
    c = hip.Context(0)
    yield c
    c.close()


def test_decodes_reference_streams(ctx, ref):
    import os
    alice = open(os.path.join(os.path.dirname(__file__), "golden", "alice29.txt"), "rb").read()
    inputs = [alice, bytes(G.mixed_corpus(400000)), bytes(G.random_bytes(50000)), bytes(100000), b"x", b""]
    for data in inputs:
        for quality, lgwin in ((0, 22), (1, 18), (2, 22), (4, 10), (5, 22), (6, 16), (9, 24), (11, 22)):
            if quality == 11 and len(data) > 200000:
                continue
            comp = ref.compress(data, quality, lgwin)
            out, res = ctx.decode_host(comp, len(data))
            assert res[0] == (len(data), 0, 1) and out == data, (len(data), quality, lgwin)


def test_device_round_trip_of_a_plan(ctx):
    """Encode 64 MiB on the device (512 shards), decode the shards as 512 concurrent pieces, compare
    on the device; then the same output as ONE stream on one wave (a prefix, to keep it short)."""
    import torch
    from brotli_amd import hip
    n, shard = 64 << 20, 128 << 10
    data = G.mixed_corpus(n, seed=17)
    d_in = hip.to_device(data, 0)
    for quality, lgwin in ((5, 22), (9, 22), (2, 18)):
        params = hip.make_params(quality, lgwin, shard, n)
        d_out = torch.empty(ctx.max_output(n, params), dtype=torch.uint8, device="cuda:0")
        d_sizes = torch.zeros(n // shard, dtype=torch.int64, device="cuda:0")
        nbytes, _ = ctx.encode_device(d_in, n, params, d_out, d_sizes)
        sizes = d_sizes.cpu().tolist()
        assert sum(sizes) == nbytes and nbytes + hip.DECODE_SLACK <= d_out.numel()
        d_back = torch.zeros(n + 64, dtype=torch.uint8, device="cuda:0")
        res, ms = ctx.decode_device(d_out, nbytes, d_back, n, hip.plan_pieces(sizes, n, shard, lgwin))
        assert all(r[1] == 0 for r in res) and res[-1][2] == 1 and [r[0] for r in res] == [shard] * len(res)
        assert torch.equal(d_back[:n], d_in[:n]), (quality, lgwin)
        print("decode q%d: %d pieces, %.2f ms, %.1f GB/s of output" % (quality, len(res), ms, n / 1e6 / ms))
    # the concatenation is one valid stream: the first shards of it on a single wave
    k = 16
    part = sum(sizes[:k])
    d_back.zero_()
    res, _ = ctx.decode_device(d_out, part, d_back, k * shard, None, check=False)
    assert res[0][0] == k * shard and res[0][1] == 0 and res[0][2] == 0     # input ends before ISLAST
    assert torch.equal(d_back[:k * shard], d_in[:k * shard])


def test_quality_1_round_trip(ctx):
    from brotli_amd import hip
    data = bytes(G.enwik_text(3 << 20, seed=8, vocab=20000))
    comp, nbits, _ = ctx.encode_fast_host(data, lgwin=18)
    out, res = ctx.decode_host(comp, len(data))
    assert res[0] == (len(data), 0, 1) and out == data


def test_damaged_streams_report_errors(ctx, ref):
    import random
    data = bytes(G.mixed_corpus(200000))
    rng = random.Random(3)
    comp = ref.compress(data, 5, 22)
    for cut in (1, len(comp) // 2, len(comp) - 1):
        out, res = ctx.decode_host(comp[:cut], len(data), check=False)
        assert not (res[0][1] == 0 and res[0][2] == 1)
    for _ in range(20):
        bad = bytearray(comp)
        bad[rng.randrange(len(bad))] ^= 1 << rng.randrange(8)
        out, res = ctx.decode_host(bytes(bad), len(data), check=False)
        assert res[0][0] <= len(data)
    out, res = ctx.decode_host(comp, len(data) - 1, check=False)      # too little room
    assert res[0][1] != 0
