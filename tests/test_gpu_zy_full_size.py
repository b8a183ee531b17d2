"""Whole-output parity at the sizes BASELINE.json quotes its metric on (1 GiB), on the GPU box:
the reference itself (oracle/_ref/libbrotli_ref.so driven by oracle/_ref/plan_bench, one encoder
instance per shard on the host cores — prebuilt, travels with the snapshot; nothing here reads
/root/reference) encodes the same input with the same plan, and the sha256 of its concatenated
output must be the sha256 of ours.  Sorted late (file name): these are the long tests."""
import hashlib
import json
import os
import subprocess
import tempfile

import pytest

import gen_inputs as G

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libbrotli_ref.so")
DRV = os.path.join(ROOT, "oracle", "_ref", "plan_bench")


def _reference(data, quality, lgwin, shard, hint, threads):
    if not (os.path.exists(REF_SO) and os.path.exists(DRV)):
        pytest.skip("oracle/_ref not built (python -c 'import __graft_entry__ as g; g.build()' where /root/reference exists)")
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    path = os.path.join(tmpdir, "brotli_amd_fullsize_%d.bin" % os.getpid())
    with open(path, "wb") as f:
        f.write(data)
    try:
        r = subprocess.run([DRV, REF_SO, path, str(quality), str(lgwin), str(shard), str(threads), str(hint), "1"],
                           capture_output=True, text=True, check=True, timeout=900)
    finally:
        os.unlink(path)
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.fixture(scope="module")
def ctx():
    import torch
    torch.cuda.init()
    from brotli_amd import hip
    c = hip.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def text1g():
    return G.enwik_text(1 << 30)


def _threads():
    return max(1, min(128, len(os.sched_getaffinity(0))))


def test_1gib_text_q5_lgwin22_128k_plan_sha256(ctx, text1g):
    """BASELINE configs[1]: every one of the 8192 shard sizes and the whole stream's sha256."""
    import torch
    from brotli_amd import hip
    n, shard = 1 << 30, 1 << 17
    p = hip.make_params(5, 22, shard)
    d_in = hip.to_device(text1g)
    d_out = torch.empty(ctx.max_output(n, p), dtype=torch.uint8, device="cuda:0")
    d_sizes = torch.zeros(n // shard, dtype=torch.int64, device="cuda:0")
    nb, info = ctx.encode_device(d_in, n, p, d_out, d_sizes)
    assert info["nshards"] == n // shard and int(d_sizes.sum().item()) == nb
    got = hashlib.sha256(d_out[:nb].cpu().numpy().tobytes()).hexdigest()
    want = _reference(text1g, 5, 22, shard, 1 << 30, _threads())
    assert want["out_bytes"] == nb and want["sha256"] == got


def test_256mib_mixed_corpus_q5_128k_plan_sha256_and_time(ctx):
    """BASELINE configs[3] stand-in (tests/gen_inputs.mixed_corpus: text, markup, source, rows,
    floats, gradients, sparse zeros, noise in 1 MiB members): same bytes as the reference, and
    no member type may throw the serial chain off its pace — the sparse-zero shards (one key run
    longer than the reference's 16-bit store counter) once took 16 s."""
    import torch
    from brotli_amd import hip
    n, shard = 256 << 20, 1 << 17
    data = G.mixed_corpus(n)
    p = hip.make_params(5, 22, shard)
    d_in = hip.to_device(data)
    d_out = torch.empty(ctx.max_output(n, p), dtype=torch.uint8, device="cuda:0")
    nb, info = ctx.encode_device(d_in, n, p, d_out)
    nb, info = ctx.encode_device(d_in, n, p, d_out)
    got = hashlib.sha256(d_out[:nb].cpu().numpy().tobytes()).hexdigest()
    del d_out, d_in
    want = _reference(data, 5, 22, shard, n, _threads())
    assert want["out_bytes"] == nb and want["sha256"] == got
    assert info["ms_total"] < 1000.0, info


def test_1gib_text_q9_lgwin24_512k_plan_sha256(ctx, text1g):
    """BASELINE configs[4] at the plan bench.py --quality 9 runs (512 KiB shards)."""
    import torch
    from brotli_amd import hip
    n, shard = 1 << 30, 1 << 19
    p = hip.make_params(9, 24, shard)
    d_in = hip.to_device(text1g)
    d_out = torch.empty(ctx.max_output(n, p), dtype=torch.uint8, device="cuda:0")
    nb, info = ctx.encode_device(d_in, n, p, d_out)
    got = hashlib.sha256(d_out[:nb].cpu().numpy().tobytes()).hexdigest()
    del d_out, d_in
    want = _reference(text1g, 9, 24, shard, 1 << 30, _threads())
    assert want["out_bytes"] == nb and want["sha256"] == got


def test_1gib_random_q1_one_call_sha256(ctx):
    """BASELINE configs[2]: 1 GiB of random bytes at quality 1, one FINISH call — the full hash,
    not a prefix.  (One reference instance: the reference has no threads.)"""
    import torch
    from brotli_amd import hip
    n = 1 << 30
    g = torch.Generator(device="cuda").manual_seed(G.SEED)
    d_in = torch.zeros(n + hip.INPUT_SLACK, dtype=torch.uint8, device="cuda:0")
    d_in[:n] = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda:0", generator=g)
    d_out = torch.empty(ctx.fast_max_output(n, 1, 22), dtype=torch.uint8, device="cuda:0")
    nbits, _ = ctx.encode_fast_device(d_in, n, d_out, 22)
    assert nbits % 8 == 0
    got = hashlib.sha256(d_out[:nbits // 8].cpu().numpy().tobytes()).hexdigest()
    data = d_in[:n].cpu().numpy().tobytes()
    del d_in, d_out
    want = _reference(data, 1, 22, 0, 0, 1)
    assert want["out_bytes"] == nbits // 8 and want["sha256"] == got
