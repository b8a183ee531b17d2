"""N > 1 path on CPU: two gloo ranks, each holding the compressed piece of its
half of a stream (pieces come from the oracle here — the test exercises the
rank parameters and the all-gather concatenation of brotli_amd.dist, which is
what bench.py --gpus N runs over RCCL)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gen_inputs as G
    from brotli_amd.dist import compact, gather_stream, rank_params
    from refharness import Oracle
    o = Oracle()
    piece, shard = 300000, 1 << 16
    total = piece * world
    data = G.enwik_text(total, seed=21, vocab=5000)
    base, is_last, hint = rank_params(rank, world, piece, total)
    mine = data[base:base + piece]
    parts, off = [], 0
    while off < piece:
        m = min(shard, piece - off)
        parts.append(o.encode_shard(mine[off:off + m], 5, 22, hint, base + off,
                                    is_last and off + m == piece))
        off += m
    comp = b"".join(parts)
    local = torch.zeros(len(comp) + 4096, dtype=torch.uint8)
    local[:len(comp)] = torch.frombuffer(bytearray(comp), dtype=torch.uint8)
    buf, sizes, pad = gather_stream(local, len(comp))
    stream = compact(buf, sizes, pad).numpy().tobytes()
    # every rank holds the same, complete stream
    want_parts, off = [], 0
    while off < total:
        m = min(shard, total - off, piece - off % piece)
        want_parts.append(o.encode_shard(data[off:off + m], 5, 22, hint, off, off + m == total))
        off += m
    ok = stream == b"".join(want_parts)
    q.put((rank, ok, len(stream)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_concatenates_stream(ref):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == res[1][2]
