"""N > 1 path on CPU: two gloo ranks run brotli_amd.dist.sharded_step — the very function
bench.py --gpus N runs over RCCL — with the oracle standing in for the device encoder
(no GPU here): rank parameters, all-gather, padding removal, and the sha256 agreement
check bench.py reports."""
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
    from brotli_amd.dist import rank_params, same_stream_on_all_ranks, sharded_step
    from refharness import Oracle
    o = Oracle()
    piece, shard = 600000, 1 << 16      # total >= 1 MiB: the size hint decides the hasher (H68, not H58: quality.h:186-204)
    total = piece * world
    data = G.enwik_text(total, seed=21, vocab=5000)
    base, is_last, size_hint = rank_params(rank, world, piece, total)
    assert size_hint == total >= (1 << 20)
    mine = data[base:base + piece]
    parts, off = [], 0
    while off < piece:
        m = min(shard, piece - off)
        parts.append(o.encode_shard(mine[off:off + m], 5, 22, size_hint, base + off,
                                    is_last and off + m == piece))
        off += m
    comp = b"".join(parts)

    def encode_local():
        local = torch.zeros(len(comp) + 4096, dtype=torch.uint8)
        local[:len(comp)] = torch.frombuffer(bytearray(comp), dtype=torch.uint8)
        return local, len(comp)
    from brotli_amd.dist import SlotHint
    scratch, pad = None, 0
    for slot in (0, None, SlotHint(256)):   # first step: no hint; then the previous step's slot, as bench.py's steps do;
        #                                     then a hint that is too small (the gather must notice and run again)
        stream_t, sizes, scratch, pad = sharded_step(encode_local, scratch=scratch, pad_hint=pad if slot is None else slot)
        assert isinstance(pad, SlotHint)
    same, _ = same_stream_on_all_ranks(stream_t)
    stream = stream_t.numpy().tobytes()
    # every rank holds the same, complete stream
    want_parts, off = [], 0
    while off < total:
        m = min(shard, total - off, piece - off % piece)
        want_parts.append(o.encode_shard(data[off:off + m], 5, 22, size_hint, off, off + m == total))
        off += m
    ok = same and stream == b"".join(want_parts) and sum(sizes) == len(stream)
    # (the hint matters: with one below 1 MiB the shards come out differently)
    ok = ok and o.encode_shard(data[:shard], 5, 22, 256, 0, False) != want_parts[0]
    # a hint that is a plain number — here a different one on every rank, slots of different sizes — is not trusted:
    # the step asks for the sizes first (under RCCL a payload collective entered with different counts would hang) and
    # still returns the stream
    stream2, sizes2, scratch, pad = sharded_step(encode_local, scratch=scratch, pad_hint=5000 * (rank + 1))
    ok = ok and stream2.numpy().tobytes() == stream and isinstance(pad, SlotHint)
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


def _worker_mix(rank, world, port, q):
    """bench.py --workload silesia --gpus N in small: every rank has a piece of its OWN (the Silesia-style mix, seed +
    rank), the stream is the pieces in rank order; three steps, the slot hint handed on as bench.py's steps do."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gen_inputs as G
    from brotli_amd.dist import rank_params, same_stream_on_all_ranks, sharded_step
    from refharness import Oracle
    o = Oracle()
    piece, shard = 96 << 10, 32 << 10
    total = piece * world
    base, is_last, size_hint = rank_params(rank, world, piece, total)

    def encode_piece(r):
        data = G.mixed_corpus(piece, seed=G.SEED + r)
        b, last, hint = rank_params(r, world, piece, total)
        return b"".join(o.encode_shard(data[off:off + shard], 5, 22, hint, b + off, last and off + shard >= piece)
                        for off in range(0, piece, shard))
    comp = encode_piece(rank)

    def encode_local():
        local = torch.zeros(len(comp) + 4096, dtype=torch.uint8)
        local[:len(comp)] = torch.frombuffer(bytearray(comp), dtype=torch.uint8)
        return local, len(comp)
    scratch, pad, stream_t, sizes = None, 0, None, None
    for _ in range(3):
        stream_t, sizes, scratch, pad = sharded_step(encode_local, scratch=scratch, pad_hint=pad)
    same, _ = same_stream_on_all_ranks(stream_t)
    ok = same and sizes[rank] == len(comp) and sum(sizes) == stream_t.numel()
    if rank == 0:
        # rank 0 checks the whole stream against every rank's piece encoded here, and that it decodes to the inputs
        want = b"".join(encode_piece(r) for r in range(world))
        ok = ok and stream_t.numpy().tobytes() == want
        from refharness import Ref, have_ref
        if have_ref():
            plain = b"".join(G.mixed_corpus(piece, seed=G.SEED + r) for r in range(world))
            ok = ok and Ref().decompress(want, len(plain)) == plain
    q.put((rank, ok, int(stream_t.numel())))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_mix_pieces_concatenate_to_one_stream(ref):
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_mix, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert len({n for _, _, n in res}) == 1
