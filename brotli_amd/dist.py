"""Multi-GPU concatenation of a sharded Brotli stream (component C1).

Rank r encodes the contiguous piece [r * piece, (r + 1) * piece) of one stream
as BROTLI_PARAM_STREAM_OFFSET shards (c/include/brotli/encode.h:231-246); the
compressed pieces are byte-aligned, so the stream is their concatenation in
rank order.  The only collective on the data path is one all-gather of the
sizes and one all-gather of the payloads padded to the largest piece (RCCL over
xGMI with backend "nccl"; "gloo" on CPU tensors in the tests).
"""
import torch
import torch.distributed as dist


def rank_params(rank, world, piece_bytes, total_bytes):
    """(stream_base, is_last, size_hint) of rank `rank`."""
    return rank * piece_bytes, rank == world - 1, min(total_bytes, 1 << 30)


def gather_stream(local, nbytes, group=None, scratch=None, align=256, pad_hint=0):
    """local: 1-D uint8 tensor holding this rank's compressed piece in local[:nbytes].  Returns (padded
    all-gather buffer, sizes as a list of ints, padded_piece_len); the stream is
    cat(buffer[r * pad : r * pad + sizes[r]]).

    Both collectives are enqueued back to back: the payload gather does not wait for the host to read the
    sizes.  Its slot size is `pad_hint` (what the previous step needed — the steps of a job compress the
    same pieces) or, the first time, this rank's own size plus a margin; the ONE host read of the step —
    the gathered sizes, needed to cut the padding off anyway — tells afterwards whether every piece fitted,
    and only a piece that did not makes the gather run again with the right slot."""
    world = dist.get_world_size(group)
    dev = local.device
    sizes = torch.zeros(2 * world, dtype=torch.int64, device=dev)
    mine = torch.tensor([nbytes, int(pad_hint)], dtype=torch.int64, device=dev)       # (the hint travels with the size: checked below)
    dist.all_gather_into_tensor(sizes, mine, group=group)

    def gather(pad, local, scratch):
        if local.numel() < pad:
            grown = torch.zeros(pad, dtype=torch.uint8, device=dev)
            grown[:nbytes] = local[:nbytes]
            local = grown
        if scratch is None or scratch.numel() < world * pad:
            scratch = torch.empty(world * pad, dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(scratch[:world * pad], local[:pad].contiguous(), group=group)
        return scratch

    def round_up(v):
        return (int(v) + align - 1) // align * align
    # every rank must choose the same slot: the hint is the same everywhere (it comes from gathered sizes);
    # without one the slot has to wait for the sizes
    pad = round_up(pad_hint) if pad_hint else 0
    if pad:
        scratch = gather(pad, local, scratch)
    both = [int(v) for v in sizes.cpu()]                     # the step's one host read
    host_sizes, hints = both[0::2], both[1::2]
    if any(h != hints[0] for h in hints):
        # pad_hint must be the same on every rank (compared raw: stricter than the slots it rounds to, so a caller bug
        # shows even while the slots still agree).  sharded_step's hint is the `pad` an earlier
        # step returned, computed from the gathered sizes, hence identical everywhere by construction — the speculative
        # gather above relies on that (it is what keeps the host read out from between the two collectives).  Ranks
        # that get here with different slots were handed hints from somewhere else: their payload gather has already
        # run with mismatched counts, so this is a diagnosis of a caller bug, not a recovery.
        raise ValueError("gather_stream: pad_hint differs between ranks: %r" % (hints,))
    need = round_up(max(host_sizes))
    if need > pad:
        pad = need
        scratch = gather(pad, local, scratch)
    return scratch, host_sizes, pad


def compact(buffer, sizes, pad):
    """Removes the padding: the final contiguous stream (uint8 tensor).  `sizes`: host integers."""
    return torch.cat([buffer[r * pad:r * pad + n] for r, n in enumerate(sizes)])


def sharded_step(encode_local, group=None, scratch=None, pad_hint=0):
    """One step of the N-rank job, the same code for bench.py (RCCL, device tensors) and the gloo
    test (CPU tensors): encode this rank's piece, all-gather, strip the padding.  The pieces of
    one stream compress to within a few per cent of each other, so the padded gather moves hardly
    more than the stream itself; an exact-size gather would cost `world` broadcasts instead of one
    collective.  encode_local() -> (uint8 tensor, nbytes).  Returns (stream, sizes, gather buffer, pad):
    hand `pad` back as pad_hint of the next step."""
    local, nbytes = encode_local()
    buf, sizes, pad = gather_stream(local, nbytes, group=group, scratch=scratch, pad_hint=pad_hint)
    return compact(buf, sizes, pad), sizes, buf, pad


def same_stream_on_all_ranks(stream, group=None):
    """sha256 of the concatenated stream, all-gathered: True iff every rank holds rank 0's bytes."""
    import hashlib
    world = dist.get_world_size(group)
    digest = hashlib.sha256(stream.detach().cpu().numpy().tobytes()).digest()
    mine = torch.frombuffer(bytearray(digest), dtype=torch.uint8).to(stream.device)
    allh = torch.zeros(world * 32, dtype=torch.uint8, device=stream.device)
    dist.all_gather_into_tensor(allh, mine, group=group)
    allh = allh.cpu().view(world, 32)
    return bool((allh == allh[0]).all().item()), digest.hex()
