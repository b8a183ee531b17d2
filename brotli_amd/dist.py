"""Multi-GPU concatenation of a sharded Brotli stream (component C1).

Rank r encodes the contiguous piece [r * piece, (r + 1) * piece) of one stream
as BROTLI_PARAM_STREAM_OFFSET shards (c/include/brotli/encode.h:231-246); the
compressed pieces are byte-aligned, so the stream is their concatenation in
rank order.  The only collective on the data path is ONE all-gather per step:
every rank's slot carries its size in a 16-byte header in front of its padded
piece (RCCL over xGMI with backend "nccl"; "gloo" on CPU tensors in the tests).
"""
import torch
import torch.distributed as dist


def rank_params(rank, world, piece_bytes, total_bytes):
    """(stream_base, is_last, size_hint) of rank `rank`."""
    return rank * piece_bytes, rank == world - 1, min(total_bytes, 1 << 30)


HDR = 16   # bytes in front of every rank's slot: its compressed size and the slot hint it came with (two int64)


class SlotHint(int):
    """The slot size a gather_stream call ended with, as handed back for the next step.  Only such a value — computed
    from sizes every rank has seen, hence the same on all of them — lets the next step start with the payload collective;
    any other hint is a number some caller made up, possibly a different one on every rank, and a collective entered with
    different counts does not raise under RCCL: it hangs.  (the comparison used to come AFTER
    the payload gather.)"""


def gather_stream(local, nbytes, group=None, scratch=None, align=256, pad_hint=0):
    """local: 1-D uint8 tensor holding this rank's compressed piece in local[:nbytes].  Returns (all-gather buffer,
    sizes as a list of ints, slot bytes); the stream is cat(buffer[r * slot + HDR : r * slot + HDR + sizes[r]]).

    ONE all-gather per step: every rank contributes a slot of `pad_hint` bytes (rounded up to `align`) = a 16-byte
    header (its size, its hint) + its piece + padding, so the sizes travel with the payload and the step's one host
    read — the headers, needed to cut the padding off anyway — comes after the collective.  The hint is what the
    previous step needed (the steps of a job compress the same pieces) and must be the SlotHint that step returned; the
    first step of a job has none and asks for the sizes first (one small all-gather more, once), as does a step whose
    piece did not fit its slot, and a step handed a plain number (not trusted to agree between ranks)."""
    world = dist.get_world_size(group)
    dev = local.device

    def round_up(v):
        return (int(v) + align - 1) // align * align

    def gather(slot, scratch):
        send = torch.zeros(slot, dtype=torch.uint8, device=dev)
        send[:HDR] = torch.tensor([nbytes, int(pad_hint)], dtype=torch.int64).view(torch.uint8).to(dev)
        n = min(nbytes, slot - HDR)
        send[HDR:HDR + n] = local[:n]
        if scratch is None or scratch.numel() < world * slot:
            scratch = torch.empty(world * slot, dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(scratch[:world * slot], send, group=group)
        head = scratch[:world * slot].view(world, slot)[:, :HDR].contiguous().cpu()      # the step's one host read
        both = head.view(torch.int64).view(world, 2)
        return scratch, [int(v) for v in both[:, 0]], [int(v) for v in both[:, 1]]

    # Only a hint an earlier gather returned (SlotHint) is the same on every rank by construction: the payload collective
    # goes first with that one.  Anything else asks for the sizes first — a fixed-size collective that cannot mismatch.
    slot = round_up(pad_hint) if isinstance(pad_hint, SlotHint) and pad_hint else 0
    if slot > HDR:
        scratch, sizes, hints = gather(slot, scratch)
        if any(h != hints[0] for h in hints):            # (cannot happen with SlotHint values of one job; kept as an assertion)
            raise ValueError("gather_stream: pad_hint differs between ranks: %r" % (hints,))
        if max(sizes) <= slot - HDR:
            return scratch, sizes, SlotHint(slot)
    else:
        # no trusted hint: the sizes first
        mine = torch.tensor([nbytes], dtype=torch.int64, device=dev)
        allsz = torch.zeros(world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allsz, mine, group=group)
        sizes = [int(v) for v in allsz.cpu()]
    slot = round_up(max(sizes) + HDR)
    pad_hint = slot                                              # (what this gather's headers carry: the same on every rank)
    scratch, sizes, _ = gather(slot, scratch)
    return scratch, sizes, SlotHint(slot)


def compact(buffer, sizes, pad):
    """Removes the headers and the padding: the final contiguous stream (uint8 tensor).  `sizes`: host integers."""
    return torch.cat([buffer[r * pad + HDR:r * pad + HDR + n] for r, n in enumerate(sizes)])


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
