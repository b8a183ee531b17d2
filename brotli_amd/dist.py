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


def gather_stream(local, nbytes, group=None, scratch=None, align=256):
    """local: 1-D uint8 tensor holding this rank's compressed piece in
    local[:nbytes] (its storage must extend to the padded size).  Returns
    (padded all-gather buffer, sizes tensor, padded_piece_len); the stream is
    cat(buffer[r * pad : r * pad + sizes[r]])."""
    world = dist.get_world_size(group)
    dev = local.device
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    mine = torch.tensor([nbytes], dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, mine, group=group)
    pad = int(sizes.max().item())
    pad = (pad + align - 1) // align * align
    if local.numel() < pad:
        grown = torch.zeros(pad, dtype=torch.uint8, device=dev)
        grown[:nbytes] = local[:nbytes]
        local = grown
    if scratch is None or scratch.numel() < world * pad:
        scratch = torch.empty(world * pad, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(scratch[:world * pad], local[:pad].contiguous(), group=group)
    return scratch, sizes, pad


def compact(buffer, sizes, pad):
    """Removes the padding: the final contiguous stream (uint8 tensor)."""
    parts = [buffer[r * pad:r * pad + int(n)] for r, n in enumerate(sizes.tolist())]
    return torch.cat(parts)
