"""Offline fuzz of the BrotliEncoder* boundary on the CPU: random call sequences (PROCESS / FLUSH /
EMIT_METADATA / FINISH, random feed sizes, TakeOutput or not, small output chunks) at random
supported parameters through tests/simt/libbrotlienc_sim.so (encode_abi.c + the kernels on the
SIMT simulator) and through the reference library; any difference is printed with its seed.
    python tools/fuzz_abi_sim.py [first_seed] [count]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import gen_inputs as G
from refharness import TABLES
from test_gpu_abi import _bind, drive
from test_sim_kernels import _fuzz_input

os.environ["BROTLI_AMD_TABLES"] = TABLES
sim = _bind(os.path.join(ROOT, "tests", "simt", "libbrotlienc_sim.so"))
stock = _bind(os.path.join(ROOT, "oracle", "_ref", "libbrotli_ref.so"))


def one(seed):
    rng = np.random.default_rng(seed)
    quality = int(rng.choice([1, 1, 2, 3, 4, 5, 5, 5, 6, 7, 8, 9]))
    lgwin = int(rng.choice([10, 12, 14, 16, 17, 18, 20, 22, 24]))
    pieces = [_fuzz_input(rng) for _ in range(int(rng.integers(1, 5)))]
    if rng.integers(0, 3) == 0:
        pieces.append(G.enwik_text(int(rng.integers(20000, 90000)), seed=seed, vocab=3000))
    data = b"".join(pieces)
    ops, off = [], 0
    whole_blocks = rng.integers(0, 5) == 0
    if whole_blocks:
        # calls of whole input blocks and an empty FINISH (the CLI on a file whose size is a multiple of its reads): the
        # reference encodes the last block before it knows the stream ends (encode.c:1700-1712; host_plan.h stream_tail_fix)
        blk = 16384 if quality in (2, 3) else 65536
        nblk = int(rng.integers(1, 7))
        while len(data) < nblk * blk:
            data += G.enwik_text(70000, seed=seed + len(data), vocab=3000)
        data = data[:nblk * blk]
        per = int(rng.choice([1, 1, 2, 3])) * blk
        while off < len(data):
            m = min(per, len(data) - off)
            ops.append((m, 0))
            off += m
    while off < len(data):
        m = int(min(len(data) - off, rng.choice([1, 7, 300, 2048, 5000, 30000, 70000])))
        r = rng.integers(0, 10)
        if r == 0 and m <= 5000:
            ops.append((m, 3))                     # these bytes are a metadata payload
        elif r == 1:
            ops.append((m, 1))
        elif r == 2:
            ops.append((0, int(rng.choice([0, 1, 3]))))
            continue
        else:
            ops.append((m, 0))
        off += m
    if whole_blocks:
        ops.append((0, 2))
    else:
        ops.append((0, 2)) if rng.integers(0, 2) else ops.__setitem__(-1, (ops[-1][0], 2 if ops[-1][1] != 3 else 3))
    if ops[-1][1] != 2:
        ops.append((0, 2))
    params = [(1, quality), (2, lgwin)]
    if quality != 1 and rng.integers(0, 3) == 0:
        params.append((5, int(rng.choice([len(data), 1 << 20, 3 << 20]))))          # SIZE_HINT
    if quality != 1 and rng.integers(0, 4) == 0:
        params.append((9, int(rng.choice([5, 70000, 1 << 22]))))                     # STREAM_OFFSET
    take = bool(rng.integers(0, 2))
    if any(k == 9 for k, _ in params):
        take = False     # reference quirk, not reproduced: with STREAM_OFFSET the flint block must be
                         # pushed out (available_out > 0); in TakeOutput style its stream stops there
    chunk = int(rng.choice([64, 4096, 1 << 16]))
    # attached dictionaries: pieces of the input itself and of unrelated text, attached before a random call
    dictionaries, attach_at = [], 0
    if rng.integers(0, 3) == 0:
        for _ in range(int(rng.integers(1, 4))):
            if rng.integers(0, 2) and len(data) > 64:
                a = int(rng.integers(0, len(data) - 32))
                dictionaries.append(bytes(data[a:a + int(rng.integers(1, 40000))]))
            else:
                dictionaries.append(bytes(G.enwik_text(int(rng.integers(1, 30000)), seed=seed + 7, vocab=3000)))
        attach_at = int(rng.integers(0, len(ops))) if rng.integers(0, 3) == 0 else 0
    def run(L):
        try:
            return drive(L, data, ops, tuple(params), out_chunk=chunk, take=take, dictionaries=dictionaries,
                         attach_before_op=attach_at)
        except AssertionError:
            return None, None                      # the library refused a call (BROTLI_FALSE)
    want, fw = run(stock)
    got, fg = run(sim)
    ok = (want is None and got is None) or (want is not None and fw and fg and got == want)
    if want is None:
        print("   (the reference refuses this sequence)", end="")
    print("seed %d q%d lgwin %d len %d calls %d params %s take %d dicts %s@%d: %s" % (
        seed, quality, lgwin, len(data), len(ops), params[2:], take, [len(d) for d in dictionaries], attach_at,
        "ok" if ok else "MISMATCH"), flush=True)
    return ok


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    bad = [s for s in range(first, first + count) if not one(s)]
    print("mismatching seeds:", bad)
