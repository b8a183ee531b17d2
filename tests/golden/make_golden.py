"""Generates tests/golden/golden.json from the REAL reference encoder
(oracle/_ref/libbrotli_ref.so, built from /root/reference by oracle/Makefile).
Run in the build container:  python tests/golden/make_golden.py
The fixtures pin both the C restatement and the HIP path on the GPU box,
where /root/reference does not exist."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import gen_inputs as G  # noqa: E402
from refharness import Ref  # noqa: E402

CASES = [
    ({"kind": "file", "name": "alice29.txt"}, 5, 22, 0),
    ({"kind": "file", "name": "alice29.txt"}, 5, 22, 65536),
    ({"kind": "text", "size": 1 << 20, "seed": 11}, 5, 22, 0),
    ({"kind": "text", "size": 1 << 20, "seed": 11}, 5, 22, 1 << 18),
    ({"kind": "text", "size": (1 << 20) + 3, "seed": 3}, 5, 22, 1 << 17),
    ({"kind": "text", "size": 3 << 20, "seed": 7}, 5, 22, 1 << 20),
    ({"kind": "text", "size": 3 << 20, "seed": 7}, 6, 22, 1 << 20),
    ({"kind": "mixed", "size": 2 << 20, "seed": 9}, 5, 22, 1 << 19),
    ({"kind": "random", "size": 1 << 18, "seed": 1}, 5, 22, 1 << 16),
    ({"kind": "repeat", "unit": "abcdefgh", "size": 333333}, 5, 22, 100000),
    ({"kind": "text", "size": 1 << 20, "seed": 11}, 9, 24, 0),
    ({"kind": "text", "size": 1 << 20, "seed": 11}, 7, 22, 1 << 18),
    # qualities 2 - 4: the quickly hashers (H54 at quality 4 from a MiB on), trivial / fast writers
    ({"kind": "file", "name": "alice29.txt"}, 2, 22, 0),
    ({"kind": "file", "name": "alice29.txt"}, 3, 22, 0),
    ({"kind": "file", "name": "alice29.txt"}, 4, 22, 0),
    ({"kind": "file", "name": "alice29.txt"}, 4, 16, 65536),
    ({"kind": "text", "size": 1 << 20, "seed": 11}, 2, 22, 1 << 18),
    ({"kind": "text", "size": 1 << 20, "seed": 11}, 3, 18, 0),
    ({"kind": "text", "size": 1 << 20, "seed": 11}, 4, 22, 1 << 17),
    ({"kind": "mixed", "size": 2 << 20, "seed": 9}, 4, 24, 1 << 19),
    ({"kind": "mixed", "size": 1 << 20, "seed": 9}, 2, 10, 0),
    ({"kind": "random", "size": 1 << 18, "seed": 1}, 3, 22, 1 << 16),
    # qualities 5 - 9 at windows of 10 - 16 bits: the forgetful-chain hashers H40 / H41 / H42
    ({"kind": "file", "name": "alice29.txt"}, 5, 16, 0),
    ({"kind": "file", "name": "alice29.txt"}, 9, 10, 0),
    ({"kind": "text", "size": 1 << 20, "seed": 11}, 6, 16, 1 << 17),
    ({"kind": "text", "size": 1 << 20, "seed": 11}, 7, 14, 0),
    ({"kind": "mixed", "size": 1 << 20, "seed": 9}, 9, 16, 1 << 18),
]


# quality 1 (the two-pass fragment compressor): (input, lgwin, KiB per CompressStream call; 0 = one FINISH call)
Q1_CASES = [
    ({"kind": "file", "name": "alice29.txt"}, 22, 0),
    ({"kind": "text", "size": 1 << 20, "seed": 11}, 22, 0),
    ({"kind": "text", "size": (1 << 20) + 3, "seed": 3}, 18, 0),
    ({"kind": "text", "size": 3 << 20, "seed": 7}, 22, 512),          # the CLI's feed pattern, with the empty FINISH
    ({"kind": "mixed", "size": 2 << 20, "seed": 9}, 20, 0),
    ({"kind": "random", "size": 1 << 18, "seed": 1}, 22, 64),
    ({"kind": "repeat", "unit": "abcdefgh", "size": 333333}, 16, 0),
]


# attached dictionaries (BrotliEncoderPrepareDictionary(RAW) + AttachPreparedDictionary), one FINISH call:
# (gen_inputs.dictionary_case arguments, quality, lgwin)
DICT_CASES = [
    ({"nbytes": 200000, "dict_bytes": 80000, "nchunks": 1, "seed": 41}, 5, 22),
    ({"nbytes": 300000, "dict_bytes": 250000, "nchunks": 3, "seed": 42}, 5, 18),
    ({"nbytes": 3000, "dict_bytes": 100000, "nchunks": 2, "seed": 43}, 5, 22),
    ({"nbytes": 200000, "dict_bytes": 80000, "nchunks": 2, "seed": 44}, 9, 24),
    ({"nbytes": 200000, "dict_bytes": 80000, "nchunks": 1, "seed": 45}, 7, 14),
    ({"nbytes": 200000, "dict_bytes": 80000, "nchunks": 1, "seed": 46}, 3, 22),
    ({"nbytes": 200000, "dict_bytes": 80000, "nchunks": 2, "seed": 47}, 4, 18),
    ({"nbytes": (1 << 20) + 50000, "dict_bytes": 150000, "nchunks": 2, "seed": 48}, 5, 22),   # H68
]
# the same in a partition plan: every shard's instance has the dictionaries attached (..., shard size)
DICT_PLAN_CASES = [
    ({"nbytes": 1 << 20, "dict_bytes": 200000, "nchunks": 2, "seed": 51}, 5, 22, 1 << 16),
    ({"nbytes": 1 << 20, "dict_bytes": 200000, "nchunks": 1, "seed": 52}, 9, 22, 1 << 17),
    ({"nbytes": 1 << 20, "dict_bytes": 100000, "nchunks": 1, "seed": 53}, 3, 18, 1 << 16),
    ({"nbytes": 1 << 20, "dict_bytes": 100000, "nchunks": 3, "seed": 54}, 6, 14, 1 << 15),
]


# BROTLI_PARAM_LGBLOCK (encode.h:190-197): (input, quality, lgwin, lgblock, shard size)
LGBLOCK_CASES = [
    ({"kind": "text", "size": 1 << 20, "seed": 11}, 5, 22, 17, 0),
    ({"kind": "text", "size": 3 << 20, "seed": 7}, 5, 18, 20, 1 << 20),
    ({"kind": "mixed", "size": 2 << 20, "seed": 9}, 5, 22, 12, 1 << 19),      # (clamped to 16)
    ({"kind": "text", "size": 1 << 20, "seed": 11}, 9, 22, 16, 0),
    ({"kind": "text", "size": 1 << 20, "seed": 11}, 6, 20, 21, 1 << 18),
    ({"kind": "file", "name": "alice29.txt"}, 4, 22, 18, 0),
]


def q1_calls(n, feed_kb):
    """[(nbytes, op)]: one FINISH call, or feed_kb KiB per PROCESS call and FINISH with the last
    one — as an extra empty call when n is a multiple of the feed (c/tools/brotli.c:1419-1463)."""
    if not feed_kb:
        return [(n, 2)]
    feed = feed_kb << 10
    calls = [(min(feed, n - o), 0) for o in range(0, n, feed)]
    if n % feed == 0:
        calls.append((0, 2))
    else:
        calls[-1] = (calls[-1][0], 2)
    return calls


def main():
    ref = Ref()
    cases = []
    for spec, q, w, shard in CASES:
        data = G.make(spec)
        out = ref.encode_plan(data, q, w, shard)
        assert ref.decompress(out, len(data)) == data
        cases.append({"input": spec, "quality": q, "lgwin": w,
                      "shard_size": shard, "size": len(out),
                      "sha256": hashlib.sha256(out).hexdigest()})
        print(cases[-1])
    q1 = []
    for spec, w, feed_kb in Q1_CASES:
        data = G.make(spec)
        out = ref.encode_calls(data, 1, w, q1_calls(len(data), feed_kb))
        assert ref.decompress(out, len(data)) == data
        q1.append({"input": spec, "quality": 1, "lgwin": w, "feed_kb": feed_kb, "size": len(out),
                   "sha256": hashlib.sha256(out).hexdigest()})
        print(q1[-1])
    dc = []
    for spec, q, w in DICT_CASES:
        data, chunks = G.dictionary_case(**spec)
        out = ref.encode_calls(data, q, w, [(len(data), 2)], dictionaries=chunks)
        assert ref.decompress_with(out, len(data), chunks) == data
        dc.append({"input": spec, "quality": q, "lgwin": w, "size": len(out),
                   "sha256": hashlib.sha256(out).hexdigest()})
        print(dc[-1])
    for spec, q, w, shard in DICT_PLAN_CASES:
        data, chunks = G.dictionary_case(**spec)
        out = ref.encode_plan(data, q, w, shard, dictionaries=chunks)
        assert ref.decompress_with(out, len(data), chunks) == data
        dc.append({"input": spec, "quality": q, "lgwin": w, "shard_size": shard, "size": len(out),
                   "sha256": hashlib.sha256(out).hexdigest()})
        print(dc[-1])
    lb = []
    for spec, q, w, lg, shard in LGBLOCK_CASES:
        data = G.make(spec)
        out = ref.encode_plan(data, q, w, shard, lgblock=lg)
        assert ref.decompress(out, len(data)) == data
        lb.append({"input": spec, "quality": q, "lgwin": w, "lgblock": lg, "shard_size": shard, "size": len(out),
                   "sha256": hashlib.sha256(out).hexdigest()})
        print(lb[-1])
    json.dump({"generator": "oracle/_ref (google/brotli c/enc, gcc x86-64)",
               "cases": cases, "quality1_cases": q1, "dictionary_cases": dc, "lgblock_cases": lb},
              open(os.path.join(HERE, "golden.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
