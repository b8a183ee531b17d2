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
]


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
    json.dump({"generator": "oracle/_ref (google/brotli c/enc, gcc x86-64)",
               "cases": cases}, open(os.path.join(HERE, "golden.json"), "w"),
              indent=1)


if __name__ == "__main__":
    main()
