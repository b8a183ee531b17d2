"""GPU measurement: a partition plan with an attached dictionary through the drop-in ABI
(host buffer -> BrotliEncoderCompressStream(FINISH) -> host buffer).  python tools/gpu_dictionary_plan.py [MiB]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen_inputs as G  # noqa: E402
from test_gpu_abi import _bind, drive  # noqa: E402

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
amd = _bind(os.path.join(ROOT, "brotli_amd", "lib", "libbrotlienc_amd.so"))
base = bytes(G.enwik_text((mib << 20) + (1 << 20), seed=4, vocab=20000))
dictionary, data = base[:1 << 20], base[1 << 20:]
for quality, shard_kb in ((5, 128), (5, 512), (9, 512), (3, 128)):
    params = ((1, quality), (2, 22), (5, min(len(data), 1 << 30)), (0x4D490001, shard_kb << 10))
    for with_dict in (True, False):
        best = None
        for rep in range(2):
            t = time.time()
            out, fin = drive(amd, data, [(len(data), 2)], params, out_chunk=len(data) // 2 + (1 << 20),
                             dictionaries=[dictionary] if with_dict else ())
            dt = time.time() - t
            best = dt if best is None else min(best, dt)
        print("q%d %d KiB shards, dictionary %s: %d -> %d bytes, %.3f s, %.0f MB/s end to end" % (
            quality, shard_kb, "1 MiB" if with_dict else "none", len(data), len(out), best, len(data) / 1e6 / best), flush=True)
