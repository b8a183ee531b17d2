"""GPU measurement: k_decode variants (waves per SIMD 4 / 8, with / without the LDS copies of the
per-symbol tables) on the pieces of a plan.  python tools/gpu_decode_variants.py [MiB] [shard KiB]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import gen_inputs as G  # noqa: E402
from brotli_amd import hip  # noqa: E402

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 512
shard = (int(sys.argv[2]) if len(sys.argv) > 2 else 64) << 10
n = mib << 20
t0 = time.time()
data = G.enwik_text(n, seed=5)
d_in = hip.to_device(data, 0)
ctx = hip.Context(0)
for quality in (5,):
    params = hip.make_params(quality, 22, shard, n)
    d_out = torch.empty(ctx.max_output(n, params), dtype=torch.uint8, device="cuda:0")
    d_sizes = torch.zeros(n // shard, dtype=torch.int64, device="cuda:0")
    nbytes, info = ctx.encode_device(d_in, n, params, d_out, d_sizes)
    pieces = hip.plan_pieces(d_sizes.cpu().tolist(), n, shard, 22)
    d_back = torch.zeros(n + 64, dtype=torch.uint8, device="cuda:0")
    print("input %d MiB, %d pieces of %d KiB, %d compressed bytes, setup %.1f s" % (mib, len(pieces), shard >> 10, nbytes, time.time() - t0), flush=True)
    for variant in (4, 8, 4 + 16, 8 + 16, 4, 8):
        os.environ["BROTLI_AMD_DECODE_VARIANT"] = str(variant)
        hip.refresh_env()
        ms_all = []
        for rep in range(3):
            res, ms = ctx.decode_device(d_out, nbytes, d_back, n, pieces)
            ms_all.append(round(ms, 2))
        ok = bool(torch.equal(d_back[:n], d_in[:n]))
        print("q%d waves/SIMD %d lds_cache %d: ms %s  %.1f GB/s  equal %s" % (
            quality, variant & 15, 0 if variant & 16 else 1, ms_all, n / 1e6 / min(ms_all), ok), flush=True)
