"""tools/gpu_stream_kinds.py: one stock BrotliEncoderCompress(5, LGWIN) per member kind of tests/gen_inputs.mixed_corpus
(PROBE_MB MiB of one kind) — which kinds the tiled stream settles slowly on (passes, sweep times: BROTLI_AMD_TILE_LOG)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_inputs as G
from stock_call import bind
N = int(os.environ.get("PROBE_MB", "16")) << 20
LGWIN = int(os.environ.get("PROBE_LGWIN", "22"))
NAMES = ["text", "xml", "source", "rows", "floats", "gradients", "sparse zeros", "noise"]


def one_kind(t, n):
    out = bytearray()
    k = t
    while len(out) < n:
        piece = G.mixed_corpus(12 << 20, seed=G.SEED + 17 * k)
        out += piece[t << 20:(t + 1) << 20] if t < 7 else piece[7 << 20:(7 << 20) + (1 << 18)]
        k += 8
    return bytes(out[:n])


L = bind(os.path.join(ROOT, "brotli_amd", "lib", "libbrotlienc_amd.so"))
only = os.environ.get("PROBE_KINDS")
for t, name in enumerate(NAMES):
    if only and str(t) not in only.split(","):
        continue
    data = one_kind(t, N)
    cap = L.BrotliEncoderMaxCompressedSize(N)
    out = C.create_string_buffer(cap)
    for rep in range(2):
        sz = C.c_size_t(cap)
        t0 = time.perf_counter()
        ok = L.BrotliEncoderCompress(5, LGWIN, 0, N, data, C.byref(sz), out)
        dt = time.perf_counter() - t0
        sys.stderr.flush()
        print("KIND %-13s call %d: ok=%d %.3f s = %.1f MB/s, %d bytes out" % (name, rep, ok, dt, N / 1e6 / dt, sz.value), flush=True)
