"""Host-buffer BrotliEncoderCompress rate (PCIe included) of the drop-in library over the knobs of
brotli_amd_encode_host: batch size and copy threads.  1 GiB text, q5, lgwin 22, 128 KiB shards."""
import ctypes as C, os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen_inputs as G
os.environ["BROTLI_AMD_SHARD_KB"] = "128"
data = G.enwik_text(1 << 30)
L = C.CDLL(os.path.join(ROOT, "brotli_amd", "lib", "libbrotlienc_amd.so"))
L.BrotliEncoderMaxCompressedSize.restype = C.c_size_t
L.BrotliEncoderMaxCompressedSize.argtypes = [C.c_size_t]
L.BrotliEncoderCompress.argtypes = [C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_char_p, C.POINTER(C.c_size_t), C.c_char_p]
cap = L.BrotliEncoderMaxCompressedSize(len(data))
out = C.create_string_buffer(cap)
first = None
for name, env in (("warm", {}), ("one_copy", {"BROTLI_AMD_NO_H2D_OVERLAP": "1"}),
                  ("b1024_t1", {"BROTLI_AMD_H2D_BATCH_MB": "2048", "BROTLI_AMD_H2D_THREADS": "1"}),
                  ("b1024_t2", {"BROTLI_AMD_H2D_BATCH_MB": "2048", "BROTLI_AMD_H2D_THREADS": "2"}),
                  ("b1024_t4", {"BROTLI_AMD_H2D_BATCH_MB": "2048", "BROTLI_AMD_H2D_THREADS": "4"}),
                  ("b1024_t8", {"BROTLI_AMD_H2D_BATCH_MB": "2048", "BROTLI_AMD_H2D_THREADS": "8"}),
                  ("b512_t1", {"BROTLI_AMD_H2D_BATCH_MB": "512", "BROTLI_AMD_H2D_THREADS": "1"}),
                  ("b512_t2", {"BROTLI_AMD_H2D_BATCH_MB": "512", "BROTLI_AMD_H2D_THREADS": "2"}),
                  ("b512_t4", {"BROTLI_AMD_H2D_BATCH_MB": "512", "BROTLI_AMD_H2D_THREADS": "4"}),
                  ("b256_t1", {"BROTLI_AMD_H2D_BATCH_MB": "256", "BROTLI_AMD_H2D_THREADS": "1"}),
                  ("b256_t2", {"BROTLI_AMD_H2D_BATCH_MB": "256", "BROTLI_AMD_H2D_THREADS": "2"}),
                  ("b256_t4", {"BROTLI_AMD_H2D_BATCH_MB": "256", "BROTLI_AMD_H2D_THREADS": "4"}),
                  ("b128_t4", {"BROTLI_AMD_H2D_BATCH_MB": "128", "BROTLI_AMD_H2D_THREADS": "4"})):
    for k in ("BROTLI_AMD_NO_H2D_OVERLAP", "BROTLI_AMD_H2D_BATCH_MB", "BROTLI_AMD_H2D_THREADS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ts = []
    for _ in range(3):
        sz = C.c_size_t(cap)
        t0 = time.perf_counter()
        ok = L.BrotliEncoderCompress(5, 22, 0, len(data), data, C.byref(sz), out)
        ts.append(time.perf_counter() - t0)
        assert ok
    h = hashlib.sha256(out.raw[:sz.value]).hexdigest()
    first = first or h
    print(name, "MB/s %.1f" % (len(data) / 1e6 / min(ts)), ["%.3f" % t for t in ts], sz.value, h == first, flush=True)
