"""Host-buffer BrotliEncoderCompress rate (PCIe included) of the drop-in library: 1 GiB text, q5,
lgwin 22, 128 KiB shards.  usage: gpu_e2e_sweep.py [plain|T]  (plain = runtime's pageable copies,
T = number of pinned copy lanes)."""
import ctypes as C, os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen_inputs as G
mode = sys.argv[1] if len(sys.argv) > 1 else "4"
os.environ["BROTLI_AMD_SHARD_KB"] = "128"
if mode == "plain":
    os.environ["BROTLI_AMD_PLAIN_COPY"] = "1"
else:
    os.environ["BROTLI_AMD_COPY_THREADS"] = mode
data = G.enwik_text(int(os.environ.get("E2E_MB", "1024")) << 20)
L = C.CDLL(os.path.join(ROOT, "brotli_amd", "lib", "libbrotlienc_amd.so"))
L.BrotliEncoderMaxCompressedSize.restype = C.c_size_t
L.BrotliEncoderMaxCompressedSize.argtypes = [C.c_size_t]
L.BrotliEncoderCompress.argtypes = [C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_char_p, C.POINTER(C.c_size_t), C.c_char_p]
cap = L.BrotliEncoderMaxCompressedSize(len(data))
out = C.create_string_buffer(cap)
ts = []
for _ in range(5):
    sz = C.c_size_t(cap)
    t0 = time.perf_counter()
    ok = L.BrotliEncoderCompress(5, 22, 0, len(data), data, C.byref(sz), out)
    ts.append(time.perf_counter() - t0)
    assert ok
print(mode, "MB/s median %.1f best %.1f" % (len(data) / 1e6 / sorted(ts)[2], len(data) / 1e6 / min(ts)),
      ["%.3f" % t for t in ts], sz.value, hashlib.sha256(out.raw[:sz.value]).hexdigest()[:16], flush=True)
