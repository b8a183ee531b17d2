"""One stock BrotliEncoderCompress(quality, lgwin) of the drop-in library on `mib` MiB of synthetic data, in THIS
process (the caller isolates it: a GPU memory fault is a SIGABRT from the HSA runtime).  Prints one JSON line per
stage so that a cut-off run still says how far it came.

  python tools/stock_call.py MIB [LGWIN] [KIND] [CALLS] [--ref]      KIND: text | mix | noise
"""
import ctypes as C
import hashlib
import json
import os
import resource
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def bind(path):
    L = C.CDLL(path)
    L.BrotliEncoderMaxCompressedSize.restype = C.c_size_t
    L.BrotliEncoderMaxCompressedSize.argtypes = [C.c_size_t]
    L.BrotliEncoderCompress.restype = C.c_int
    L.BrotliEncoderCompress.argtypes = [C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_char_p, C.POINTER(C.c_size_t), C.c_char_p]
    return L


def main():
    resource.setrlimit(resource.RLIMIT_CORE, (0, 0))
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    mib = float(args[0])
    lgwin = int(args[1]) if len(args) > 1 else 22
    kind = args[2] if len(args) > 2 else "text"
    calls = int(args[3]) if len(args) > 3 else 2
    quality = int(os.environ.get("STOCK_QUALITY", "5"))
    import gen_inputs as G
    import numpy as np
    n = int(mib * (1 << 20))
    t0 = time.time()
    if kind == "text":
        data = bytes(G.enwik_text(n))
    elif kind == "mix":
        data = bytes(G.mixed_corpus(n, seed=3))
    else:
        data = bytes(np.random.default_rng(5).integers(0, 256, n, dtype=np.uint8))
    print(json.dumps({"stage": "input", "bytes": n, "kind": kind, "s": round(time.time() - t0, 2)}), flush=True)
    L = bind(os.path.join(ROOT, "brotli_amd", "lib", "libbrotlienc_amd.so"))
    cap = L.BrotliEncoderMaxCompressedSize(n)
    out = C.create_string_buffer(cap)
    sha = None
    times = []
    for k in range(calls):
        sz = C.c_size_t(cap)
        t0 = time.perf_counter()
        ok = L.BrotliEncoderCompress(quality, lgwin, 0, n, data, C.byref(sz), out)
        dt = time.perf_counter() - t0
        times.append(dt)
        if not ok:
            print(json.dumps({"stage": "call", "k": k, "error": "BROTLI_FALSE"}), flush=True)
            sys.exit(3)
        h = hashlib.sha256(out.raw[:sz.value]).hexdigest()
        print(json.dumps({"stage": "call", "k": k, "s": round(dt, 3), "MBps": round(n / 1e6 / dt, 1), "out_bytes": sz.value,
                          "sha256": h[:16]}), flush=True)
        if sha is not None and h != sha:
            print(json.dumps({"stage": "call", "error": "not deterministic"}), flush=True)
            sys.exit(4)
        sha = h
    res = {"stage": "done", "MiB": mib, "lgwin": lgwin, "kind": kind, "quality": quality, "out_bytes": sz.value,
           "MBps_best": round(n / 1e6 / min(times), 1), "seconds_all": [round(t, 3) for t in times]}
    if "--ref" in sys.argv:
        R = bind(os.path.join(ROOT, "oracle", "_ref", "libbrotli_ref.so"))
        out2 = C.create_string_buffer(cap)
        sz2 = C.c_size_t(cap)
        t0 = time.perf_counter()
        assert R.BrotliEncoderCompress(quality, lgwin, 0, n, data, C.byref(sz2), out2)
        res["reference_1core_MBps"] = round(n / 1e6 / (time.perf_counter() - t0), 1)
        res["bytes_equal_reference"] = sz2.value == sz.value and hashlib.sha256(out2.raw[:sz2.value]).hexdigest() == sha
    print(json.dumps(res), flush=True)
    if res.get("bytes_equal_reference") is False:
        sys.exit(5)


if __name__ == "__main__":
    main()
