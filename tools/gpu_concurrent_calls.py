"""tools/gpu_concurrent_calls.py: T caller threads, each making stock BrotliEncoderCompress(5, LGWIN) calls of MIB MiB of
text on its own encoder instance (the drop-in library lends each instance a device context with its own HIP stream:
encode_abi.c, the context pool).  One call of a few MiB is latency-bound — a tile's serial chain — and leaves most
of the device idle; a server compressing many files at once is what fills it.  Prints one JSON line per T.
  python tools/gpu_concurrent_calls.py [MIB] [LGWIN] [CALLS_PER_THREAD] [T,T,...]"""
import ctypes as C, hashlib, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_inputs as G
from stock_call import bind


def run(mib=4.0, lgwin=22, calls=4, threads=(1, 2, 4, 8, 16, 32), check=True):
    n = int(mib * (1 << 20)) - 16
    L = bind(os.path.join(ROOT, "brotli_amd", "lib", "libbrotlienc_amd.so"))
    tmax = max(threads)
    datas = [bytes(G.enwik_text(n, seed=900 + k)) for k in range(min(tmax, 8))]
    want = None
    if check:
        R = bind(os.path.join(ROOT, "oracle", "_ref", "libbrotli_ref.so"))
        cap = R.BrotliEncoderMaxCompressedSize(n)
        want = []
        for d in datas:
            out = C.create_string_buffer(cap); sz = C.c_size_t(cap)
            assert R.BrotliEncoderCompress(5, lgwin, 0, n, d, C.byref(sz), out)
            want.append(hashlib.sha256(out.raw[:sz.value]).hexdigest())
    cap = L.BrotliEncoderMaxCompressedSize(n)
    res = []
    for T in threads:
        outs = [C.create_string_buffer(cap) for _ in range(T)]
        bad = []
        def work(k, reps):
            for r in range(reps):
                sz = C.c_size_t(cap)
                ok = L.BrotliEncoderCompress(5, lgwin, 0, n, datas[k % len(datas)], C.byref(sz), outs[k])
                if not ok or (want is not None and r == reps - 1 and hashlib.sha256(outs[k].raw[:sz.value]).hexdigest() != want[k % len(datas)]):
                    bad.append(k)
        # warm-up: every thread's context exists and has its workspace
        ws = [threading.Thread(target=work, args=(k, 1)) for k in range(T)]
        [t.start() for t in ws]; [t.join() for t in ws]
        ws = [threading.Thread(target=work, args=(k, calls)) for k in range(T)]
        t0 = time.perf_counter()
        [t.start() for t in ws]; [t.join() for t in ws]
        dt = time.perf_counter() - t0
        rec = {"threads": T, "calls": T * calls, "MiB_per_call": mib, "lgwin": lgwin, "seconds": round(dt, 4),
               "aggregate_MBps": round(T * calls * n / 1e6 / dt, 1), "ms_per_call": round(dt / calls * 1e3, 2),
               "bytes_equal_reference": (not bad) if want is not None else None, "failed": len(bad)}
        print(json.dumps(rec), flush=True)
        res.append(rec)
    return res


if __name__ == "__main__":
    a = sys.argv[1:]
    run(float(a[0]) if a else 4.0, int(a[1]) if len(a) > 1 else 22, int(a[2]) if len(a) > 2 else 4,
        tuple(int(v) for v in a[3].split(",")) if len(a) > 3 else (1, 2, 4, 8, 16, 32))
