"""tools/gpu_fuzz_windows.py FIRST COUNT: stock BrotliEncoderCompress(5, lgwin) of the drop-in library next to the reference
library on streams built to stress what round 6 added to the tiled stream — lgwin 23 / 24, chunks of half a window
(k_index.h IxGeom::older: candidates below the chunk's base, i.e. matches 8 ... 16 MiB back through keys with fewer than 16
occurrences in between), lengths around the chunk boundaries:
  kind 0  text
  kind 1  text repeated with a period near the window / half the window / the ring, mutated a little
  kind 2  a constant background with rare tokens that come again 9 ... 15 MiB later (rare keys: the far candidates are in the ring)
  kind 3  text with rare binary tokens repeated far back
  kind 4  tiny vocabulary (keys stored more than 65536 times: the store counter's zones over half-window chunks)
FUZZ_LGWINS=20,21,22: the same at smaller windows (one window of look-back per chunk).
Prints one line per seed; exit code 1 on a mismatch."""
import ctypes as C, hashlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_inputs as G
from stock_call import bind


def make(seed):
    rng = np.random.default_rng(seed)
    lgwin = int(rng.choice([int(v) for v in os.environ["FUZZ_LGWINS"].split(",")] if os.environ.get("FUZZ_LGWINS") else [24, 24, 24, 23]))
    W = 1 << lgwin
    kind = seed % 5
    n = int(rng.integers(W + 70000, 3 * W + 200000)) if kind != 4 else int(rng.integers(W + 70000, 2 * W))
    if rng.integers(0, 4) == 0:
        n = int(rng.choice([W + 1, W + 65536 + 3, 2 * W, 2 * W + 1, (3 * W) // 2 + 17, W + (W >> 1)]))
    if kind == 0:
        data = bytes(G.enwik_text(n, seed=seed))
    elif kind == 1:
        period = int(rng.choice([W, W - 16, W + 7, W >> 1, (W >> 1) + 11, 2 * W - 3, (W >> 1) - 16, 3 * (W >> 2)]))
        base = np.frombuffer(bytes(G.enwik_text(min(period, n), seed=seed)), dtype=np.uint8)
        buf = np.tile(base, -(-n // len(base)))[:n].copy()
        nm = int(rng.integers(0, n // 4000 + 2))
        buf[rng.integers(0, n, nm)] = rng.integers(32, 127, nm)
        data = buf.tobytes()
    elif kind in (2, 3):
        if kind == 2:
            buf = np.full(n, int(rng.integers(0, 256)), dtype=np.uint8)
        else:
            buf = np.frombuffer(bytes(G.enwik_text(n, seed=seed)), dtype=np.uint8).copy()
        ntok = int(rng.integers(200, 3000))
        toks = [rng.integers(0, 256, int(rng.integers(6, 40)), dtype=np.uint8) for _ in range(ntok)]
        back = int(rng.integers(9 << 20, 15 << 20)) if lgwin == 24 else int(rng.integers(5 << 20, 7 << 20)) if lgwin == 23 else int(rng.integers((W >> 1) + (W >> 3), W - (W >> 4)))
        first = rng.integers(0, max(1, n - back - 64), ntok)
        for t, p in zip(toks, first):
            p = int(p)
            while p + len(t) < n:
                buf[p:p + len(t)] = t
                p += back + int(rng.integers(-3, 4)) * int(rng.integers(0, 2))
        data = buf.tobytes()
    else:
        words = [bytes(rng.integers(97, 123, int(rng.integers(4, 9)), dtype=np.uint8)) + b" " for _ in range(int(rng.integers(2, 5)))]
        idx = rng.integers(0, len(words), n // 4)
        data = (bytes(rng.integers(97, 123, 30000, dtype=np.uint8)) + b"".join(words[i] for i in idx))[:n]
    return data, lgwin, kind


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    L = bind(os.path.join(ROOT, "brotli_amd", "lib", "libbrotlienc_amd.so"))
    R = bind(os.path.join(ROOT, "oracle", "_ref", "libbrotli_ref.so"))
    bad = 0
    for seed in range(first, first + count):
        data, lgwin, kind = make(seed)
        n = len(data)
        cap = L.BrotliEncoderMaxCompressedSize(n)
        out = C.create_string_buffer(cap); out2 = C.create_string_buffer(cap)
        sz = C.c_size_t(cap); sz2 = C.c_size_t(cap)
        t0 = time.perf_counter()
        ok = L.BrotliEncoderCompress(5, lgwin, 0, n, data, C.byref(sz), out)
        dt = time.perf_counter() - t0
        assert R.BrotliEncoderCompress(5, lgwin, 0, n, data, C.byref(sz2), out2)
        same = bool(ok) and sz.value == sz2.value and hashlib.sha256(out.raw[:sz.value]).digest() == hashlib.sha256(out2.raw[:sz2.value]).digest()
        bad += 0 if same else 1
        print("seed %d kind %d lgwin %d n %d: %s (%d -> %d bytes, %.2f s = %.0f MB/s)" % (
            seed, kind, lgwin, n, "ok" if same else "MISMATCH", n, sz.value, dt, n / 1e6 / dt), flush=True)
    print("mismatches: %d" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
