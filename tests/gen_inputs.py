"""Seeded synthetic inputs.  Pure numpy, deterministic."""
import numpy as np

SEED = 20260807


def _vocab(rng, n_words):
    """Flat byte pool of n_words pseudo-English words (length 1..14)."""
    lens = rng.integers(1, 15, size=n_words)
    letters = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
    p = np.array([12.7, 9.1, 8.2, 7.5, 7.0, 6.7, 6.3, 6.1, 6.0, 4.3, 4.0, 2.8,
                  2.8, 2.4, 2.4, 2.2, 2.0, 2.0, 1.9, 1.5, 1.0, 0.8, 0.15,
                  0.15, 0.1, 0.07])
    p = p / p.sum()
    pool = rng.choice(letters, size=int(lens.sum()), p=p)
    offs = np.concatenate(([0], np.cumsum(lens)[:-1]))
    return [pool[o:o + l].tobytes() for o, l in zip(offs, lens)]


_TOKEN_CACHE = {}


def _token_table(seed, vocab):
    """Token table: 8 surface forms per word, flattened for vector gather."""
    key = (seed, vocab)
    if key in _TOKEN_CACHE:
        return _TOKEN_CACHE[key]
    rng = np.random.default_rng(seed)
    words = _vocab(rng, vocab)
    forms = []
    for w in words:
        cap = w[:1].upper() + w[1:]
        forms.append((w + b" ", w + b" ", w + b" ", w + b", ", w + b". ",
                      cap + b" ", b"[[" + w + b"]] ", w + b".\n\n"))
    extra = [b"&amp; ", b"== ", b" ==\n", b"''", b"{{", b"}} ", b"|", b"* "]
    extra += [str(i).encode() + b" " for i in range(1000, 2100)]
    toks = [f for fs in forms for f in fs] + extra
    lens = np.array([len(t) for t in toks], dtype=np.int64)
    offs = np.concatenate(([0], np.cumsum(lens)[:-1]))
    pool = np.frombuffer(b"".join(toks), dtype=np.uint8)
    ranks = np.arange(1, vocab + 1, dtype=np.float64)
    pz = ranks ** -1.1
    cdf = np.cumsum(pz / pz.sum())
    _TOKEN_CACHE[key] = (pool, offs, lens, cdf, len(extra))
    return _TOKEN_CACHE[key]


def enwik_text(nbytes, seed=SEED, vocab=200000):
    """enwik-style text: Zipf(1.1) draws from a `vocab`-word vocabulary in 8
    surface forms (plain, comma, full stop, capitalised, [[link]], paragraph)
    plus wiki markup and numbers.  Vectorised: ~1 s per 64 MiB."""
    pool, offs, lens, cdf, n_extra = _token_table(SEED, vocab)
    rng = np.random.default_rng(seed)
    out = np.empty(nbytes, dtype=np.uint8)
    filled = 0
    form_cdf = np.cumsum([0.30, 0.25, 0.22, 0.07, 0.06, 0.06, 0.03, 0.01])
    while filled < nbytes:
        n = min(1 << 22, (nbytes - filled) // 4 + 1024)
        w = np.searchsorted(cdf, rng.random(n))
        w = np.minimum(w, vocab - 1)
        f = np.searchsorted(form_cdf, rng.random(n))
        f = np.minimum(f, 7)
        tok = w * 8 + f
        x = rng.random(n)
        ex = x < 0.03
        tok[ex] = vocab * 8 + (rng.integers(0, n_extra, size=int(ex.sum())))
        tl = lens[tok]
        ends = np.cumsum(tl)
        total = int(ends[-1])
        starts = ends - tl
        idx = np.repeat(offs[tok] - starts, tl) + np.arange(total)
        chunk = pool[idx]
        m = min(total, nbytes - filled)
        out[filled:filled + m] = chunk[:m]
        filled += m
    return out.tobytes()


def random_bytes(nbytes, seed=SEED):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=nbytes, dtype=np.uint8).tobytes()


def mixed_corpus(nbytes, seed=SEED):
    """Silesia-style mix: text, XML-ish, source-ish, rows, floats, gradients,
    sparse zeros, noise; member types repeat with reseeding."""
    rng = np.random.default_rng(seed)
    out = bytearray()
    k = 0
    member = max(1 << 12, min(1 << 20, nbytes // 12 + 1))
    while len(out) < nbytes:
        t = k % 8
        s = seed + 1000 + k
        if t == 0:
            out += enwik_text(member, s)
        elif t == 1:
            txt = enwik_text(member // 2, s).split(b" ")
            out += b"".join(b"<w id=\"%d\">%s</w>\n" % (i % 977, w)
                            for i, w in enumerate(txt))[:member]
        elif t == 2:
            txt = enwik_text(member // 2, s).split(b" ")
            out += b"".join(b"  if (%s != %s) { return %s(%d); }\n" %
                            (txt[i], txt[i + 1], txt[i + 2], i % 31)
                            for i in range(0, len(txt) - 3, 3))[:member]
        elif t == 3:
            a = rng.integers(0, 10 ** 6, size=member // 24)
            out += b"".join(b"%08d|%06d|OK\n" % (i, v)
                            for i, v in enumerate(a))[:member]
        elif t == 4:
            f = np.cumsum(rng.normal(size=member // 4)).astype(np.float32)
            out += f.tobytes()
        elif t == 5:
            g = (np.arange(member) // 7 + rng.integers(0, 3, size=member))
            out += (g & 255).astype(np.uint8).tobytes()
        elif t == 6:
            z = np.zeros(member, dtype=np.uint8)
            pos = rng.integers(0, member, size=member // 50)
            z[pos] = rng.integers(1, 256, size=pos.size)
            out += z.tobytes()
        else:
            out += random_bytes(member // 4, s)
        k += 1
    return bytes(out[:nbytes])


def make(spec):
    """Build an input from a golden-vector spec dict."""
    kind = spec["kind"]
    if kind == "file":
        import os
        here = os.path.dirname(os.path.abspath(__file__))
        return open(os.path.join(here, "golden", spec["name"]), "rb").read()
    if kind == "text":
        return enwik_text(spec["size"], seed=spec.get("seed", SEED),
                          vocab=spec.get("vocab", 20000))
    if kind == "random":
        return random_bytes(spec["size"], seed=spec.get("seed", SEED))
    if kind == "mixed":
        return mixed_corpus(spec["size"], seed=spec.get("seed", SEED))
    if kind == "repeat":
        return (spec["unit"].encode() * (spec["size"] // len(spec["unit"]) + 1))[:spec["size"]]
    raise ValueError(kind)


def dictionary_case(nbytes, dict_bytes, nchunks=1, seed=SEED):
    """Input + raw dictionaries for the attached-dictionary tests: text whose
    vocabulary the dictionary shares, with pieces of the dictionary pasted in every ~2 KB, and the
    dictionary cut into `nchunks` chunks (attached in order)."""
    import random
    rng = random.Random(seed)
    base = bytes(enwik_text(nbytes + dict_bytes, seed=seed, vocab=3000))
    d = base[:dict_bytes]
    body = bytearray(base[dict_bytes:dict_bytes + nbytes])
    for _ in range(nbytes // 2000 + 1):
        ln = rng.randrange(8, 300)
        a = rng.randrange(0, max(1, dict_bytes - ln))
        b = rng.randrange(0, max(1, nbytes - ln))
        piece = d[a:a + ln]
        body[b:b + len(piece)] = piece
    body = bytes(body[:nbytes])
    k = max(1, dict_bytes // nchunks)
    chunks = [d[i * k:(i + 1) * k] for i in range(nchunks - 1)] + [d[(nchunks - 1) * k:]]
    return body, chunks
