"""The stock call on a big buffer, on the GPU: BrotliEncoderCompress(5, lgwin, ...) of libbrotlienc_amd.so — no partition
plan, no vendor parameter — on inputs longer than the window, next to the reference library's BrotliEncoderCompress
(oracle/_ref/libbrotli_ref.so, prebuilt; nothing here reads /root/reference).  Such a stream takes the tiled stream path
(JOB_FLAG_STREAMT: k_tile.h stream_*, index chunks with a look-back, meta-block cuts, the ring's physical end, the
16-bit store counter) or, where it does not suit the tiles, the serial device stream; the bytes must be the
reference's either way.  ctypes only (no torch): the boundary is the C ABI."""
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np
import pytest

import gen_inputs as G
from test_gpu_abi import LIBDIR, ROOT, _bind

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1500, method="thread")]
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def amd():
    return _bind(os.path.join(LIBDIR, "libbrotlienc_amd.so"))


@pytest.fixture(scope="module")
def stock(ref):
    return _bind(os.path.join(ROOT, "oracle", "_ref", "libbrotli_ref.so"))


def one_shot(L, data, lgwin, quality=5):
    cap = L.BrotliEncoderMaxCompressedSize(len(data))
    out = C.create_string_buffer(cap)
    n = C.c_size_t(cap)
    t = time.time()
    assert L.BrotliEncoderCompress(quality, lgwin, 0, len(data), data, C.byref(n), out)
    return out.raw[:n.value], time.time() - t


@pytest.mark.parametrize("mib,lgwin,seed", [(48, 22, 5), (9, 18, 6), (3, 17, 7), (20, 20, 8)])
def test_text_longer_than_the_window(amd, stock, mib, lgwin, seed):
    """6 / 18 / 12 / 10 laps of the ring buffer, a dozen and more meta-blocks, keys stored more than 65536 times."""
    data = bytes(G.enwik_text((mib << 20) + 12345, seed=seed))
    got, _ = one_shot(amd, data, lgwin)
    want, _ = one_shot(stock, data, lgwin)
    assert got == want


@pytest.mark.parametrize("nbytes,lgwin,seed", [((6 << 20) + 777, 23, 31), ((12 << 20) + 5, 24, 32), (1 << 24, 24, 33),
                                               ((1 << 24) - 16, 24, 34), ((40 << 20) + 99, 23, 35), ((5 << 20), 24, 36),
                                               ((16 << 20) + 4321, 23, 37), ((1 << 24) + 1, 24, 38),
                                               ((1 << 24) + 65536 + 5, 24, 39), ((3 << 23) + 17, 24, 40)])
def test_the_windows_the_cli_chooses(amd, stock, nbytes, lgwin, seed):
    """lgwin 23 / 24 — what the reference's CLI picks by itself for a file above 4 MiB / 8 MiB (c/tools/brotli.c:1434-1447).
    An index chunk with its look-back is two windows of 24-bit positions: every length at lgwin 23 (2.4 and 5 laps of
    the 16 MiB ring here), and at lgwin 24 the streams that fit one chunk — a file up to 16 MiB, i.e. every file for which
    the CLI's choice of 24 still covers the whole file.  Longer streams at lgwin 24 stay on the serial device stream."""
    data = bytes(G.enwik_text(nbytes, seed=seed))
    got, dt = one_shot(amd, data, lgwin)
    want, _ = one_shot(stock, data, lgwin)
    assert got == want
    got, dt = one_shot(amd, data, lgwin)
    print("lgwin %d, %.1f MiB: %.3f s = %.0f MB/s" % (lgwin, nbytes / 1048576.0, dt, nbytes / 1e6 / dt))
    assert dt < 1.5           # (the serial device stream: ~2 MB/s)


@pytest.mark.parametrize("mib,piece_kb,seed,tail", [(40, 1024, 11, 4321), (24, 64, 12, 4321), (64, 1024, 13, 0), (40, 64, 14, 0), (88, 1024, 15, 0)])
def test_process_fed_stream_without_a_size_hint(amd, stock, mib, piece_kb, seed, tail):
    """What Compressor.process of the Python module (or the CLI on a pipe) does: PROCESS calls of 1 MiB / 64 KiB, no
    BROTLI_PARAM_SIZE_HINT, then FINISH.  The library holds the pieces and parses the whole stream in tiles at the FINISH;
    the size hint is the one the reference latches at its first full block (1 MiB: H68; 64 KiB: H58 for the whole
    stream), so the bytes are the stock library's driven the same way."""
    from test_gpu_abi import drive
    # (tail 0: the PROCESS calls end on a block boundary and the FINISH comes empty — the reference has encoded the
    #  last block with is_last = 0 by then: host_plan.h stream_tail_fix)
    data = bytes(G.enwik_text((mib << 20) + tail, seed=seed))
    piece = piece_kb << 10
    ops = [(piece, 0)] * (len(data) // piece) + [(len(data) % piece, 2)]
    t = time.time()
    got, fin = drive(amd, data, ops, out_chunk=1 << 22)
    dt = time.time() - t
    want, fin2 = drive(stock, data, ops, out_chunk=1 << 22)
    assert fin and fin2 and bytes(got) == bytes(want)
    print("process-fed %d MiB in %d KiB pieces: %.2f s = %.0f MB/s" % (mib, piece_kb, dt, len(data) / 1e6 / dt))
    assert dt < 30.0          # (the serial device stream needs minutes for this)


def test_english_keeps_the_dictionary_gate_open(amd, stock):
    """alice29.txt over and over with synthetic text in between, 12 MiB (half of the positions end up unstored: the copies
    are tens of kilobytes long, several sweeps): the static dictionary's gate stays open behind
    the first block, the tiles start over with it taken as open for good (k_tile.h: TILE_GATE_OPEN)."""
    alice = open(os.path.join(ROOT, "tests", "golden", "alice29.txt"), "rb").read()
    parts = []
    for k in range(120):
        parts.append(alice[(k * 7919) % 50000:])
        parts.append(bytes(G.enwik_text(60000, seed=100 + k)))
    data = b"".join(parts)[:12 << 20]
    got, _ = one_shot(amd, data, 22)
    want, _ = one_shot(stock, data, 22)
    assert got == want


def test_duplicated_pieces_copies_longer_than_a_block(amd, stock):
    """An archive with files in it twice: 1.5 MiB pieces that come again — ExtendLastCommand consumes whole input
    blocks, dozens in a row (k_tile.h: tiles without a command and without pending literals)."""
    text = bytes(G.enwik_text(9 << 20, seed=21))
    piece = text[1 << 20:(5 << 19)]
    data = text[:6 << 20] + piece + text[6 << 20:7 << 20] + piece + piece[:900000] + text[7 << 20:]
    for lgwin in (22, 20):
        got, _ = one_shot(amd, data, lgwin)
        want, _ = one_shot(stock, data, lgwin)
        assert got == want


def test_plain_chain_copy_to_the_block_end(amd, stock):
    """tools/fuzz_stream_sim.py seed 15 (a copy of the chain's fast path that runs to its block's end: the three
    positions the next block's stitch stores were marked unstored) through the paths a stream of that size takes:
    inside the window (tiled shard), longer than it (tiled stream), and as a partition plan of 128 KiB shards."""
    import fuzz_stream_sim
    from test_gpu_abi import drive
    data, lgwin, kind = fuzz_stream_sim.make(15)
    for lw in (22, 19):
        got, _ = one_shot(amd, data, lw)
        want, _ = one_shot(stock, data, lw)
        assert got == want
    got, fin = drive(amd, data, [(len(data), 2)], params=((0x4D490001, 128 << 10),))
    assert fin
    from refharness import Ref
    assert bytes(got) == Ref().encode_plan(data, 5, 22, 128 << 10)


def test_mixed_data_and_raw_meta_blocks(amd, stock):
    """The mixed corpus (floats, sparse zeros, noise, text) and text with 2 MiB of random bytes in it (meta-blocks
    stored raw, the distance cache rolled back behind them: k_stream_scan / k_stream_rollback) stay on the tiled path;
    random bytes alone leave it for the serial stream.  The bytes are the reference's every time."""
    rng = np.random.default_rng(4)
    text = bytes(G.enwik_text(6 << 20, seed=9))
    cases = [(bytes(G.mixed_corpus(6 << 20, seed=3)), 22),
             (text[:4 << 20] + bytes(rng.integers(0, 256, 2 << 20, dtype=np.uint8)) + text[4 << 20:], 18),
             (bytes(rng.integers(0, 256, 3 << 20, dtype=np.uint8)), 20)]
    for data, lgwin in cases:
        got, _ = one_shot(amd, data, lgwin)
        want, _ = one_shot(stock, data, lgwin)
        assert got == want


def test_reference_cli_on_a_big_file(tmp_path):
    """`brotli -q 5 -w 22 file` — the reference's CLI, unmodified, linked against our library — on a 20 MiB file: the
    CLI announces the file's size (BROTLI_PARAM_SIZE_HINT) and feeds it piece by piece; the library holds the pieces
    until FINISH and the stream takes the tiled path.  Same bytes as the CLI linked against the reference library."""
    import subprocess
    cli, cli_ref = (os.path.join(ROOT, "oracle", "_ref", n) for n in ("brotli_cli_amd", "brotli_cli_ref"))
    if not (os.path.exists(cli) and os.path.exists(cli_ref)):
        pytest.skip("oracle/_ref/brotli_cli_amd / brotli_cli_ref not built")
    dropin = os.path.join(LIBDIR, "dropin")
    os.makedirs(dropin, exist_ok=True)
    for name, target in (("libbrotlienc.so.1", "../libbrotlienc_amd.so"), ("libbrotli_amd_hip.so", "../libbrotli_amd_hip.so")):
        p = os.path.join(dropin, name)
        if not os.path.lexists(p):
            os.symlink(target, p)
    env = dict(os.environ, LD_LIBRARY_PATH=dropin + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    src = tmp_path / "big.txt"
    src.write_bytes(bytes(G.enwik_text(20 << 20, seed=77)))
    got = subprocess.run([cli, "-q", "5", "-w", "22", "-c", str(src)], capture_output=True, env=env, check=True).stdout
    want = subprocess.run([cli_ref, "-q", "5", "-w", "22", "-c", str(src)], capture_output=True, check=True).stdout
    assert got == want
    # no -w: the CLI chooses lgwin 24 for a file of 8 ... 16 MiB (and 23 for 4 ... 8 MiB), which still covers the file
    for n in (13 << 20, (7 << 20) + 31):
        src.write_bytes(bytes(G.enwik_text(n, seed=78)))
        t = time.time()
        got = subprocess.run([cli, "-q", "5", "-c", str(src)], capture_output=True, env=env, check=True).stdout
        dt = time.time() - t
        want = subprocess.run([cli_ref, "-q", "5", "-c", str(src)], capture_output=True, check=True).stdout
        assert got == want
        assert dt < 20.0      # (process start + context creation included; the serial stream needs ~2 MB/s)


def run_isolated(args, timeout_s, env=None):
    """tools/stock_call.py in a child process with a time limit: whatever a kernel does at that size (a GPU memory
    fault ends the process with SIGABRT), this suite goes on and the test that asked reports it."""
    import subprocess
    e = dict(os.environ)
    e.pop("BROTLI_AMD_SHARD_KB", None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stock_call.py")] + [str(a) for a in args],
                       capture_output=True, text=True, timeout=timeout_s, env=e)
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r.returncode, lines, r.stderr[-2000:]


@pytest.mark.parametrize("mib", [96, 320])
def test_stock_call_between_the_sizes(mib):
    """Sizes between the 48 MiB case above and the 1 GiB call (four chain groups per wave from 64 MiB on, hundreds of index
    chunks, dozens of meta-blocks), each in a process of its own, next to the reference library."""
    rc, lines, err = run_isolated([mib, 22, "text", 1, "--ref"], 600)
    assert rc == 0 and lines and lines[-1].get("bytes_equal_reference") is True, (rc, lines[-2:], err)


@pytest.mark.parametrize("mib,lgwin,kind,env", [(40, 24, "text", None), (100, 24, "text", None), (40, 24, "mix", None),
                                                 (48, 22, "text", {"BROTLI_AMD_HALF_CHUNKS": "1"}),
                                                 (20, 19, "mix", {"BROTLI_AMD_HALF_CHUNKS": "1"})])
def test_lgwin_24_longer_than_the_window(mib, lgwin, kind, env):
    """What the CLI does to every file above 16 MiB: lgwin 24 and a stream longer than the window.  A chunk with its
    look-back has to stay within 24-bit positions, so the chunks are HALF a window there (host_plan.h plan_stream): a
    search with fewer than 16 same-key entries before it in its chunk is the chain's, which goes on in the chunk before
    (k_index.h IxGeom::older, k_chain.h c_search_exact, k_tile.h stream_events).  BROTLI_AMD_HALF_CHUNKS=1 runs the
    smaller windows the same way.  Each next to the reference library, in a process of its own."""
    rc, lines, err = run_isolated([mib, lgwin, kind, 2, "--ref"], 600, env=env)
    rec = lines[-1] if lines else {}
    print(json.dumps(rec))
    assert rc == 0 and rec.get("bytes_equal_reference") is True, (rc, lines[-2:], err)
    if kind == "text":
        assert rec["MBps_best"] > 100.0      # (the serial device stream: ~2 MB/s; the mix settles in ~17 passes at any window: ~8 MB/s)


def test_1gib_cli_default_window():
    """`brotli -q 5 file` of a 1 GiB file = BrotliEncoderCompress(5, 24): bytes of the reference (sha256)."""
    rc, lines, err = run_isolated([1024, 24, "text", 2, "--ref"], 900)
    rec = lines[-1] if lines else {}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "stock_call_1GiB_lgwin24.json"), "w") as f:
        json.dump({"rc": rc, "lines": lines, "stderr_tail": err[-500:]}, f)
    print(json.dumps(rec))
    assert rc == 0 and rec.get("bytes_equal_reference") is True, (rc, lines[-2:], err)


def test_1gib_stock_call():
    """The metric's own call: BrotliEncoderCompress(5, 22) of 1 GiB with no vendor setting, byte-identical to the
    reference's (sha256), timed from the host buffer to the host buffer (PCIe both ways included); the second call
    has the context's allocations behind it (bench.py reports the same call as config.stock_call_no_plan.whole_input).
    In a child process: see run_isolated."""
    rc, lines, err = run_isolated([1024, 22, "text", 2, "--ref"], 900)
    rec = lines[-1] if lines else {}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "stock_call_1GiB.json"), "w") as f:
        json.dump({"rc": rc, "lines": lines, "stderr_tail": err[-500:]}, f)
    print(json.dumps(rec))
    assert rc == 0 and rec.get("bytes_equal_reference") is True, (rc, lines[-2:], err)
