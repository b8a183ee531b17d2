"""ctypes binding of the HIP C-ABI layer (include/brotli_amd_hip.h).

Device buffers are torch tensors (torch is only the allocator here); every
call goes through the C ABI of brotli_amd/lib/libbrotli_amd_hip.so.  There is
no CPU fallback: if the library is missing or no gfx950 device is present the
constructors raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BROTLI_AMD_HIP_LIB") or os.path.join(_HERE, "lib", "libbrotli_amd_hip.so")
TABLES_PATH = os.path.join(_HERE, "data", "brotli_tables.bin")
INPUT_SLACK = 64

OK, ERROR, UNSUPPORTED, OVERFLOW, DEVICE_FAULT = 0, -1, -2, -3, -4
FLAG_NO_PAIR, FLAG_NO_QUAD, FLAG_FORCE_SLOW = 1, 2, 4


class JobParams(C.Structure):
    _fields_ = [("quality", C.c_int32), ("lgwin", C.c_int32),
                ("size_hint", C.c_uint32), ("flags", C.c_uint32),
                ("shard_size", C.c_uint64), ("stream_base", C.c_uint64),
                ("is_last", C.c_int32), ("reserved", C.c_int32)]


class JobInfo(C.Structure):
    _fields_ = [("nshards", C.c_uint64), ("out_bytes", C.c_uint64),
                ("ws_bytes", C.c_uint64), ("rounds", C.c_uint32),
                ("reserved", C.c_uint32), ("ms_total", C.c_float),
                ("ms_init", C.c_float), ("ms_parse", C.c_float),
                ("ms_build", C.c_float), ("ms_store", C.c_float),
                ("ms_gather", C.c_float), ("ms_index", C.c_float),
                ("ms_ix_bucket", C.c_float), ("searches", C.c_uint64),
                ("search_steps", C.c_uint64), ("commands", C.c_uint64),
                ("exact_searches", C.c_uint64),
                ("prof", C.c_uint64 * 12)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k not in ("reserved", "prof")}
        d["prof"] = list(self.prof)
        # tiled jobs (k_tile.h): sweeps of the chain, shards that left the tiled path for the plain chain
        d["tile_sweeps"] = self.reserved & 0xFF
        d["tile_fallback_shards"] = self.reserved >> 8
        return d


class FastParams(C.Structure):
    _fields_ = [("lgwin", C.c_int32), ("carry_bits", C.c_uint32),
                ("carry_value", C.c_uint32), ("is_last", C.c_int32)]


def fast_stream_header(lgwin):
    """(bits, value) of the stream header at quality 1: EncodeWindowBits of
    max(lgwin, 18) (c/enc/encode.c:191-211, 670-674)."""
    return 4, ((max(lgwin, 18) - 17) << 1) | 1


CMD_DTYPE = np.dtype([("insert_len", "<u4"), ("copy_len", "<u4"),
                      ("dist_extra", "<u4"), ("cmd_prefix", "<u2"),
                      ("dist_prefix", "<u2")])


class DecodePiece(C.Structure):
    _fields_ = [("in_off", C.c_uint64), ("in_len", C.c_uint64), ("out_off", C.c_uint64),
                ("out_cap", C.c_uint64), ("flags", C.c_uint32), ("lgwin", C.c_uint32)]


class DecodeResult(C.Structure):
    _fields_ = [("out_bytes", C.c_uint64), ("in_bits", C.c_uint64), ("error", C.c_uint32),
                ("finished", C.c_uint32), ("lgwin", C.c_uint32), ("metablocks", C.c_uint32)]


PIECE_HEADER, PIECE_ISOLATED = 1, 2
DECODE_SLACK = 4096


def plan_pieces(shard_sizes, total, shard_size, lgwin):
    """The pieces of a partition plan's output: compressed sizes per shard (as
    brotli_amd_encode_device reports them), decoded size of the stream, bytes per shard."""
    pieces, off = [], 0
    for k, n in enumerate(shard_sizes):
        out_off = k * shard_size
        pieces.append((off, int(n), out_off, min(shard_size, total - out_off),
                       PIECE_HEADER if k == 0 else PIECE_ISOLATED, lgwin))
        off += int(n)
    return pieces


class BrotliAmdError(RuntimeError):
    pass


def refresh_env():
    """Has the library read its BROTLI_AMD_* experiment knobs again (it reads them when a context is created): for tools
    and tests that change one between two jobs of a living context."""
    L = load_library()
    if hasattr(L, "brotli_amd_refresh_env"):
        L.brotli_amd_refresh_env()


def load_library(path=LIB_PATH):
    if not os.path.exists(path):
        raise BrotliAmdError(
            "HIP library %s is missing: run `python -c 'import __graft_entry__ "
            "as g; g.build()'` (no CPU fallback exists)" % path)
    # torch first: its wheel carries a HIP / HSA runtime of its own under the same SONAMEs.  Loaded first, this library
    # binds to that copy and the process has ONE runtime; the other way round there are two, and the second one to
    # initialise finds no device ("No HIP GPUs are available" from torch: tools/gpu_kinds_groups.py met it).  The device
    # tensors this module hands the library are torch's, so torch is needed anyway.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    L.brotli_amd_ctx_create.argtypes = [C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]
    L.brotli_amd_ctx_destroy.argtypes = [C.c_void_p]
    try:
        L.brotli_amd_refresh_env.argtypes = []
        L.brotli_amd_refresh_env.restype = None
    except AttributeError:      # (an experiment build of an older tree, BROTLI_AMD_HIP_LIB)
        pass
    L.brotli_amd_last_error.argtypes = [C.c_void_p]
    L.brotli_amd_last_error.restype = C.c_char_p
    L.brotli_amd_max_output.argtypes = [C.c_uint64, C.POINTER(JobParams)]
    L.brotli_amd_max_output.restype = C.c_uint64
    L.brotli_amd_encode_device.argtypes = [
        C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(JobParams), C.c_void_p,
        C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(JobInfo)]
    L.brotli_amd_encode_host.argtypes = [
        C.c_void_p, C.c_char_p, C.c_uint64, C.POINTER(JobParams), C.c_void_p,
        C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(JobInfo)]
    L.brotli_amd_fast_max_output.argtypes = [C.c_uint64, C.c_uint64, C.c_int]
    L.brotli_amd_fast_max_output.restype = C.c_uint64
    L.brotli_amd_encode_fast_device.argtypes = [
        C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_uint64,
        C.POINTER(FastParams), C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(JobInfo)]
    L.brotli_amd_encode_fast_host.argtypes = [
        C.c_void_p, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_uint64,
        C.POINTER(FastParams), C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(JobInfo)]
    L.brotli_amd_decode_device.argtypes = [
        C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(DecodePiece), C.c_uint64, C.c_void_p,
        C.c_uint64, C.POINTER(DecodeResult), C.POINTER(C.c_float)]
    L.brotli_amd_decode_host.argtypes = [
        C.c_void_p, C.c_char_p, C.c_uint64, C.POINTER(DecodePiece), C.c_uint64, C.c_void_p,
        C.c_uint64, C.POINTER(DecodeResult), C.POINTER(C.c_float)]
    L.brotli_amd_debug_parse.argtypes = [
        C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(JobParams), C.c_void_p,
        C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(JobInfo)]
    return L


FLAG_STREAM_TILES = 64   # BROTLI_AMD_FLAG_STREAM_TILES (include/brotli_amd_hip.h): one whole quality-5 stream, longer than
                         # the window, parsed in tiles; the call fails with BROTLI_AMD_SERIAL (-5) when the data does not suit them


def make_params(quality=5, lgwin=22, shard_size=0, size_hint=0, stream_base=0,
                is_last=True, flags=0):
    return JobParams(quality, lgwin, size_hint, flags, shard_size, stream_base,
                     1 if is_last else 0, 0)


class Context:
    """One HIP context (device + stream + cached workspace)."""

    def __init__(self, device=0, lib_path=LIB_PATH, tables_path=TABLES_PATH):
        self.L = load_library(lib_path)
        h = C.c_void_p()
        rc = self.L.brotli_amd_ctx_create(device, tables_path.encode(), C.byref(h))
        self.h = h
        if rc != OK:
            msg = self.L.brotli_amd_last_error(h).decode() if h else "?"
            if h:
                self.L.brotli_amd_ctx_destroy(h)
            self.h = None
            raise BrotliAmdError("brotli_amd_ctx_create failed: " + msg)
        self.device = device

    def close(self):
        if self.h:
            self.L.brotli_amd_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != OK:
            raise BrotliAmdError("%s failed (%d): %s" % (
                what, rc, self.L.brotli_amd_last_error(self.h).decode()))

    def max_output(self, n, params):
        return int(self.L.brotli_amd_max_output(n, C.byref(params)))

    # -- device-resident job (torch uint8 tensors) --------------------------
    def encode_device(self, d_in, n, params, d_out, d_shard_sizes=None):
        """d_in: uint8 cuda tensor with >= n + INPUT_SLACK elements; d_out:
        uint8 cuda tensor.  Returns (out_bytes, info dict)."""
        assert d_in.is_cuda and d_in.numel() >= n + INPUT_SLACK
        _wait_for_torch(d_in)
        info = JobInfo()
        out_size = C.c_uint64(0)
        rc = self.L.brotli_amd_encode_device(
            self.h, d_in.data_ptr(), n, C.byref(params), d_out.data_ptr(),
            d_out.numel(), C.byref(out_size),
            d_shard_sizes.data_ptr() if d_shard_sizes is not None else None,
            C.byref(info))
        self._check(rc, "brotli_amd_encode_device")
        return int(out_size.value), info.as_dict()

    def encode_host(self, data, params):
        data = bytes(data)
        cap = self.max_output(len(data), params)
        if cap == 0:
            raise BrotliAmdError("parameters outside the GPU path")
        out = C.create_string_buffer(cap)
        info = JobInfo()
        out_size = C.c_uint64(0)
        rc = self.L.brotli_amd_encode_host(self.h, data, len(data), C.byref(params),
                                           out, cap, C.byref(out_size), C.byref(info))
        self._check(rc, "brotli_amd_encode_host")
        return out.raw[:out_size.value], info.as_dict()

    # -- quality 1 (two-pass fragment compressor) ------------------------------
    def _fast_args(self, n, lgwin, call_sizes, is_last, carry):
        call_sizes = [n] if call_sizes is None else list(call_sizes)
        sizes = (C.c_uint64 * len(call_sizes))(*call_sizes)
        bits, value = fast_stream_header(lgwin) if carry is None else carry
        return sizes, len(call_sizes), FastParams(lgwin, bits, value, 1 if is_last else 0)

    def fast_max_output(self, n, ncalls, lgwin):
        return int(self.L.brotli_amd_fast_max_output(n, ncalls, lgwin))

    def encode_fast_device(self, d_in, n, d_out, lgwin=22, call_sizes=None, is_last=True, carry=None):
        """One run of quality-1 CompressStream calls (call k feeds call_sizes[k]
        bytes); returns (out_bits, info).  `carry` = (bits, value) pending at the
        start of the output, default: the stream header."""
        assert d_in.is_cuda and d_in.numel() >= n + INPUT_SLACK
        _wait_for_torch(d_in)
        sizes, ncalls, p = self._fast_args(n, lgwin, call_sizes, is_last, carry)
        info = JobInfo()
        out_bits = C.c_uint64(0)
        rc = self.L.brotli_amd_encode_fast_device(
            self.h, d_in.data_ptr(), n, sizes, ncalls, C.byref(p), d_out.data_ptr(),
            d_out.numel(), C.byref(out_bits), C.byref(info))
        self._check(rc, "brotli_amd_encode_fast_device")
        return int(out_bits.value), info.as_dict()

    def encode_fast_host(self, data, lgwin=22, call_sizes=None, is_last=True, carry=None):
        data = bytes(data)
        sizes, ncalls, p = self._fast_args(len(data), lgwin, call_sizes, is_last, carry)
        cap = self.fast_max_output(len(data), ncalls, lgwin)
        if cap == 0:
            raise BrotliAmdError("parameters outside the GPU path")
        out = C.create_string_buffer(cap)
        info = JobInfo()
        out_bits = C.c_uint64(0)
        rc = self.L.brotli_amd_encode_fast_host(self.h, data, len(data), sizes, ncalls,
                                                C.byref(p), out, cap, C.byref(out_bits), C.byref(info))
        self._check(rc, "brotli_amd_encode_fast_host")
        return out.raw[:(out_bits.value + 7) // 8], int(out_bits.value), info.as_dict()

    # -- decoder (k_decode.h): round trips on the device -------------------------
    @staticmethod
    def _pieces(pieces, n_in, n_out):
        if pieces is None:
            pieces = [(0, n_in, 0, n_out, PIECE_HEADER, 0)]
        arr = (DecodePiece * len(pieces))(*[DecodePiece(*p) for p in pieces])
        return arr, (DecodeResult * len(pieces))()

    def decode_device(self, d_in, n_in, d_out, n_out, pieces=None, check=True):
        """d_in: uint8 cuda tensor with >= n_in + DECODE_SLACK elements; d_out: uint8 cuda tensor of
        >= n_out.  `pieces`: [(in_off, in_len, out_off, out_cap, flags, lgwin)], default one stream.
        Returns (results, kernel ms); results[k] = (out_bytes, error, finished)."""
        assert d_in.is_cuda and d_in.numel() >= n_in + DECODE_SLACK and d_out.numel() >= n_out
        _wait_for_torch(d_in)
        arr, res = self._pieces(pieces, n_in, n_out)
        ms = C.c_float(0)
        rc = self.L.brotli_amd_decode_device(self.h, d_in.data_ptr(), n_in, arr, len(arr),
                                             d_out.data_ptr(), n_out, res, C.byref(ms))
        if check or rc not in (OK, DEVICE_FAULT):
            self._check(rc, "brotli_amd_decode_device")
        return [(int(r.out_bytes), int(r.error), int(r.finished)) for r in res], float(ms.value)

    def decode_host(self, comp, n_out, pieces=None, check=True, with_bits=False):
        """Host buffers.  Returns (bytes, results) — and the bits every piece consumed with
        `with_bits` (a stream that ends before its input does has trailing data)."""
        comp = bytes(comp)
        arr, res = self._pieces(pieces, len(comp), n_out)
        out = C.create_string_buffer(max(n_out, 1))
        ms = C.c_float(0)
        rc = self.L.brotli_amd_decode_host(self.h, comp, len(comp), arr, len(arr), out, n_out, res,
                                           C.byref(ms))
        if check or rc not in (OK, DEVICE_FAULT):
            self._check(rc, "brotli_amd_decode_host")
        results = [(int(r.out_bytes), int(r.error), int(r.finished)) for r in res]
        if with_bits:
            return out.raw[:n_out], results, [int(r.in_bits) for r in res]
        return out.raw[:n_out], results

    def debug_parse(self, d_in, n, params):
        cap = n // 2 + 64 * (1 + (n // max(1, params.shard_size or n)))
        arr = np.zeros(cap, dtype=CMD_DTYPE)
        info = JobInfo()
        ncmds = C.c_uint64(0)
        rc = self.L.brotli_amd_debug_parse(self.h, d_in.data_ptr(), n, C.byref(params),
                                           arr.ctypes.data, cap, C.byref(ncmds),
                                           C.byref(info))
        self._check(rc, "brotli_amd_debug_parse")
        return arr[:ncmds.value], info.as_dict()


def _wait_for_torch(t):
    """The library works on a HIP stream of its own (non-blocking: it does not order itself
    behind torch's streams) and returns when its work is done.  Whatever torch still has in
    flight for the input — a fill, a copy — has to land first."""
    import torch
    torch.cuda.current_stream(t.device).synchronize()


def to_device(data, device=0):
    """bytes -> uint8 cuda tensor with the required slack."""
    import torch
    a = np.frombuffer(bytes(data), dtype=np.uint8)
    t = torch.zeros(len(a) + INPUT_SLACK, dtype=torch.uint8, device="cuda:%d" % device)
    t[:len(a)] = torch.from_numpy(a.copy()).to(t.device)
    return t
