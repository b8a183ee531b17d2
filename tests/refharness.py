"""Test-only helpers: ctypes access to the checkers under oracle/.

`Ref`  = the real reference encoder/decoder (oracle/_ref/libbrotli_ref.so, built
         by oracle/Makefile from /root/reference; travels to the GPU box).
`Oracle` = the C restatement (oracle/liboracle.so).
Nothing in the product imports this module.
"""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libbrotli_ref.so")
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
TABLES = os.path.join(ROOT, "brotli_amd", "data", "brotli_tables.bin")

# BrotliEncoderParameter ids, c/include/brotli/encode.h:160-265
PARAM_MODE, PARAM_QUALITY, PARAM_LGWIN, PARAM_LGBLOCK = 0, 1, 2, 3
PARAM_SIZE_HINT, PARAM_STREAM_OFFSET = 5, 9
OP_PROCESS, OP_FLUSH, OP_FINISH = 0, 1, 2


def have_ref():
    return os.path.exists(REF_SO)


def have_oracle():
    return os.path.exists(ORACLE_SO)


class Ref:
    def __init__(self, path=REF_SO):
        L = C.CDLL(path)
        self.L = L
        L.BrotliEncoderCreateInstance.restype = C.c_void_p
        L.BrotliEncoderCreateInstance.argtypes = [C.c_void_p] * 3
        L.BrotliEncoderDestroyInstance.argtypes = [C.c_void_p]
        L.BrotliEncoderSetParameter.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
        L.BrotliEncoderCompressStream.argtypes = [
            C.c_void_p, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_void_p),
            C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.BrotliEncoderIsFinished.argtypes = [C.c_void_p]
        L.BrotliEncoderHasMoreOutput.argtypes = [C.c_void_p]
        L.BrotliEncoderCompress.argtypes = [
            C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_char_p,
            C.POINTER(C.c_size_t), C.c_char_p]
        L.BrotliEncoderMaxCompressedSize.restype = C.c_size_t
        L.BrotliEncoderMaxCompressedSize.argtypes = [C.c_size_t]
        if hasattr(L, "BrotliDecoderDecompress"):
            L.BrotliDecoderDecompress.argtypes = [
                C.c_size_t, C.c_char_p, C.POINTER(C.c_size_t), C.c_char_p]

    def compress(self, data, quality=5, lgwin=22):
        cap = max(self.L.BrotliEncoderMaxCompressedSize(len(data)), 16)
        out = C.create_string_buffer(cap)
        n = C.c_size_t(cap)
        ok = self.L.BrotliEncoderCompress(quality, lgwin, 0, len(data), data,
                                          C.byref(n), out)
        assert ok
        return out.raw[:n.value]

    def encode_shard(self, data, quality, lgwin, size_hint, stream_offset,
                     is_last):
        """One instance per shard, SURVEY.md §8(e) contract."""
        L = self.L
        st = L.BrotliEncoderCreateInstance(None, None, None)
        assert L.BrotliEncoderSetParameter(st, PARAM_QUALITY, quality)
        assert L.BrotliEncoderSetParameter(st, PARAM_LGWIN, lgwin)
        assert L.BrotliEncoderSetParameter(st, PARAM_SIZE_HINT, size_hint)
        if stream_offset:
            assert L.BrotliEncoderSetParameter(st, PARAM_STREAM_OFFSET,
                                               stream_offset)
        cap = 2 * len(data) + 1024
        out = C.create_string_buffer(cap)
        inbuf = C.create_string_buffer(bytes(data), len(data))
        avail_in = C.c_size_t(len(data))
        next_in = C.c_void_p(C.addressof(inbuf))
        avail_out = C.c_size_t(cap)
        next_out = C.c_void_p(C.addressof(out))
        total = C.c_size_t(0)
        op = OP_FINISH if is_last else OP_FLUSH
        while True:
            ok = L.BrotliEncoderCompressStream(
                st, op, C.byref(avail_in), C.byref(next_in),
                C.byref(avail_out), C.byref(next_out), C.byref(total))
            assert ok
            if avail_in.value == 0 and not L.BrotliEncoderHasMoreOutput(st):
                break
        L.BrotliEncoderDestroyInstance(st)
        return out.raw[:total.value]

    def encode_calls(self, data, quality, lgwin, calls, size_hint=0):
        """One instance driven with an explicit call sequence [(nbytes, op), ...];
        every call is repeated until its input is consumed and the output drained
        (what the CLI / bindings do)."""
        L = self.L
        st = L.BrotliEncoderCreateInstance(None, None, None)
        assert L.BrotliEncoderSetParameter(st, PARAM_QUALITY, quality)
        assert L.BrotliEncoderSetParameter(st, PARAM_LGWIN, lgwin)
        if size_hint:
            assert L.BrotliEncoderSetParameter(st, PARAM_SIZE_HINT, size_hint)
        cap = 2 * len(data) + 1024 + 64 * len(calls)
        out = C.create_string_buffer(cap)
        inbuf = C.create_string_buffer(bytes(data), max(len(data), 1))
        avail_out = C.c_size_t(cap)
        next_out = C.c_void_p(C.addressof(out))
        total = C.c_size_t(0)
        pos = 0
        for nbytes, op in calls:
            avail_in = C.c_size_t(nbytes)
            next_in = C.c_void_p(C.addressof(inbuf) + pos)
            pos += nbytes
            while True:
                ok = L.BrotliEncoderCompressStream(
                    st, op, C.byref(avail_in), C.byref(next_in),
                    C.byref(avail_out), C.byref(next_out), C.byref(total))
                assert ok
                if avail_in.value == 0 and not L.BrotliEncoderHasMoreOutput(st):
                    break
        L.BrotliEncoderDestroyInstance(st)
        # (metadata payloads are copied to next_out without being counted in total_out,
        # encode.c:1590-1600: measure what actually left)
        return out.raw[:cap - avail_out.value]

    def encode_plan(self, data, quality, lgwin, shard_size):
        n = len(data)
        if n == 0:
            return b"\x06"
        if shard_size == 0 or shard_size >= n:
            shard_size = n
        hint = min(n, 1 << 30)
        parts = []
        off = 0
        while off < n:
            m = min(shard_size, n - off)
            parts.append(self.encode_shard(
                data[off:off + m], quality, lgwin, hint, min(off, 1 << 30),
                off + m == n))
            off += m
        return b"".join(parts)

    def decompress(self, comp, max_out):
        out = C.create_string_buffer(max(max_out, 1))
        n = C.c_size_t(max_out)
        r = self.L.BrotliDecoderDecompress(len(comp), comp, C.byref(n), out)
        assert r == 1, "decoder rejected stream (result %d)" % r
        return out.raw[:n.value]


class Oracle:
    def __init__(self, path=ORACLE_SO):
        L = C.CDLL(path)
        self.L = L
        L.oracle_init.argtypes = [C.c_char_p]
        assert L.oracle_init(TABLES.encode()) == 0
        L.oracle_encode_shard.restype = C.c_size_t
        L.oracle_encode_shard.argtypes = [
            C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_uint32, C.c_uint32,
            C.c_int, C.c_char_p, C.c_size_t]
        L.oracle_encode_fast.restype = C.c_size_t
        L.oracle_encode_fast.argtypes = [
            C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_uint64),
            C.POINTER(C.c_uint8), C.c_size_t, C.c_char_p, C.c_size_t]
        L.oracle_encode_plan.restype = C.c_size_t
        L.oracle_encode_plan.argtypes = [
            C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_size_t, C.c_char_p,
            C.c_size_t, C.POINTER(C.c_uint64)]

    def encode_fast(self, data, lgwin=22, calls=None):
        """Quality 1; `calls` = [(nbytes, op), ...], default one FINISH call."""
        if calls is None:
            calls = [(len(data), OP_FINISH)]
        sizes = (C.c_uint64 * len(calls))(*[c[0] for c in calls])
        ops = (C.c_uint8 * len(calls))(*[c[1] for c in calls])
        cap = 2 * len(data) + 1024 + 64 * len(calls)
        out = C.create_string_buffer(cap)
        n = self.L.oracle_encode_fast(bytes(data), len(data), lgwin, sizes, ops,
                                      len(calls), out, cap)
        assert n > 0
        return out.raw[:n]

    def encode_shard(self, data, quality, lgwin, size_hint, stream_offset,
                     is_last):
        cap = 2 * len(data) + 1024
        out = C.create_string_buffer(cap)
        n = self.L.oracle_encode_shard(bytes(data), len(data), quality, lgwin,
                                       size_hint, stream_offset,
                                       1 if is_last else 0, out, cap)
        assert n > 0
        return out.raw[:n]

    def encode_plan(self, data, quality=5, lgwin=22, shard_size=0):
        nsh = 1 if not shard_size else max(1, -(-len(data) // shard_size))
        cap = 2 * len(data) + 1024 * (nsh + 1)
        out = C.create_string_buffer(cap)
        n = self.L.oracle_encode_plan(bytes(data), len(data), quality, lgwin,
                                      shard_size, out, cap, None)
        assert n > 0
        return out.raw[:n]
