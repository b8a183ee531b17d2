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
        L.BrotliEncoderPrepareDictionary.restype = C.c_void_p
        L.BrotliEncoderPrepareDictionary.argtypes = [C.c_int, C.c_size_t, C.c_char_p, C.c_int,
                                                     C.c_void_p, C.c_void_p, C.c_void_p]
        L.BrotliEncoderDestroyPreparedDictionary.argtypes = [C.c_void_p]
        L.BrotliEncoderAttachPreparedDictionary.argtypes = [C.c_void_p, C.c_void_p]
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
                     is_last, dictionaries=(), lgblock=0):
        """One instance per shard, SURVEY.md §8(e) contract; `dictionaries` are attached to it."""
        L = self.L
        st = L.BrotliEncoderCreateInstance(None, None, None)
        assert L.BrotliEncoderSetParameter(st, PARAM_QUALITY, quality)
        assert L.BrotliEncoderSetParameter(st, PARAM_LGWIN, lgwin)
        assert L.BrotliEncoderSetParameter(st, PARAM_SIZE_HINT, size_hint)
        if lgblock:
            assert L.BrotliEncoderSetParameter(st, PARAM_LGBLOCK, lgblock)
        if stream_offset:
            assert L.BrotliEncoderSetParameter(st, PARAM_STREAM_OFFSET,
                                               stream_offset)
        keep = [C.create_string_buffer(bytes(d), max(len(d), 1)) for d in dictionaries]
        prepared = []
        for d, buf in zip(dictionaries, keep):
            pd = L.BrotliEncoderPrepareDictionary(0, len(d), buf, 11, None, None, None)
            assert pd and L.BrotliEncoderAttachPreparedDictionary(st, pd)
            prepared.append(pd)
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
        for pd in prepared:
            L.BrotliEncoderDestroyPreparedDictionary(pd)
        return out.raw[:total.value]

    def encode_calls(self, data, quality, lgwin, calls, size_hint=0, dictionaries=()):
        """One instance driven with an explicit call sequence [(nbytes, op), ...];
        every call is repeated until its input is consumed and the output drained
        (what the CLI / bindings do).  `dictionaries`: raw LZ77 prefixes prepared and
        attached in order before the first call (encode.h:318-363)."""
        L = self.L
        st = L.BrotliEncoderCreateInstance(None, None, None)
        assert L.BrotliEncoderSetParameter(st, PARAM_QUALITY, quality)
        assert L.BrotliEncoderSetParameter(st, PARAM_LGWIN, lgwin)
        if size_hint:
            assert L.BrotliEncoderSetParameter(st, PARAM_SIZE_HINT, size_hint)
        keep = [C.create_string_buffer(bytes(d), max(len(d), 1)) for d in dictionaries]
        prepared = []
        for d, buf in zip(dictionaries, keep):
            pd = L.BrotliEncoderPrepareDictionary(0, len(d), buf, 11, None, None, None)
            assert pd
            assert L.BrotliEncoderAttachPreparedDictionary(st, pd)
            prepared.append(pd)
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
        for pd in prepared:
            L.BrotliEncoderDestroyPreparedDictionary(pd)
        # (metadata payloads are copied to next_out without being counted in total_out,
        # encode.c:1590-1600: measure what actually left)
        return out.raw[:cap - avail_out.value]

    def encode_plan(self, data, quality, lgwin, shard_size, dictionaries=(), lgblock=0):
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
                off + m == n, dictionaries, lgblock=lgblock))
            off += m
        return b"".join(parts)

    def decompress_with(self, comp, max_out, dictionaries):
        """Streaming decoder with raw dictionaries attached (c/include/brotli/decode.h
        BrotliDecoderAttachDictionary)."""
        L = self.L
        L.BrotliDecoderCreateInstance.restype = C.c_void_p
        L.BrotliDecoderCreateInstance.argtypes = [C.c_void_p] * 3
        L.BrotliDecoderDestroyInstance.argtypes = [C.c_void_p]
        L.BrotliDecoderAttachDictionary.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_char_p]
        L.BrotliDecoderDecompressStream.argtypes = [
            C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_void_p),
            C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        st = L.BrotliDecoderCreateInstance(None, None, None)
        keep = [C.create_string_buffer(bytes(d), max(len(d), 1)) for d in dictionaries]
        for d, buf in zip(dictionaries, keep):
            assert L.BrotliDecoderAttachDictionary(st, 0, len(d), buf)
        out = C.create_string_buffer(max(max_out, 1))
        inbuf = C.create_string_buffer(bytes(comp), max(len(comp), 1))
        avail_in, next_in = C.c_size_t(len(comp)), C.c_void_p(C.addressof(inbuf))
        avail_out, next_out = C.c_size_t(max_out), C.c_void_p(C.addressof(out))
        total = C.c_size_t(0)
        r = L.BrotliDecoderDecompressStream(st, C.byref(avail_in), C.byref(next_in),
                                            C.byref(avail_out), C.byref(next_out), C.byref(total))
        L.BrotliDecoderDestroyInstance(st)
        assert r == 1, "decoder rejected stream (result %d)" % r
        return out.raw[:max_out - avail_out.value]

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

        L.oracle_set_dictionary.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t]
        self._dict_keep = None

    def set_dictionary(self, dictionaries=()):
        """Attach raw dictionaries to every encoder instance created from now on; () detaches."""
        n = len(dictionaries)
        bufs = [C.create_string_buffer(bytes(d), max(len(d), 1)) for d in dictionaries]
        ptrs = (C.c_char_p * max(n, 1))(*[C.cast(b, C.c_char_p) for b in bufs])
        sizes = (C.c_size_t * max(n, 1))(*[len(d) for d in dictionaries])
        assert self.L.oracle_set_dictionary(ptrs, sizes, n) == 1
        self._dict_keep = bufs

    def set_lgblock(self, lgblock=0):
        """BROTLI_PARAM_LGBLOCK of every encoder instance created from now on; 0 = the default."""
        self.L.oracle_set_lgblock(int(lgblock))

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
