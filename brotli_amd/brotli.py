"""Python mirror of the reference binding's encoder surface (python/brotli.py:26-53,
python/_brotli.c:403-470 `Compressor`, :568-640 `compress`) on top of the drop-in
C ABI library brotli_amd/lib/libbrotlienc_amd.so.

    import brotli_amd.brotli as brotli
    data = brotli.compress(b"...", quality=5, lgwin=22)
    c = brotli.Compressor(quality=5); out = c.process(chunk) + c.flush() + c.finish()

Names, argument meaning and error behaviour follow the reference module:
`brotli.error` is raised where the reference raises it (invalid parameters,
use after finish(), failed compression).  There is no CPU encoder: parameters
outside the GPU path raise `brotli.error`.  `decompress` (python/brotli.py:56-73)
runs the device decoder (brotli_amd/csrc/k_decode.h) through the HIP layer: one
stream = one wave, meant for round trips and checks, not for throughput.

Extension: `shard_size=<bytes>` (module-level `compress` and `Compressor`)
selects a partition plan; 0 = single stream, bytes
identical to the stock library.
"""
import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("BROTLI_AMD_ENC_LIB") or os.path.join(_HERE, "lib", "libbrotlienc_amd.so")

MODE_GENERIC, MODE_TEXT, MODE_FONT = 0, 1, 2          # python/brotli.py:15-23
_OP_PROCESS, _OP_FLUSH, _OP_FINISH = 0, 1, 2
_P_MODE, _P_QUALITY, _P_LGWIN, _P_LGBLOCK, _P_SIZE_HINT = 0, 1, 2, 3, 5
_P_AMD_SHARD_BYTES = 0x4D490001


class error(Exception):
    """Same role as `brotli.error` of the reference binding."""


_lib = None
_lib_lock = threading.Lock()


def _load():
    global _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(_LIB_PATH):
                raise error("%s is missing: build it with __graft_entry__.build()" % _LIB_PATH)
            L = C.CDLL(_LIB_PATH)
            L.BrotliEncoderCreateInstance.restype = C.c_void_p
            L.BrotliEncoderCreateInstance.argtypes = [C.c_void_p] * 3
            L.BrotliEncoderDestroyInstance.argtypes = [C.c_void_p]
            L.BrotliEncoderSetParameter.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
            L.BrotliEncoderCompressStream.argtypes = [
                C.c_void_p, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_void_p),
                C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
            L.BrotliEncoderIsFinished.argtypes = [C.c_void_p]
            L.BrotliEncoderHasMoreOutput.argtypes = [C.c_void_p]
            L.BrotliEncoderTakeOutput.restype = C.c_void_p
            L.BrotliEncoderTakeOutput.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
            L.BrotliEncoderVersion.restype = C.c_uint32
            _lib = L
    return _lib


def version():
    v = _load().BrotliEncoderVersion()
    return "%d.%d.%d" % (v >> 24, (v >> 12) & 0xFFF, v & 0xFFF)


class Compressor(object):
    """Streaming compressor (python/_brotli.c:403-470): process() / flush() /
    finish(); an instance must not be used from two threads at once — the
    reference raises `brotli.error` on concurrent use (:25-34) and so does this."""

    def __init__(self, mode=MODE_GENERIC, quality=11, lgwin=22, lgblock=0, size_hint=0, shard_size=0):
        L = _load()
        self._L = L
        self._busy = threading.Lock()
        self._finished = False
        self._st = L.BrotliEncoderCreateInstance(None, None, None)
        if not self._st:
            raise error("BrotliEncoderCreateInstance failed")
        # parameter validation as in python/_brotli.c:430-470
        if mode not in (MODE_GENERIC, MODE_TEXT, MODE_FONT):
            raise error("Invalid mode")
        if not 0 <= quality <= 11:
            raise error("Invalid quality. Range is 0 to 11.")
        if not 10 <= lgwin <= 24:
            raise error("Invalid lgwin. Range is 10 to 24.")
        if lgblock != 0 and not 16 <= lgblock <= 24:
            raise error("Invalid lgblock. Can be 0 or in range 16 to 24.")
        for p, v in ((_P_MODE, mode), (_P_QUALITY, quality), (_P_LGWIN, lgwin), (_P_LGBLOCK, lgblock)):
            L.BrotliEncoderSetParameter(self._st, p, v)
        if size_hint:
            L.BrotliEncoderSetParameter(self._st, _P_SIZE_HINT, min(size_hint, 1 << 30))
        if shard_size:
            L.BrotliEncoderSetParameter(self._st, _P_AMD_SHARD_BYTES, shard_size)

    def __del__(self):
        st, self._st = getattr(self, "_st", None), None
        if st:
            self._L.BrotliEncoderDestroyInstance(st)

    def _stream(self, data, op):
        if not self._busy.acquire(False):
            raise error("Concurrently sharing Compressor instances is not supported")
        try:
            if self._finished:
                raise error("BrotliEncoderCompressStream failed: stream is already finished")
            L = self._L
            data = bytes(data)
            buf = C.create_string_buffer(data, len(data)) if data else None
            avail_in = C.c_size_t(len(data))
            next_in = C.c_void_p(C.addressof(buf) if buf is not None else 0)
            out = bytearray()
            while True:
                avail_out = C.c_size_t(0)
                next_out = C.c_void_p(0)
                if not L.BrotliEncoderCompressStream(self._st, op, C.byref(avail_in), C.byref(next_in),
                                                     C.byref(avail_out), C.byref(next_out), None):
                    raise error("BrotliEncoderCompressStream failed while processing the stream")
                while True:          # TakeOutput, as the Go / Java bindings do
                    size = C.c_size_t(0)
                    p = L.BrotliEncoderTakeOutput(self._st, C.byref(size))
                    if not size.value:
                        break
                    out += C.string_at(p, size.value)
                if avail_in.value == 0 and not L.BrotliEncoderHasMoreOutput(self._st):
                    break
            if op == _OP_FINISH:
                self._finished = True
                if not L.BrotliEncoderIsFinished(self._st):
                    raise error("BrotliEncoderCompressStream failed while finishing the stream")
            return bytes(out)
        finally:
            self._busy.release()

    def process(self, string):
        """Feeds bytes; returns whatever output is ready (possibly b'')."""
        return self._stream(string, _OP_PROCESS)

    compress = process   # alias kept by the reference module for old callers

    def flush(self):
        """Everything fed so far becomes decodable."""
        return self._stream(b"", _OP_FLUSH)

    def finish(self):
        """Ends the stream; the object cannot be used afterwards."""
        return self._stream(b"", _OP_FINISH)


_dec_ctx = None


def decompress(string):
    """python/brotli.py:56-73: the decoded bytes of one complete Brotli stream; `brotli.error` if the
    stream is damaged or incomplete.  Device decoder (k_decode.h); the output size is not known in
    advance, so the buffer is grown until the stream fits."""
    data = bytes(string)
    ctx = _decoder_context()
    with _lib_lock:
        cap = max(1 << 16, 6 * len(data))
        while True:
            out, res, bits = ctx.decode_host(data, cap, check=False, with_bits=True)
            n, err, finished = res[0]
            if err == 6 and cap < (1 << 31):      # a meta-block announces more than the buffer holds (error 10,
                                                  # a command past its meta-block's length, is damage: no retry)
                cap *= 4
                continue
            if err != 0 or not finished:
                raise error("BrotliDecoderDecompress failed (device decoder error %d%s)" % (
                    err, "" if finished or err else ", stream incomplete"))
            if (bits[0] + 7) // 8 != len(data):          # python/_brotli.c:917: input left over is an error
                raise error("BrotliDecoderDecompress failed (data after the end of the stream)")
            return out[:n]


def _decoder_context():
    global _dec_ctx
    from . import hip
    with _lib_lock:
        if _dec_ctx is None:
            try:
                _dec_ctx = hip.Context(int(os.environ.get("BROTLI_AMD_DEVICE", "0")))
            except hip.BrotliAmdError as e:
                raise error(str(e))
    return _dec_ctx


class Decompressor(object):
    """python/_brotli.c:640-860 `Decompressor`: process(data) returns the bytes that became available,
    is_finished() tells whether the stream has ended.  The device decoder works on resident data: a
    call decodes what has arrived so far from the start, and output is handed out once the stream is
    complete (what a wave decodes from the last, cut-off bytes of a partial input is not trustworthy,
    so nothing is returned before that) — the concatenation of the returned pieces is the reference's,
    their timing is not (this class is for checks, not for streaming at rate).  `output_buffer_limit`
    is accepted and ignored, so can_accept_more_data() is always True."""

    def __init__(self):
        self._in = bytearray()
        self._finished = False
        self._cap = 1 << 16
        self._busy = threading.Lock()

    def is_finished(self):
        return self._finished

    def can_accept_more_data(self):
        return True

    def process(self, string, output_buffer_limit=None):
        if not self._busy.acquire(False):
            raise error("Concurrently sharing Decompressor instances is not supported")
        try:
            if self._finished:
                if len(string):
                    raise error("BrotliDecoderDecompressStream failed: data after the end of the stream")
                return b""
            self._in += bytes(string)
            if not self._in:
                return b""
            ctx = _decoder_context()
            data = bytes(self._in)
            with _lib_lock:     # the shared decoder context is not re-entrant (decompress() holds the same lock)
                while True:
                    out, res, bits = ctx.decode_host(data, self._cap, check=False, with_bits=True)
                    n, err, finished = res[0]
                    if err == 6 and self._cap < (1 << 31):   # a meta-block announces more than the buffer holds
                        self._cap *= 4
                        continue
                    break
            if err not in (0, 7) or (err == 7 and finished):
                raise error("BrotliDecoderDecompressStream failed (device decoder error %d)" % err)
            if not finished:
                return b""
            if (bits[0] + 7) // 8 != len(data):
                raise error("BrotliDecoderDecompressStream failed: data after the end of the stream")
            self._finished = True
            return out[:n]
        finally:
            self._busy.release()


def compress(string, mode=MODE_GENERIC, quality=11, lgwin=22, lgblock=0, shard_size=0):
    """Same two calls as the reference (python/brotli.py:51-53):
    Compressor(...).process(string) + .finish()."""
    c = Compressor(mode=mode, quality=quality, lgwin=lgwin, lgblock=lgblock, shard_size=shard_size)
    return c.process(string) + c.finish()
