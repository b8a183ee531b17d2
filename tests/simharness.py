"""Test-only ctypes wrapper around tests/simt/libsimkernels.so (kernel logic on
the host SIMT simulator) and the oracle's command tap."""
import ctypes as C
import os
import subprocess

import numpy as np

from refharness import ROOT, TABLES, Oracle

SIM_SO = os.path.join(ROOT, "tests", "simt", "libsimkernels.so")

CMD_DTYPE = np.dtype([("insert_len", "<u4"), ("copy_len", "<u4"),
                      ("dist_extra", "<u4"), ("cmd_prefix", "<u2"),
                      ("dist_prefix", "<u2")])


def build_sim():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "simt")],
                   check=True)


class Sim:
    def __init__(self):
        build_sim()
        self.L = C.CDLL(SIM_SO)
        self.L.sim_parse.restype = C.c_long
        self.L.sim_parse.argtypes = [
            C.c_char_p, C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_uint32,
            C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
            C.POINTER(C.c_uint64)]

        self.L.sim_encode_fast.restype = C.c_long
        self.L.sim_encode_fast.argtypes = [
            C.c_char_p, C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_uint64), C.c_size_t,
            C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
        self.L.sim_stream.restype = C.c_long
        self.L.sim_stream.argtypes = [
            C.c_char_p, C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_uint32, C.c_uint32,
            C.POINTER(C.c_uint64), C.POINTER(C.c_uint8), C.c_size_t, C.c_int, C.c_char_p, C.c_size_t]
        self.L.sim_decode.restype = C.c_long
        self.L.sim_decode.argtypes = [
            C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.c_size_t, C.c_uint32, C.c_int,
            C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64)]
        self.L.sim_encode.restype = C.c_long
        self.L.sim_encode.argtypes = [
            C.c_char_p, C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_uint32,
            C.c_size_t, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t]

    def encode_stream(self, data, lgwin=22, size_hint=0, reverse=0, flags=0):
        """One unpartitioned quality-5 stream on the tiled path (JOB_FLAG_STREAMT).  Returns (bytes, info) — bytes
        is None when the stream leaves the tiled path; info = (reasons, sweeps, meta-blocks)."""
        self.L.sim_encode_stream.restype = C.c_long
        self.L.sim_encode_stream.argtypes = [
            C.c_char_p, C.c_char_p, C.c_size_t, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_char_p, C.c_size_t,
            C.POINTER(C.c_uint32)]
        cap = len(data) + len(data) // 512 + 8192
        out = C.create_string_buffer(cap)
        info = (C.c_uint32 * 4)()
        n = self.L.sim_encode_stream(TABLES.encode(), bytes(data), len(data), lgwin, size_hint, reverse, flags,
                                     out, cap, info)
        if n == -10:
            return None, tuple(info[:3])
        assert n >= 0, n
        return out.raw[:n], tuple(info[:3])

    def decode(self, comp, n_out, pieces=None, arena_words=0, reverse=0):
        """k_decode on the simulator.  pieces = [(in_off, in_len, out_off, out_cap, flags, lgwin)],
        default: one whole stream.  Returns (bytes, [(out_bytes, in_bits, error, finished)])."""
        if pieces is None:
            pieces = [(0, len(comp), 0, n_out, 1, 0)]
        arr = (C.c_uint64 * (5 * len(pieces)))()
        for k, (io, il, oo, oc, fl, lw) in enumerate(pieces):
            arr[5 * k:5 * k + 5] = [io, il, oo, oc, fl | (lw << 32)]
        out = C.create_string_buffer(n_out + 64)
        res = (C.c_uint64 * (4 * len(pieces)))()
        comp = bytes(comp)
        rc = self.L.sim_decode(TABLES.encode(), comp, len(comp), arr, len(pieces), arena_words, reverse,
                               out, n_out, res)
        assert rc == 0, rc
        return out.raw[:n_out], [(res[4 * k], res[4 * k + 1], res[4 * k + 2] & 0xFFFFFFFF, res[4 * k + 2] >> 32)
                                 for k in range(len(pieces))]

    def stream(self, data, calls, quality=5, lgwin=22, size_hint=0, stream_offset=0, reverse=0):
        """One encoder instance driven call by call on the simulator (k_parse / k_parse_deep +
        k_build + k_store); `calls` = [(nbytes, op)], op 3 = EMIT_METADATA."""
        sizes = (C.c_uint64 * len(calls))(*[c[0] for c in calls])
        ops = (C.c_uint8 * len(calls))(*[c[1] for c in calls])
        cap = 2 * len(data) + 4096 + 64 * len(calls)
        out = C.create_string_buffer(cap)
        n = self.L.sim_stream(TABLES.encode(), bytes(data), len(data), quality, lgwin, size_hint,
                              stream_offset, sizes, ops, len(calls), reverse, out, cap)
        assert n >= 0, n
        return out.raw[:n]

    def encode_fast(self, data, lgwin=22, calls=None, reverse=0):
        """Quality 1 through the k_fast_* kernels; `calls` = [(nbytes, op), ...] ending
        with FINISH, no FLUSH inside (the host layer splits runs at flushes)."""
        if calls is None:
            calls = [(len(data), 2)]
        assert calls[-1][1] == 2 and all(c[1] == 0 for c in calls[:-1])
        sizes = (C.c_uint64 * len(calls))(*[c[0] for c in calls])
        hl = max(lgwin, 18)
        cap = len(data) + 64 * len(calls) + 64 * (len(data) >> min(lgwin, 17)) + 1024
        out = C.create_string_buffer(cap)
        nbits = self.L.sim_encode_fast(TABLES.encode(), bytes(data), len(data), lgwin, sizes,
                                       len(calls), 4, ((hl - 17) << 1) | 1, 1, reverse, out, cap)
        assert nbits >= 0 and nbits % 8 == 0, nbits
        return out.raw[:nbits // 8]

    def encode(self, data, quality=5, lgwin=22, size_hint=0, shard_size=0,
               stream_base=0, is_last=True, reverse=0, flags=0):
        nsh = 1 if not shard_size else max(1, -(-len(data) // shard_size))
        cap = 2 * len(data) + 2048 * (nsh + 1)
        out = C.create_string_buffer(cap)
        n = self.L.sim_encode(TABLES.encode(), bytes(data), len(data), quality,
                              lgwin, size_hint, shard_size, stream_base,
                              1 if is_last else 0, reverse, flags, out, cap)
        assert n >= 0, n
        return out.raw[:n]

    def parse(self, data, quality=5, lgwin=22, size_hint=0, shard_size=0,
              reverse=0, no_pair=0):
        cap = len(data) // 2 + 64
        arr = np.zeros(cap, dtype=CMD_DTYPE)
        stats = (C.c_uint64 * 3)()
        n = self.L.sim_parse(TABLES.encode(), bytes(data), len(data), quality,
                             lgwin, size_hint, shard_size, reverse, no_pair,
                             arr.ctypes.data, cap, stats)
        assert n >= 0, n
        return arr[:n], list(stats)


def oracle_commands(oracle, data, quality=5, lgwin=22, size_hint=0,
                    shard_size=0):
    """Command lists (first real meta-block of every shard), concatenated.
    The 2-byte flint meta-block of continuation shards (its single insert-only
    command) is dropped: the device emits that block inline."""
    L = oracle.L
    L.oracle_set_command_tap.argtypes = [C.c_void_p, C.c_size_t,
                                         C.POINTER(C.c_size_t)]
    n_total = len(data)
    if not shard_size or shard_size >= n_total:
        shard_size = n_total
    hint = size_hint or min(n_total, 1 << 30)
    out = []
    off = 0
    while off < n_total:
        m = min(shard_size, n_total - off)
        cap = m // 2 + 64
        arr = np.zeros(cap, dtype=CMD_DTYPE)
        n = C.c_size_t(0)
        L.oracle_set_command_tap(arr.ctypes.data, cap, C.byref(n))
        try:
            oracle.encode_shard(data[off:off + m], quality, lgwin, hint,
                                min(off, 1 << 30), off + m == n_total)
        finally:
            L.oracle_set_command_tap(None, 0, None)
        cmds = arr[:n.value]
        if off and m > 2:
            cmds = cmds[1:]
        out.append(cmds)
        off += m
    return np.concatenate(out)
