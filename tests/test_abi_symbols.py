"""The C-ABI libraries load and export every symbol that include/*.h declares
(no compute calls: this runs without a GPU)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "brotli_amd", "lib")


def _ensure_built():
    if not (os.path.exists(os.path.join(LIBDIR, "libbrotli_amd_hip.so")) and
            os.path.exists(os.path.join(LIBDIR, "libbrotlienc_amd.so"))):
        subprocess.run(["make", "-C", os.path.join(ROOT, "brotli_amd", "csrc")], check=True)


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b((?:brotli_amd_|BrotliEncoder)\w+)\s*\(", text)
    return sorted(set(n for n in names if not n.endswith("_func")))


@pytest.mark.parametrize("header,lib", [("brotli_amd_hip.h", "libbrotli_amd_hip.so"),
                                        ("brotli_amd_encode.h", "libbrotlienc_amd.so")])
def test_exports_every_declared_symbol(header, lib):
    _ensure_built()
    names = _declared(header)
    assert len(names) >= 9
    L = C.CDLL(os.path.join(LIBDIR, lib))
    for n in names:
        assert hasattr(L, n), "%s does not export %s" % (lib, n)


def test_boundary_matches_reference_symbol_list():
    """the symbols of libbrotlienc.so.1 (13 BROTLI_ENC_API + the two
    BROTLI_ENC_EXTRA_API entries of 1.2.0, c/include/brotli/encode.h:531-535)."""
    _ensure_built()
    want = {"BrotliEncoderCreateInstance", "BrotliEncoderDestroyInstance", "BrotliEncoderSetParameter",
            "BrotliEncoderCompress", "BrotliEncoderCompressStream", "BrotliEncoderIsFinished",
            "BrotliEncoderHasMoreOutput", "BrotliEncoderTakeOutput", "BrotliEncoderMaxCompressedSize",
            "BrotliEncoderVersion", "BrotliEncoderPrepareDictionary",
            "BrotliEncoderDestroyPreparedDictionary", "BrotliEncoderAttachPreparedDictionary",
            "BrotliEncoderEstimatePeakMemoryUsage", "BrotliEncoderGetPreparedDictionarySize"}
    ref_header = "/root/reference/c/include/brotli/encode.h"
    if os.path.exists(ref_header):     # (this container only) the list IS the reference header's
        import re
        text = open(ref_header).read()
        assert set(re.findall(r"BROTLI_ENC(?:_EXTRA)?_API[^;(]*?(BrotliEncoder\w+)\s*\(", text)) == want
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(LIBDIR, "libbrotlienc_amd.so")],
                         capture_output=True, text=True, check=True).stdout
    got = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert want <= got
    assert not [s for s in got if s.startswith("Brotli") and s not in want]


def test_protocol_without_device():
    """Host-side protocol of the boundary that needs no GPU: parameter latching
    and validation (encode.c:60-123), the size bound, the version."""
    _ensure_built()
    L = C.CDLL(os.path.join(LIBDIR, "libbrotlienc_amd.so"))
    L.BrotliEncoderCreateInstance.restype = C.c_void_p
    L.BrotliEncoderCreateInstance.argtypes = [C.c_void_p] * 3
    L.BrotliEncoderSetParameter.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
    L.BrotliEncoderDestroyInstance.argtypes = [C.c_void_p]
    L.BrotliEncoderMaxCompressedSize.restype = C.c_size_t
    L.BrotliEncoderMaxCompressedSize.argtypes = [C.c_size_t]
    L.BrotliEncoderVersion.restype = C.c_uint32
    assert L.BrotliEncoderVersion() == 0x1002000
    assert L.BrotliEncoderMaxCompressedSize(0) == 2
    assert L.BrotliEncoderMaxCompressedSize(1 << 20) == (1 << 20) + 2 + 4 * 64 + 3 + 1
    s = L.BrotliEncoderCreateInstance(None, None, None)
    assert s
    assert L.BrotliEncoderSetParameter(s, 1, 5) == 1          # QUALITY
    assert L.BrotliEncoderSetParameter(s, 9, (1 << 30) + 1) == 0   # STREAM_OFFSET too large
    assert L.BrotliEncoderSetParameter(s, 12, 3) == 0         # SIMD_HASHER out of range
    assert L.BrotliEncoderSetParameter(s, 4, 2) == 0          # DISABLE_LITERAL_CONTEXT_MODELING not 0/1
    assert L.BrotliEncoderSetParameter(s, 77, 1) == 0         # unknown id
    L.BrotliEncoderDestroyInstance(s)
