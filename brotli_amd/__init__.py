"""brotli_amd — MI355X-native Brotli encoder hot path behind the reference's
C ABI.  See DESIGN.md; the HIP C-ABI binding lives in brotli_amd.hip."""
__version__ = "0.1.0"
