/* include/brotli_amd_encode.h — the drop-in boundary: the encoder C ABI of
 * google/brotli (libbrotlienc.so.1: the 13 BROTLI_ENC_API + 2 BROTLI_ENC_EXTRA_API symbols of 1.2.0) as
 * exported by brotli_amd/lib/libbrotlienc_amd.so.
 *
 * Every entry point below has the name, argument order, argument meaning and
 * error behaviour of the reference declaration cited next to it
 * (c/include/brotli/encode.h); callers compiled against the reference's own
 * <brotli/encode.h> (the CLI c/tools/brotli.c, python/_brotli.c, go/cbrotli,
 * java/org/brotli/wrapper/enc/encoder_jni.cc) link against this library
 * unchanged.  The hot path behind them runs on the GPU through
 * include/brotli_amd_hip.h; there is no CPU encoder in the library:
 * parameter combinations the kernels do not implement make
 * BrotliEncoderCompressStream / BrotliEncoderCompress return BROTLI_FALSE.
 *
 * Vendor extensions (all optional; without them the bytes equal the stock
 * library's for the same calls):
 *   env BROTLI_AMD_SHARD_KB=<n>   partition plan: n KiB per encoder shard
 *   env BROTLI_AMD_DEVICE=<i>     HIP device index (default 0)
 *   env BROTLI_AMD_TABLES=<path>  format tables blob (default: next to the .so)
 *   BrotliEncoderSetParameter(s, BROTLI_AMD_PARAM_SHARD_BYTES, bytes)
 */
#ifndef BROTLI_AMD_ENCODE_H_
#define BROTLI_AMD_ENCODE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef BROTLI_BOOL
#define BROTLI_BOOL int
#define BROTLI_TRUE 1
#define BROTLI_FALSE 0
#endif

/* c/include/brotli/types.h: allocation hooks */
typedef void* (*brotli_amd_alloc_func)(void* opaque, size_t size);
typedef void (*brotli_amd_free_func)(void* opaque, void* address);

/* Opaque handles (same tags as the reference so that both headers can be
   included by one translation unit). */
typedef struct BrotliEncoderStateStruct BrotliEncoderState;
typedef struct BrotliEncoderPreparedDictionaryStruct BrotliEncoderPreparedDictionary;

/* Parameter ids outside the reference's 0..12 range (encode.h:160-265); the
   stock library rejects unknown ids (encode.c:121). */
#define BROTLI_AMD_PARAM_SHARD_BYTES 0x4D490001u

/* encode.h:306  BrotliEncoderCreateInstance */
BrotliEncoderState* BrotliEncoderCreateInstance(brotli_amd_alloc_func alloc_func,
                                                brotli_amd_free_func free_func, void* opaque);
/* encode.h:314  BrotliEncoderDestroyInstance */
void BrotliEncoderDestroyInstance(BrotliEncoderState* state);
/* encode.h:289  BrotliEncoderSetParameter (ids encode.h:160-265; only before the
   first compress call, encode.c:63) */
BROTLI_BOOL BrotliEncoderSetParameter(BrotliEncoderState* state, int param, uint32_t value);
/* encode.h:375  BrotliEncoderMaxCompressedSize */
size_t BrotliEncoderMaxCompressedSize(size_t input_size);
/* encode.h:405  BrotliEncoderCompress (one shot; SIZE_HINT = input_size,
   encode.c:1331; empty input -> the single byte 0x06, encode.c:1310-1314) */
BROTLI_BOOL BrotliEncoderCompress(int quality, int lgwin, int mode, size_t input_size,
                                  const uint8_t* input_buffer, size_t* encoded_size,
                                  uint8_t* encoded_buffer);
/* encode.h:473  BrotliEncoderCompressStream (op: 0 PROCESS, 1 FLUSH, 2 FINISH,
   3 EMIT_METADATA; encode.h:99-156) */
BROTLI_BOOL BrotliEncoderCompressStream(BrotliEncoderState* state, int op, size_t* available_in,
                                        const uint8_t** next_in, size_t* available_out,
                                        uint8_t** next_out, size_t* total_out);
/* encode.h:486  BrotliEncoderIsFinished */
BROTLI_BOOL BrotliEncoderIsFinished(BrotliEncoderState* state);
/* encode.h:495  BrotliEncoderHasMoreOutput */
BROTLI_BOOL BrotliEncoderHasMoreOutput(BrotliEncoderState* state);
/* encode.h:526  BrotliEncoderTakeOutput */
const uint8_t* BrotliEncoderTakeOutput(BrotliEncoderState* state, size_t* size);
/* encode.h:542  BrotliEncoderVersion */
uint32_t BrotliEncoderVersion(void);
/* encode.h:531  BrotliEncoderEstimatePeakMemoryUsage (host memory of one instance; the encoder
   state itself is device memory) */
size_t BrotliEncoderEstimatePeakMemoryUsage(int quality, int lgwin, size_t input_size);
/* encode.h:534  BrotliEncoderGetPreparedDictionarySize: host bytes of the index this library built
   (the raw dictionary bytes stay the caller's), 0 = not a dictionary of this library */
size_t BrotliEncoderGetPreparedDictionarySize(const BrotliEncoderPreparedDictionary* dictionary);
/* encode.h:342 / 348 / 361 (c/enc/encode.c:1756-1880, c/enc/compound_dictionary.c): raw LZ77-prefix
   dictionaries (type BROTLI_SHARED_DICTIONARY_RAW = 0), up to 15 per encoder instance.  The caller
   keeps the dictionary bytes alive while a prepared dictionary exists, and the prepared dictionary
   alive while an encoder uses it — as with the reference.  Any other type returns NULL (the
   reference's serialized form is an experimental build option).  Attached dictionaries are looked
   up on the device at qualities 2 - 9 — by the single stream, and by every shard of a partition plan
   (BROTLI_AMD_SHARD_KB: each shard's instance has them attached, as the reference driven with the same
   plan would); qualities 0 - 1 ignore them. */
BrotliEncoderPreparedDictionary* BrotliEncoderPrepareDictionary(
    int type, size_t data_size, const uint8_t* data, int quality,
    brotli_amd_alloc_func alloc_func, brotli_amd_free_func free_func, void* opaque);
void BrotliEncoderDestroyPreparedDictionary(BrotliEncoderPreparedDictionary* dictionary);
BROTLI_BOOL BrotliEncoderAttachPreparedDictionary(BrotliEncoderState* state,
                                                  const BrotliEncoderPreparedDictionary* dictionary);

#ifdef __cplusplus
}
#endif
#endif  /* BROTLI_AMD_ENCODE_H_ */
