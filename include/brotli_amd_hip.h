/* include/brotli_amd_hip.h — the thin HIP C-ABI layer under the Brotli encoder
 * boundary (include/brotli_amd_encode.h).
 *
 * Plain C: opaque handles, raw pointers, sizes.  Device pointers are ordinary
 * HIP allocations of the calling process (hipMalloc, or a torch tensor's
 * data_ptr()).  The layer owns one HIP stream per context, is re-entrant across
 * contexts (no process-wide state; SURVEY.md §8b "Threading") and never falls
 * back to a CPU encoder: a parameter combination the kernels do not implement
 * returns BROTLI_AMD_UNSUPPORTED.
 *
 * What it replaces in the reference: the body of EncodeData ->
 * BrotliCreateBackwardReferences -> WriteMetaBlockInternal
 * (c/enc/encode.c:985-1221, c/enc/backward_references.c:251-299,
 * c/enc/encode.c:498-614) for every shard of a partition plan
 * (BROTLI_PARAM_STREAM_OFFSET contract, c/include/brotli/encode.h:231-246).
 */
#ifndef BROTLI_AMD_HIP_H_
#define BROTLI_AMD_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BROTLI_AMD_OK 0
#define BROTLI_AMD_ERROR (-1)        /* HIP failure, see brotli_amd_last_error */
#define BROTLI_AMD_UNSUPPORTED (-2)  /* parameters outside the GPU path */
#define BROTLI_AMD_OVERFLOW (-3)     /* output capacity too small */
#define BROTLI_AMD_SERIAL (-5)       /* BROTLI_AMD_FLAG_STREAM_TILES: the stream has to go through brotli_amd_stream_* */
#define BROTLI_AMD_DEVICE_FAULT (-4) /* a shard reported an internal error */

/* Device input buffers must stay readable this many bytes past `len`
   (the kernels load 32-byte strings at the last positions; the values read
   there never influence the result). */
#define BROTLI_AMD_INPUT_SLACK 64

typedef struct BrotliAmdCtx BrotliAmdCtx;

/* One job = one contiguous piece of one Brotli stream handled by one GPU. */
typedef struct BrotliAmdJobParams {
  int32_t quality;       /* BROTLI_PARAM_QUALITY (encode.h:175) */
  int32_t lgwin;         /* BROTLI_PARAM_LGWIN (encode.h:185) */
  uint32_t size_hint;    /* BROTLI_PARAM_SIZE_HINT of the WHOLE stream, capped
                            at 1<<30 (encode.h:209, encode.c:1619-1632);
                            0 = min(stream_base + len, 1<<30) */
  uint32_t flags;        /* BROTLI_AMD_FLAG_* */
  uint64_t shard_size;   /* partition plan: bytes per encoder shard; 0 = one
                            shard (== BrotliEncoderCompress on the buffer) */
  uint64_t stream_base;  /* offset of this buffer inside the whole stream
                            (multi-GPU: rank r passes the offset of its first
                            shard); shard k starts at stream_base + k * shard_size */
  int32_t is_last;       /* 1: the buffer ends the stream (last shard FINISHes);
                            0: every shard ends with FLUSH */
  int32_t reserved;
} BrotliAmdJobParams;

#define BROTLI_AMD_FLAG_LGBLOCK_SHIFT 24 /* bits 24..28 of `flags` (jobs and streams): BROTLI_PARAM_LGBLOCK (encode.h:190-197), 0 = the
                                           default of the quality; looked at from quality 4 on and clamped to 16 .. 24 as the
                                           reference does (quality.h:75-92).  A tiled stream (BROTLI_AMD_FLAG_STREAM_TILES)
                                           needs the default */
#define BROTLI_AMD_FLAG_LGBLOCK(lg) (((uint32_t)(lg) & 31u) << BROTLI_AMD_FLAG_LGBLOCK_SHIFT)
#define BROTLI_AMD_FLAG_NO_PAIR 1u    /* debugging: no speculative (p,p+1) search (k_parse) */
#define BROTLI_AMD_FLAG_NO_QUAD 2u    /* always one shard per wave (k_parse) */
#define BROTLI_AMD_FLAG_FORCE_SLOW 4u /* k_parse4: step-by-step candidate resolve */
#define BROTLI_AMD_FLAG_NO_HEADER 8u  /* the stream header (window bits) has already been
                                         written by the caller (empty FLUSH at stream start,
                                         encode.c:1356-1415): the first shard starts byte aligned */
#define BROTLI_AMD_FLAG_NO_LITERAL_CONTEXT 32u /* BROTLI_PARAM_DISABLE_LITERAL_CONTEXT_MODELING: one literal context
                                        (encode.c:561; qualities below 5 have one anyway) */
#define BROTLI_AMD_FLAG_STREAM_TILES 64u /* quality 5, shard_size 0, the whole stream in this one call (stream_base 0, is_last):
                                          parse it in tiles although it is longer than the window (k_tile.h, JOB_FLAG_STREAMT).
                                          BROTLI_AMD_SERIAL comes back when the stream turns out not to suit the tiles (static
                                          dictionary still consulted, a meta-block stored raw ...): nothing was written, the
                                          caller runs it through brotli_amd_stream_* instead */
#define BROTLI_AMD_FLAG_TAIL_FINISH 128u /* with BROTLI_AMD_FLAG_STREAM_TILES through brotli_amd_encode_host: the stream reached the caller's
                                           encoder in PROCESS calls ending on an input-block boundary, the FINISH came empty
                                           (encode.c:1700-1712) — the last meta-block is closed as the reference closes it then;
                                           brotli_amd_encode_device refuses the flag (BROTLI_AMD_UNSUPPORTED): half of it is host work */
#define BROTLI_AMD_FLAG_NO_INDEX 16u  /* quality 5: hash-table parse (k_parse4) instead of the position index
                                         (k_index.h + k_chain.h) */

typedef struct BrotliAmdJobInfo {
  uint64_t nshards;
  uint64_t out_bytes;
  uint64_t ws_bytes;        /* HBM workspace used by the job */
  uint32_t rounds;          /* parse/build/store rounds (1 unless a shard holds
                               several meta-blocks) */
  uint32_t reserved;
  float ms_total;           /* HIP-event time of the whole job on the stream */
  float ms_init, ms_parse, ms_build, ms_store, ms_gather;
  float ms_index;           /* indexed quality-5 job: the index kernels (k_index.h); ms_parse is the chain */
  float ms_ix_bucket;       /* of ms_index: k_ix_bucket alone (sort inside the buckets + window search of every position) */
  uint64_t searches;        /* FindLongestMatch calls (reference count) */
  uint64_t search_steps;    /* paired search steps actually executed */
  uint64_t commands;
  uint64_t exact_searches;  /* indexed job: searches the chain had to redo itself (k_chain.h) */
  uint64_t prof[12];        /* debug_parse with a -DQ_PROFILE build: cycles per phase */
} BrotliAmdJobInfo;

/* `tables_path`: brotli_amd/data/brotli_tables.bin (RFC 7932 format data). */
int brotli_amd_ctx_create(int device, const char* tables_path, BrotliAmdCtx** ctx);
/* The BROTLI_AMD_* experiment knobs of the device layer (tile size, index / chain layouts, logging ...) are read from
   the environment when a context is created, not on the launch path of a job; a tool or test that changes one between
   two jobs of a living context calls this to have them read again.  Not for use beside running jobs. */
void brotli_amd_refresh_env(void);
void brotli_amd_ctx_destroy(BrotliAmdCtx* ctx);
const char* brotli_amd_last_error(const BrotliAmdCtx* ctx);

/* Upper bound of the bytes a job can produce (sum over shards of the
   reference's 2*n+503 storage bound, encode.c:1187, plus flush padding). */
uint64_t brotli_amd_max_output(uint64_t len, const BrotliAmdJobParams* p);

/* Encodes d_in[0,len) (device memory, BROTLI_AMD_INPUT_SLACK readable past
   the end) into d_out (device memory, out_cap bytes): the concatenation of
   the shards' outputs in order.  d_shard_sizes (device, may be NULL) receives
   nshards u64 compressed sizes.  Synchronous on return.  The work runs on a stream of the
   context's own, created non-blocking: it is NOT ordered behind work the caller has in flight
   on other streams — whatever produces d_in must have completed (brotli_amd/hip.py
   synchronises torch's current stream before every call). */
int brotli_amd_encode_device(BrotliAmdCtx* ctx, const void* d_in, uint64_t len,
                             const BrotliAmdJobParams* p, void* d_out,
                             uint64_t out_cap, uint64_t* out_size,
                             uint64_t* d_shard_sizes, BrotliAmdJobInfo* info);

/* Same with host buffers (H2D + job + D2H on the context's stream). */
int brotli_amd_encode_host(BrotliAmdCtx* ctx, const uint8_t* in, uint64_t len,
                           const BrotliAmdJobParams* p, uint8_t* out,
                           uint64_t out_cap, uint64_t* out_size,
                           BrotliAmdJobInfo* info);

/* ---- one encoder instance fed incrementally (BrotliEncoderCompressStream) ----
   A stream is ONE shard whose state (hash table, ring positions, distance
   cache, partial last byte) stays on the device between calls, so any sequence
   of PROCESS / FLUSH / FINISH operations yields the reference's bytes
   (c/enc/encode.c:1634-1722).  `stream_offset` as BROTLI_PARAM_STREAM_OFFSET. */
typedef struct BrotliAmdStream BrotliAmdStream;
#define BROTLI_AMD_OP_PROCESS 0
#define BROTLI_AMD_OP_FLUSH 1
#define BROTLI_AMD_OP_FINISH 2
#define BROTLI_AMD_OP_FLUSH_OPEN 3   /* flush the pending input as a meta-block, keep the partial last
                                        byte pending (what EMIT_METADATA needs, encode.c:1569-1573) */
/* `flags`: BROTLI_AMD_FLAG_NO_HEADER when the stream header has left the encoder already (an empty
   FLUSH or a metadata block came before the first data, encode.c:1356-1415): the stream then starts
   byte aligned without the window bits. */
int brotli_amd_stream_create(BrotliAmdCtx* ctx, int quality, int lgwin, uint32_t size_hint,
                             uint32_t stream_offset, uint32_t flags, BrotliAmdStream** stream);
/* Appends `len` host bytes and applies `op`.  `*out` / `*out_len` receive the
   bytes produced by this call; the pointer stays valid until the next call on
   the stream. */
int brotli_amd_stream_write(BrotliAmdStream* stream, const uint8_t* data, uint64_t len, int op,
                            const uint8_t** out, uint64_t* out_len);
/* Attached dictionaries of the stream (BrotliEncoderAttachPreparedDictionary, c/enc/encode.c:1828-1880,
   raw LZ77 prefixes only): chunk d is its bytes plus the index the device lookup walks —
   starts[key] .. starts[key + 1] are the positions (newest first, at most 32) whose eight bytes hash
   to `key` (brotli_amd/csrc/dict_index.h builds it; brotli_amd/csrc/k_dict.h reads it).  All
   pointers are host memory; everything is copied to the device by the call.  The call REPLACES the
   stream's dictionary list (pass all chunks attached so far, in attach order, at most 15); it takes
   effect with the next input block, as in the reference.  Every hasher that has a dictionary variant
   in the reference looks the chunks up (H3 - H6, H40 - H42, H58, H68); H2 and H54 only shift their
   distances by the dictionary size (backward_references.c:194-243, 256-282). */
typedef struct BrotliAmdDictChunk {
  const uint8_t* source;
  const uint32_t* starts;
  const uint32_t* items;
  uint32_t source_size;
  uint32_t bucket_bits;
} BrotliAmdDictChunk;
int brotli_amd_stream_attach_dictionary(BrotliAmdStream* stream, const BrotliAmdDictChunk* chunks,
                                        uint32_t nchunks);
/* The same for the JOBS of a context (brotli_amd_encode_device / _host with a partition plan): every
   shard's encoder instance has the chunks attached — bytes identical to the reference driven with the
   same plan and BrotliEncoderAttachPreparedDictionary on every instance; the concatenation decodes
   with the dictionary attached once.  Jobs with a dictionary run one shard per wave on the hash-table
   kernels (the indexed quality-5 parse has no dictionary lookup).  nchunks = 0 detaches; a context
   taken from a pool should be cleared by its new user. */
int brotli_amd_ctx_set_dictionary(BrotliAmdCtx* ctx, const BrotliAmdDictChunk* chunks, uint32_t nchunks);
/* Hands the pending partial byte (s->last_bytes_ / last_bytes_bits_) to the caller and
   clears it on the device: the caller continues the byte (metadata header). */
int brotli_amd_stream_take_partial(BrotliAmdStream* stream, uint32_t* nbits, uint32_t* value);
void brotli_amd_stream_destroy(BrotliAmdStream* stream);

/* ---- quality 1: the two-pass fragment compressor (k_fast.h) -------------------
   Replaces BrotliCompressFragmentTwoPass as driven by
   BrotliEncoderCompressStreamFast (c/enc/encode.c:1425-1547,
   c/enc/compress_fragment_two_pass.c:564-641) for one run of calls between
   flushes: call k hands call_sizes[k] bytes and is cut into fragments of
   min(1 << lgwin, bytes left in the call), exactly as the reference does.
   The output starts with `carry_bits` pending bits `carry_value` (the stream
   header, or the partial byte a previous run left, s->last_bytes_) and is
   *out_bits bits long; when is_last the ISLAST / ISLASTEMPTY bits and the
   final padding are included and *out_bits is a multiple of 8. */
typedef struct BrotliAmdFastParams {
  int32_t lgwin;          /* 10 .. 24 */
  uint32_t carry_bits;    /* 0 .. 15 */
  uint32_t carry_value;
  int32_t is_last;
} BrotliAmdFastParams;
uint64_t brotli_amd_fast_max_output(uint64_t len, uint64_t ncalls, int lgwin);
int brotli_amd_encode_fast_device(BrotliAmdCtx* ctx, const void* d_in, uint64_t len,
                                  const uint64_t* call_sizes, uint64_t ncalls,
                                  const BrotliAmdFastParams* p, void* d_out,
                                  uint64_t out_cap, uint64_t* out_bits,
                                  BrotliAmdJobInfo* info);
int brotli_amd_encode_fast_host(BrotliAmdCtx* ctx, const uint8_t* in, uint64_t len,
                                const uint64_t* call_sizes, uint64_t ncalls,
                                const BrotliAmdFastParams* p, uint8_t* out,
                                uint64_t out_cap, uint64_t* out_bits,
                                BrotliAmdJobInfo* info);

/* ---- decoder on the device (k_decode.h) -----------------------------
   What it is for: round trips of what the encoder produced, at rate, with the data staying in HBM
   (tests, bench.py's round-trip check).  It decodes any RFC 7932 stream (no large window); the
   counterpart in the reference is BrotliDecoderDecompress (c/dec/decode.c), a serial state machine.
   The unit of work is a PIECE, decoded by one wave: a whole stream, or one shard of a partition
   plan — shard k of the encoder's output starts at the sum of the compressed sizes before it
   (brotli_amd_encode_device's d_shard_sizes), decodes to stream offset k * shard_size, and only
   shard 0 carries the stream header.  Pieces of one call decode concurrently. */
typedef struct BrotliAmdDecodePiece {
  uint64_t in_off, in_len;   /* the piece's compressed bytes inside the input buffer */
  uint64_t out_off;          /* where its bytes go in the output buffer == its offset in the stream */
  uint64_t out_cap;          /* room there (the shard size; the whole capacity for one stream) */
  uint32_t flags;            /* BROTLI_AMD_PIECE_* */
  uint32_t lgwin;            /* the stream's window, for pieces without BROTLI_AMD_PIECE_HEADER */
} BrotliAmdDecodePiece;
#define BROTLI_AMD_PIECE_HEADER 1u     /* starts with the stream header (window bits) */
#define BROTLI_AMD_PIECE_ISOLATED 2u   /* a shard of a plan: a copy reaching before the piece is an error */
typedef struct BrotliAmdDecodeResult {
  uint64_t out_bytes;        /* bytes written; with error != 0 the last of them are not to be trusted (a cut
                                input is noticed a few symbols late) */
  uint64_t in_bits;          /* bits consumed */
  uint32_t error;            /* 0 = ok; 1 header, 2 prefix code, 3 context map, 4 distance, 5 dictionary,
                                6 output overrun, 7 input overrun, 8 arena (too many prefix codes for the
                                per-piece workspace), 9 unsupported (large window) */
  uint32_t finished;         /* the ISLAST meta-block was decoded */
  uint32_t lgwin;
  uint32_t metablocks;
} BrotliAmdDecodeResult;
/* Device input buffers of the decoder must stay readable this many bytes past their end (a damaged
   stream is noticed at most one prefix code late). */
#define BROTLI_AMD_DECODE_SLACK 4096
/* d_in / d_out: device memory.  pieces / results: host arrays of npieces entries.  Returns
   BROTLI_AMD_OK when the kernel ran (look at results[k].error per piece), BROTLI_AMD_DEVICE_FAULT if
   any piece reported an error, BROTLI_AMD_ERROR for HIP failures.  *ms (may be NULL): HIP-event time
   of the kernel. */
int brotli_amd_decode_device(BrotliAmdCtx* ctx, const void* d_in, uint64_t in_len,
                             const BrotliAmdDecodePiece* pieces, uint64_t npieces, void* d_out,
                             uint64_t out_cap, BrotliAmdDecodeResult* results, float* ms);
/* Same with host buffers. */
int brotli_amd_decode_host(BrotliAmdCtx* ctx, const uint8_t* in, uint64_t in_len,
                           const BrotliAmdDecodePiece* pieces, uint64_t npieces, uint8_t* out,
                           uint64_t out_cap, BrotliAmdDecodeResult* results, float* ms);

/* Parity tap used by tests/: runs table init + the LZ77 parse only and copies
   the command list of the first meta-block of every shard (16-byte records,
   c/enc/command.h:106-116 field order) to host memory. */
int brotli_amd_debug_parse(BrotliAmdCtx* ctx, const void* d_in, uint64_t len,
                           const BrotliAmdJobParams* p, void* h_cmds,
                           uint64_t cmd_cap, uint64_t* ncmds,
                           BrotliAmdJobInfo* info);

#ifdef __cplusplus
}
#endif
#endif  /* BROTLI_AMD_HIP_H_ */
