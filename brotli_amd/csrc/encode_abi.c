/* brotli_amd/csrc/encode_abi.c — host side of the drop-in boundary
 * (include/brotli_amd_encode.h): the google/brotli encoder C ABI
 * (c/include/brotli/encode.h, 13 symbols) implemented over the HIP C-ABI layer
 * (include/brotli_amd_hip.h).  Plain C; built into
 * brotli_amd/lib/libbrotlienc_amd.so.
 *
 * What is re-stated here from c/enc/encode.c is only the *stream protocol*:
 * parameter latching (encode.c:60-123), the size-hint rule (:1619-1632), the
 * PROCESS / FLUSH / FINISH state machine and output hand-off (:1393-1423,
 * :1634-1750), the one-shot wrapper and its raw fallback (:1251-1354).  All
 * compression work happens on the GPU; unsupported parameters fail with
 * BROTLI_FALSE (there is no CPU encoder behind this library).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>

#include "../../include/brotli_amd_encode.h"
#include "../../include/brotli_amd_hip.h"
#include "dict_index.h"

enum { OP_PROCESS = 0, OP_FLUSH = 1, OP_FINISH = 2, OP_EMIT_METADATA = 3 };
enum { ST_PROCESSING = 0, ST_FLUSH_REQUESTED = 1, ST_FINISHED = 2, ST_METADATA = 3 };
enum {  /* BrotliEncoderParameter, encode.h:160-265 */
  P_MODE = 0, P_QUALITY = 1, P_LGWIN = 2, P_LGBLOCK = 3, P_DISABLE_CTX = 4, P_SIZE_HINT = 5,
  P_LARGE_WINDOW = 6, P_NPOSTFIX = 7, P_NDIRECT = 8, P_STREAM_OFFSET = 9, P_BASE64_MODE = 10,
  P_MAX_BASE64_REGIONS = 11, P_SIMD_HASHER = 12
};

struct BrotliEncoderStateStruct {
  brotli_amd_alloc_func alloc;
  brotli_amd_free_func free_;
  void* opaque;
  /* parameters as set by the caller */
  int mode, quality, lgwin, lgblock, disable_ctx, large_window, base64_mode, simd_hasher;
  uint32_t size_hint, npostfix, ndirect, stream_offset, max_base64_regions;
  uint64_t shard_bytes;
  int device;
  /* latched at first use */
  int initialized, failed;
  uint32_t eff_hint;
  int hint_fixed;
  /* input not yet handed to the device */
  uint8_t* in_buf;
  size_t in_len, in_cap;
  size_t in_mapped;          /* != 0: in_buf is an anonymous mapping of that many bytes (grow_in), not an allocation */
  uint64_t total_in, submitted;
  /* produced, not yet taken */
  /* one FLUSH / FINISH call that brings its whole input and has room for the output: the job
     reads the caller's input and writes the caller's output (no staging copies on the host) */
  int tail_finish;       /* this FINISH brought no input and what PROCESS calls brought before ends on an input-block boundary */
  int empty_finish;      /* this FINISH brought no input while bytes of earlier PROCESS calls are still held here */
  /* the boundary's BROTLI_AMD_* knobs, read once when the instance is created (not with getenv() per call) */
  long env_feed_kb;      /* BROTLI_AMD_FEED_KB, -1: not set */
  long env_hold_mb;      /* BROTLI_AMD_HOLD_MB, -1: not set */
  int env_stream_tiles;  /* BROTLI_AMD_STREAM_TILES, -1: not set */
  int env_verbose;       /* BROTLI_AMD_VERBOSE set */
  uint8_t* direct_out;
  size_t direct_cap, direct_n;
  uint8_t* out_buf;
  size_t out_len, out_pos, out_cap;
  uint64_t total_out;
  int stream_state;
  int header_written;   /* plan mode: the stream header left the library */
  /* quality 1 (BrotliEncoderCompressStreamFast, encode.c:1425-1547): the sizes of the
     calls buffered since the last flush (they decide the fragment boundaries) and the
     pending partial byte (s->last_bytes_ / last_bytes_bits_) */
  uint64_t* calls;
  size_t ncalls, calls_cap;
  uint32_t carry_bits, carry_value;
  BrotliAmdCtx* ctx;
  BrotliAmdStream* stream;
  /* attached raw dictionaries in attach order (params.dictionary.compound, encode.c:1828-1850) */
  const DictIndex* dicts[15];
  uint32_t ndicts;
  uint32_t ctx_dicts;      /* how many of them the context's jobs already carry (plan mode) */
  uint64_t dict_total;
};

static void* st_alloc(BrotliEncoderState* s, size_t n) {
  return s->alloc ? s->alloc(s->opaque, n) : malloc(n);
}
static void st_free(BrotliEncoderState* s, void* p) {
  if (!p) return;
  if (s->free_) s->free_(s->opaque, p); else free(p);
}
static int grow(BrotliEncoderState* s, uint8_t** buf, size_t* cap, size_t used, size_t need) {
  uint8_t* n;
  size_t c = *cap ? *cap : 65536;
  if (need <= *cap) return 1;
  while (c < need) c *= 2;
  n = (uint8_t*)st_alloc(s, c);
  if (!n) return 0;
  if (used) memcpy(n, *buf, used);
  st_free(s, *buf);
  *buf = n;
  *cap = c;
  return 1;
}

/* The held input (feed_threshold: a stream fed by PROCESS calls is kept until FINISH): from 4 MiB on an anonymous
   mapping of its own — grown in place by mremap, with transparent huge pages asked for — instead of allocations that
   double: those copied the held bytes again at every doubling and took a page fault per 4 KiB of every new buffer,
   which was most of the time of a PROCESS-fed 256 MiB (0.42 s against 0.09 s for the same bytes in one call).  Custom
   allocation functions are honoured as before. */
static void free_in(BrotliEncoderState* s) {
  if (s->in_mapped) munmap(s->in_buf, s->in_mapped); else st_free(s, s->in_buf);
  s->in_buf = NULL;
  s->in_cap = s->in_mapped = 0;
}
static int grow_in(BrotliEncoderState* s, size_t need) {
  size_t c;
  void* n;
  if (need <= s->in_cap) return 1;
  if (s->alloc || need < ((size_t)4 << 20)) return grow(s, &s->in_buf, &s->in_cap, s->in_len, need);
  c = s->in_cap > ((size_t)4 << 20) ? s->in_cap : ((size_t)4 << 20);
  if (s->size_hint > c && need <= (size_t)s->size_hint + 65536) c = (size_t)s->size_hint + 65536;   /* announced: once */
  while (c < need) c *= 2;
  c = (c + (((size_t)2 << 20) - 1)) & ~(((size_t)2 << 20) - 1);
  if (s->in_mapped) {
    n = mremap(s->in_buf, s->in_mapped, c, MREMAP_MAYMOVE);
    if (n == MAP_FAILED) return 0;
  } else {
    n = mmap(NULL, c, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (n == MAP_FAILED) return 0;
    if (s->in_len) memcpy(n, s->in_buf, s->in_len);
    st_free(s, s->in_buf);
  }
#ifdef MADV_HUGEPAGE
  (void)madvise(n, c, MADV_HUGEPAGE);
#endif
  s->in_buf = (uint8_t*)n;
  s->in_cap = s->in_mapped = c;
  return 1;
}

static void tables_path(char* out, size_t cap) {
  const char* env = getenv("BROTLI_AMD_TABLES");
  Dl_info info;
  if (env) { snprintf(out, cap, "%s", env); return; }
  if (dladdr((void*)&tables_path, &info) && info.dli_fname) {
    char dir[4096];
    char* slash;
    /* the library may have been found through the dropin/ symlink */
    if (!realpath(info.dli_fname, dir)) snprintf(dir, sizeof(dir), "%s", info.dli_fname);
    slash = strrchr(dir, '/');
    if (slash) *slash = 0; else snprintf(dir, sizeof(dir), ".");
    snprintf(out, cap, "%s/../data/brotli_tables.bin", dir);
    return;
  }
  snprintf(out, cap, "brotli_tables.bin");
}

/* Device contexts are expensive (HIP context, tables, a workspace sized for the largest job so
   far): an encoder instance borrows one from a small process-wide pool and hands it back when it
   is destroyed, so a program that compresses many buffers pays for the set-up once. */
#define CTX_POOL 32
static pthread_mutex_t g_pool_lock = PTHREAD_MUTEX_INITIALIZER;
static struct { BrotliAmdCtx* ctx; int device; } g_pool[CTX_POOL];

static BrotliAmdCtx* pool_take(int device) {
  BrotliAmdCtx* c = NULL;
  int i;
  pthread_mutex_lock(&g_pool_lock);
  for (i = 0; i < CTX_POOL; ++i) {
    if (g_pool[i].ctx && g_pool[i].device == device) { c = g_pool[i].ctx; g_pool[i].ctx = NULL; break; }
  }
  pthread_mutex_unlock(&g_pool_lock);
  return c;
}
static void pool_give(BrotliAmdCtx* c, int device) {
  int i;
  pthread_mutex_lock(&g_pool_lock);
  for (i = 0; i < CTX_POOL; ++i) {
    if (!g_pool[i].ctx) { g_pool[i].ctx = c; g_pool[i].device = device; c = NULL; break; }
  }
  pthread_mutex_unlock(&g_pool_lock);
  if (c) brotli_amd_ctx_destroy(c);
}

BrotliEncoderState* BrotliEncoderCreateInstance(brotli_amd_alloc_func alloc_func,
                                                brotli_amd_free_func free_func, void* opaque) {
  BrotliEncoderState* s;
  const char* e;
  if ((alloc_func == NULL) != (free_func == NULL)) return NULL;   /* encode.h:295-299 */
  s = (BrotliEncoderState*)(alloc_func ? alloc_func(opaque, sizeof(*s)) : malloc(sizeof(*s)));
  if (!s) return NULL;
  memset(s, 0, sizeof(*s));
  s->alloc = alloc_func;
  s->free_ = free_func;
  s->opaque = opaque;
  s->quality = 11;   /* BROTLI_DEFAULT_QUALITY */
  s->lgwin = 22;     /* BROTLI_DEFAULT_WINDOW */
  s->max_base64_regions = 16;
  e = getenv("BROTLI_AMD_SHARD_KB");
  if (e) s->shard_bytes = strtoull(e, NULL, 10) << 10;
  e = getenv("BROTLI_AMD_DEVICE");
  if (e) s->device = atoi(e);
  e = getenv("BROTLI_AMD_FEED_KB");
  s->env_feed_kb = e ? (long)strtoull(e, NULL, 10) : -1;
  e = getenv("BROTLI_AMD_HOLD_MB");
  s->env_hold_mb = e ? (long)strtoull(e, NULL, 10) : -1;
  e = getenv("BROTLI_AMD_STREAM_TILES");
  s->env_stream_tiles = e ? atoi(e) : -1;
  s->env_verbose = getenv("BROTLI_AMD_VERBOSE") != NULL;
  return s;
}

void BrotliEncoderDestroyInstance(BrotliEncoderState* s) {
  if (!s) return;
  if (s->stream) brotli_amd_stream_destroy(s->stream);
  if (s->ctx) {
    if (s->failed) brotli_amd_ctx_destroy(s->ctx);   /* whatever went wrong, do not pass it on */
    else {
      if (s->ctx_dicts) (void)brotli_amd_ctx_set_dictionary(s->ctx, NULL, 0);
      pool_give(s->ctx, s->device);
    }
  }
  free_in(s);
  st_free(s, s->out_buf);
  st_free(s, s->calls);
  st_free(s, s);
}

BROTLI_BOOL BrotliEncoderSetParameter(BrotliEncoderState* s, int p, uint32_t value) {
  if (s->initialized) return BROTLI_FALSE;   /* encode.c:63 */
  switch ((uint32_t)p) {
    case P_MODE: s->mode = (int)value; return BROTLI_TRUE;
    case P_QUALITY: s->quality = (int)value; return BROTLI_TRUE;
    case P_LGWIN: s->lgwin = (int)value; return BROTLI_TRUE;
    case P_LGBLOCK: s->lgblock = (int)value; return BROTLI_TRUE;
    case P_DISABLE_CTX:
      if (value != 0 && value != 1) return BROTLI_FALSE;
      s->disable_ctx = (int)value;
      return BROTLI_TRUE;
    case P_SIZE_HINT: s->size_hint = value; return BROTLI_TRUE;
    case P_LARGE_WINDOW: s->large_window = value != 0; return BROTLI_TRUE;
    case P_NPOSTFIX: s->npostfix = value; return BROTLI_TRUE;
    case P_NDIRECT: s->ndirect = value; return BROTLI_TRUE;
    case P_STREAM_OFFSET:
      if (value > (1u << 30)) return BROTLI_FALSE;
      s->stream_offset = value;
      return BROTLI_TRUE;
    case P_BASE64_MODE: s->base64_mode = (int)(value & 1); return BROTLI_TRUE;
    case P_MAX_BASE64_REGIONS: s->max_base64_regions = value; return BROTLI_TRUE;
    case P_SIMD_HASHER:
      if (value > 2) return BROTLI_FALSE;
      s->simd_hasher = (int)value;
      return BROTLI_TRUE;
    case BROTLI_AMD_PARAM_SHARD_BYTES: s->shard_bytes = value; return BROTLI_TRUE;
    default: return BROTLI_FALSE;   /* encode.c:121 */
  }
}

static void window_bits(int lgwin, uint32_t* bits, uint32_t* nbits);

static int ensure_initialized(BrotliEncoderState* s) {
  char path[4200];
  if (s->initialized) return !s->failed;
  s->initialized = 1;
  /* SanitizeParams, c/enc/quality.h:59-73 */
  if (s->quality < 0) s->quality = 0;
  if (s->quality > 11) s->quality = 11;
  if (s->lgwin < 10) s->lgwin = 10;
  if (s->lgwin > 24 && !s->large_window) s->lgwin = 24;
  /* What the kernels implement: qualities 2..4 (H2 / H3 / H4 / H54) and 5..9 (H68 / H58 / H6 / H5; H40 / H41 / H42 at
     windows of 10..16 bits), any block size (BROTLI_PARAM_LGBLOCK), default distance parameters, no base64 regions,
     literal context modelling on, the x86-64 default hashers. */
  if (s->quality == 1) {
    /* The two-pass fragment compressor ignores mode, block size, distance
       parameters, context modelling, size hint and hasher selection
       (encode.c:1660-1664 branches before any of them is read). */
    if (s->large_window || s->shard_bytes != 0) s->failed = 1;
  } else if (s->quality < 2 || s->quality > 9 || s->lgwin > 24 || s->large_window ||
      s->mode == 2 /* FONT: non-zero distance parameters, encode.c:616-640 */ ||
      s->npostfix != 0 || s->ndirect != 0 || s->base64_mode != 0 ||
      s->simd_hasher != 0 /* ENABLE / DISABLE change the hasher choice at q5-q7 */) {
    s->failed = 1;
  }
  if (s->failed) {
    if (s->env_verbose)
      fprintf(stderr, "brotli_amd: parameters outside the GPU path (quality %d, lgwin %d); "
                      "no CPU fallback exists\n", s->quality, s->lgwin);
    return 0;
  }
  tables_path(path, sizeof(path));
  s->ctx = pool_take(s->device);
  if (!s->ctx && brotli_amd_ctx_create(s->device, path, &s->ctx) != BROTLI_AMD_OK) {
    s->failed = 1;
    if (s->env_verbose) fprintf(stderr, "brotli_amd: %s\n", brotli_amd_last_error(s->ctx));
    return 0;
  }
  /* a pooled context may still carry its previous user's dictionaries */
  if (brotli_amd_ctx_set_dictionary(s->ctx, NULL, 0) != BROTLI_AMD_OK) { s->failed = 1; return 0; }
  s->ctx_dicts = 0;
  if (s->size_hint != 0) { s->eff_hint = s->size_hint; s->hint_fixed = 1; }
  if (s->quality == 1 && s->stream_offset == 0) {
    /* the stream header waits in the partial byte; quality 0/1 announce at least
       an 18-bit window (encode.c:668-677) */
    window_bits(s->lgwin < 18 ? 18 : s->lgwin, &s->carry_value, &s->carry_bits);
  }
  return 1;
}

/* EncodeWindowBits, encode.c:191-211 (no large window). */
static void window_bits(int lgwin, uint32_t* bits, uint32_t* nbits) {
  if (lgwin == 16) { *bits = 0; *nbits = 1; }
  else if (lgwin == 17) { *bits = 1; *nbits = 7; }
  else if (lgwin > 17) { *bits = (uint32_t)(((lgwin - 17) << 1) | 1); *nbits = 4; }
  else { *bits = (uint32_t)(((lgwin - 8) << 4) | 1); *nbits = 7; }
}

static int out_append(BrotliEncoderState* s, const uint8_t* p, size_t n) {
  if (s->out_pos == s->out_len) s->out_pos = s->out_len = 0;
  if (!grow(s, &s->out_buf, &s->out_cap, s->out_len, s->out_len + n)) return 0;
  memcpy(s->out_buf + s->out_len, p, n);
  s->out_len += n;
  return 1;
}

/* The device stream of one encoder instance, with the dictionaries attached so far. */
static int push_dictionaries_to(BrotliEncoderState* s, int to_context);
static int push_dictionaries(BrotliEncoderState* s) { return push_dictionaries_to(s, 0); }
/* Plan mode: the jobs of the context carry the dictionaries (every shard's instance has them attached). */
static int sync_context_dictionaries(BrotliEncoderState* s) {
  if (s->ctx_dicts == s->ndicts) return 1;
  if (!push_dictionaries_to(s, 1)) return 0;
  s->ctx_dicts = s->ndicts;
  return 1;
}
static int push_dictionaries_to(BrotliEncoderState* s, int to_context) {
  BrotliAmdDictChunk ch[15];
  uint32_t i;
  for (i = 0; i < s->ndicts; ++i) {
    ch[i].source = s->dicts[i]->source;
    ch[i].starts = s->dicts[i]->starts;
    ch[i].items = s->dicts[i]->items;
    ch[i].source_size = s->dicts[i]->source_size;
    ch[i].bucket_bits = s->dicts[i]->bucket_bits;
  }
  if (to_context) return brotli_amd_ctx_set_dictionary(s->ctx, ch, s->ndicts) == BROTLI_AMD_OK;
  return brotli_amd_stream_attach_dictionary(s->stream, ch, s->ndicts) == BROTLI_AMD_OK;
}
/* BROTLI_PARAM_LGBLOCK as the device layer takes it (bits 24..28 of the flags; 0 = the quality's default), and the
   input block it results in (ComputeLgBlock, quality.h:75-92: looked at from quality 4 on, clamped to 16 .. 24). */
static uint32_t lgblock_flag(const BrotliEncoderState* s) {
  const int lg = s->lgblock;
  if (lg == 0 || s->quality < 4) return 0u;
  return BROTLI_AMD_FLAG_LGBLOCK(lg < 16 ? 16 : lg > 24 ? 24 : lg);
}
static int eff_lgblock(const BrotliEncoderState* s) {
  if (s->quality < 4) return 14;
  if (s->lgblock != 0) return s->lgblock < 16 ? 16 : s->lgblock > 24 ? 24 : s->lgblock;
  if (s->quality >= 9 && s->lgwin > 16) return s->lgwin < 18 ? s->lgwin : 18;
  return 16;
}
static int open_stream(BrotliEncoderState* s) {
  if (s->stream) return 1;
  if (brotli_amd_stream_create(s->ctx, s->quality, s->lgwin, s->eff_hint, s->stream_offset,
                               (s->header_written == 2 ? BROTLI_AMD_FLAG_NO_HEADER : 0u) |
                               (s->disable_ctx ? BROTLI_AMD_FLAG_NO_LITERAL_CONTEXT : 0u) | lgblock_flag(s),
                               &s->stream) != BROTLI_AMD_OK) return 0;
  if (s->ndicts && !push_dictionaries(s)) return 0;
  return 1;
}

/* Quality 1: the calls buffered since the last flush become one device job; the
   bytes completed so far leave, the partial byte stays pending (encode.c:1521-1535). */
static int submit_fast(BrotliEncoderState* s, int op) {
  if (s->ncalls != 0) {
    BrotliAmdFastParams p;
    BrotliAmdJobInfo info;
    uint64_t cap, nbits = 0;
    size_t nbytes;
    uint8_t* dst;
    p.lgwin = s->lgwin;
    p.carry_bits = s->carry_bits;
    p.carry_value = s->carry_value;
    p.is_last = op == OP_FINISH;
    cap = brotli_amd_fast_max_output(s->in_len, s->ncalls, s->lgwin);
    if (cap == 0) return 0;
    if (s->out_pos == s->out_len) s->out_pos = s->out_len = 0;
    if (!grow(s, &s->out_buf, &s->out_cap, s->out_len, s->out_len + cap)) return 0;
    dst = s->out_buf + s->out_len;
    if (brotli_amd_encode_fast_host(s->ctx, s->in_buf, s->in_len, s->calls, s->ncalls, &p, dst, cap,
                                    &nbits, &info) != BROTLI_AMD_OK) {
      if (s->env_verbose) fprintf(stderr, "brotli_amd: %s\n", brotli_amd_last_error(s->ctx));
      return 0;
    }
    nbytes = (size_t)(nbits >> 3);
    s->carry_bits = (uint32_t)(nbits & 7u);
    s->carry_value = s->carry_bits ? (uint32_t)(dst[nbytes] & ((1u << s->carry_bits) - 1u)) : 0u;
    s->out_len += nbytes;
    s->submitted += s->in_len;
    s->in_len = 0;
    s->ncalls = 0;
  }
  if (op == OP_FLUSH && s->carry_bits != 0) {
    /* InjectBytePaddingBlock, encode.c:1356-1380: an empty metadata block seals the byte */
    uint32_t seal = s->carry_value | (0x6u << s->carry_bits);
    const uint32_t seal_bits = s->carry_bits + 6;
    uint8_t b[2];
    b[0] = (uint8_t)seal;
    b[1] = (uint8_t)(seal >> 8);
    s->carry_bits = 0;
    s->carry_value = 0;
    return out_append(s, b, (seal_bits + 7) >> 3);
  }
  return 1;
}

/* One encoder instance: the persistent device stream reproduces the reference for any op sequence. */
static int submit_serial(BrotliEncoderState* s, int op) {
  const uint8_t* out;
  uint64_t out_len;
  if (!open_stream(s)) {
    if (s->env_verbose) fprintf(stderr, "brotli_amd: %s\n", brotli_amd_last_error(s->ctx));
    return 0;
  }
  if (s->empty_finish && op == OP_FINISH && s->in_len != 0) {
    /* what the caller did: the held bytes as the PROCESS calls they came in — the reference encodes every block they
       complete with is_last = 0 (encode.c:1700-1712), counted from the last flush, i.e. over submitted + in_len, not over
       what happens to be held here — then the empty FINISH (which takes an incomplete last block with it) */
    if (brotli_amd_stream_write(s->stream, s->in_buf, s->in_len, BROTLI_AMD_OP_PROCESS, &out, &out_len) != BROTLI_AMD_OK) {
      if (s->env_verbose) fprintf(stderr, "brotli_amd: %s\n", brotli_amd_last_error(s->ctx));
      return 0;
    }
    s->submitted += s->in_len;
    s->in_len = 0;
    s->header_written = 1;
    if (!out_append(s, out, (size_t)out_len)) return 0;
  }
  if (brotli_amd_stream_write(s->stream, s->in_buf, s->in_len, op, &out, &out_len) != BROTLI_AMD_OK) {
    if (s->env_verbose) fprintf(stderr, "brotli_amd: %s\n", brotli_amd_last_error(s->ctx));
    return 0;
  }
  s->submitted += s->in_len;
  s->in_len = 0;
  s->header_written = 1;
  return out_append(s, out, (size_t)out_len);
}

/* Can the rule that closes a meta-block (encode.c:1141-1166: literals or commands >= a eighth of the largest
   meta-block, or no room for another input block) apply to an input of n bytes at all?  (tail_finish: a one-shard job
   writes the one-shot form, right as long as the last block cannot have closed its meta-block.) */
static int metablock_may_close(const BrotliEncoderState* s, size_t n) {
  const int lgb = eff_lgblock(s);
  const int rb = 1 + (s->lgwin > lgb ? s->lgwin : lgb);
  const size_t mm = (size_t)1 << (rb < 24 ? rb : 24);
  if (s->quality < 4) return 1;          /* (no block splitting: flushed at 12 287 symbols, encode.c:1151-1153) */
  if (s->stream_offset != 0) return 1;   /* (the flint block shifts the boundaries: not reasoned about) */
  return n >= mm / 8;
}

/* A whole quality-5 stream in one FINISH that is longer than the window (BrotliEncoderCompress of a big buffer, the
   CLI on a big file): the tiled stream path of the HIP layer (BROTLI_AMD_FLAG_STREAM_TILES; k_tile.h) parses all its
   input blocks at once.  BROTLI_AMD_STREAM_TILES=0 keeps such streams on the serial device stream. */
static int wants_stream_tiles(const BrotliEncoderState* s, int op) {
  if (s->env_stream_tiles == 0) return 0;
  return s->quality == 5 && s->shard_bytes == 0 && op == OP_FINISH && s->submitted == 0 && !s->stream && s->ndicts == 0 &&
         s->stream_offset == 0 && eff_lgblock(s) == 16 &&
         s->lgwin >= 17 && s->lgwin <= 24 && s->in_len > ((size_t)1 << (s->lgwin < 22 ? s->lgwin : 22)) - 16 &&
         s->in_len < ((size_t)1 << 31);
}

/* Hands everything buffered to the device and applies `op` (1 flush, 2 finish). */
static int submit(BrotliEncoderState* s, int op) {
  const int stream_tiles = wants_stream_tiles(s, op);
  if (s->quality == 1) return submit_fast(s, op);
  if (stream_tiles) {
    /* (the plan code below with shard size 0 and the flag; BROTLI_AMD_SERIAL sends it to submit_serial) */
  } else if (s->shard_bytes == 0 && op == OP_FINISH && s->submitted == 0 && !s->stream &&
      s->in_len != 0 && s->ndicts == 0 && !(s->tail_finish && metablock_may_close(s, s->in_len)) &&
      (s->quality != 5 || (s->lgwin >= 17 && s->in_len <= ((size_t)1 << (s->lgwin < 22 ? s->lgwin : 22)) - 16))) {
    /* Everything in one FINISH (BrotliEncoderCompress, the CLI on a small file): the same bytes come from a
       one-shard job (falls through to the plan code below with shard size 0 = one shard).  Qualities 6-9: any
       length; quality 5: an input that fits the window — and 4 MiB: beyond that a text has keys with more than 65 520
       positions, whose searches count the 16-bit store counter's wraps the slow way (k_chain.h, c_search_exact) — runs
       the position index and the tiled chain (k_index.h, k_chain.h: every tile of the stream parsed at once) instead
       of one wave on the whole stream. */
  } else if (s->shard_bytes == 0 && !(s->in_len == 0 && s->submitted == 0 && !s->stream)) {
    /* One encoder instance: the persistent device stream reproduces the
       reference for any op sequence (qualities 6-9: up to the window size).
       (An operation before the first data byte is answered on the host, below: the reference has
       not chosen its hasher yet either — UpdateSizeHint, encode.c:1619-1632, keeps a hint of 0
       open — so the device stream is only created once there is data to size it by.) */
    return submit_serial(s, op);
  }
  /* Partition plan: the buffered bytes become ceil(n / shard) independent
     shards that start at stream offset stream_offset + submitted. */
  if (s->in_len == 0) {
    if (!s->header_written) {
      /* Nothing was ever encoded: the header bits are still pending
         (encode.c:1356-1415 for FLUSH, :1004-1014 for FINISH). */
      uint32_t bits, nbits, v;
      uint8_t b[2];
      if (s->stream_offset != 0) {
        if (op == OP_FINISH) { b[0] = 3; return out_append(s, b, 1); }
        return 1;
      }
      window_bits(s->lgwin, &bits, &nbits);
      if (op == OP_FINISH) { v = bits | (3u << nbits); nbits += 2; }
      else { v = bits | (6u << nbits); nbits += 6; s->header_written = 2; /* header gone, no data yet */ }
      b[0] = (uint8_t)v;
      b[1] = (uint8_t)(v >> 8);
      if (op == OP_FINISH) s->header_written = 1;
      return out_append(s, b, (nbits + 7) >> 3);
    }
    if (op == OP_FINISH) { uint8_t b = 3; return out_append(s, &b, 1); }   /* ISLAST + ISEMPTY */
    return 1;   /* every job ends byte aligned: an empty flush adds nothing */
  }
  {
    BrotliAmdJobParams p;
    BrotliAmdJobInfo info;
    uint64_t cap, n = 0;
    memset(&p, 0, sizeof(p));
    p.quality = s->quality;
    p.lgwin = s->lgwin;
    p.size_hint = s->eff_hint;
    p.shard_size = s->shard_bytes;
    p.stream_base = (uint64_t)s->stream_offset + s->submitted;
    p.is_last = op == OP_FINISH;
    if (s->header_written == 2 && p.stream_base == 0) p.flags |= BROTLI_AMD_FLAG_NO_HEADER;
    if (s->disable_ctx) p.flags |= BROTLI_AMD_FLAG_NO_LITERAL_CONTEXT;
    p.flags |= lgblock_flag(s);
    if (stream_tiles) p.flags |= BROTLI_AMD_FLAG_STREAM_TILES | (s->tail_finish ? BROTLI_AMD_FLAG_TAIL_FINISH : 0u);
    cap = brotli_amd_max_output(s->in_len, &p);
    if (cap == 0) return 0;
    if (!sync_context_dictionaries(s)) return 0;
    if (s->out_pos == s->out_len) s->out_pos = s->out_len = 0;
    if (s->direct_out && s->out_len == 0) {
      /* straight into the caller's buffer; if the result does not fit (the caller offered less
         than the worst case), once more through the instance's own buffer */
      const int rc = brotli_amd_encode_host(s->ctx, s->in_buf, s->in_len, &p, s->direct_out, s->direct_cap, &n, &info);
      if (rc == BROTLI_AMD_OK) {
        s->direct_n = (size_t)n;
        s->submitted += s->in_len;
        s->in_len = 0;
        s->header_written = 1;
        return 1;
      }
      if (rc == BROTLI_AMD_SERIAL) return submit_serial(s, op);
      if (rc != BROTLI_AMD_OVERFLOW) {
        if (s->env_verbose) fprintf(stderr, "brotli_amd: %s\n", brotli_amd_last_error(s->ctx));
        return 0;
      }
      n = 0;
    }
    if (!grow(s, &s->out_buf, &s->out_cap, s->out_len, s->out_len + cap)) return 0;
    {
      const int rc = brotli_amd_encode_host(s->ctx, s->in_buf, s->in_len, &p, s->out_buf + s->out_len, cap, &n, &info);
      if (rc == BROTLI_AMD_SERIAL) return submit_serial(s, op);
      if (rc != BROTLI_AMD_OK) {
        if (s->env_verbose) fprintf(stderr, "brotli_amd: %s\n", brotli_amd_last_error(s->ctx));
        return 0;
      }
    }
    s->out_len += (size_t)n;
    s->submitted += s->in_len;
    s->in_len = 0;
    s->header_written = 1;
  }
  return 1;
}

/* PROCESS with a lot of input waiting (BROTLI_AMD_FEED_KB): hand it to the device now, so that host memory stays
   bounded and output starts to flow (the reference emits during PROCESS too, encode.c:1665-1722).
   Plan mode: the complete shards that are waiting — shard boundaries sit at multiples of the shard
   size from the last flush either way, so the bytes do not depend on when a shard is submitted.
   One encoder instance: the device stream takes input with BROTLI_AMD_OP_PROCESS. */
static size_t feed_threshold(const BrotliEncoderState* s) {
  const int e = s->env_feed_kb >= 0;                   /* default: 256 MiB of shards / 4 MiB of one stream */
  size_t kb = e ? (size_t)s->env_feed_kb : (s->shard_bytes ? (256u << 10) : (4u << 10));
  if (!e && s->shard_bytes == 0 && s->quality == 5 && s->lgwin >= 17 && s->lgwin <= 24 && s->ndicts == 0 &&
      !s->stream && s->submitted == 0 && s->stream_offset == 0 && eff_lgblock(s) == 16) {
    /* One quality-5 stream fed with PROCESS calls (the CLI, Compressor.process of the Python module): the input is held
       until FINISH, so that the whole stream takes the tiled stream path (submit / wants_stream_tiles) instead of going
       to the serial device stream 4 MiB at a time — what the stream looks like does not depend on how the calls cut it
       (encode.c:1666-1681 re-blocks the input; the size hint is latched by the first full block either way, above).
       Announced (BROTLI_PARAM_SIZE_HINT: the CLI does that for files): held up to the announced size; not announced:
       up to BROTLI_AMD_HOLD_MB (default 1024; 0 = never hold).  A FLUSH, or input beyond that, goes to the serial
       stream as before. */
    const size_t cap_mb = s->env_hold_mb >= 0 ? (size_t)s->env_hold_mb : 1024u;
    if (s->env_stream_tiles != 0 && cap_mb != 0) {
      if (s->size_hint == 0) return cap_mb << 20;
      if (((size_t)s->size_hint >> 20) < cap_mb) return (size_t)s->size_hint + 1u;
    }
  }
  if (kb == 0) kb = 1;
  return kb << 10;
}
static int forward_pending_input(BrotliEncoderState* s, int force) {
  if (s->quality == 1 || !s->hint_fixed || s->in_len == 0 || (!force && s->in_len < feed_threshold(s))) return 1;
  if (s->shard_bytes == 0) {
    const uint8_t* out;
    uint64_t out_len;
    size_t off = 0;
    if (!open_stream(s)) return 0;
    /* in pieces of 4 MiB (what the stream was fed with before inputs were held for the tiled path): a hold that ran into
       its cap hands over up to a GiB here, which as ONE call would want device buffers of several GiB and keep the
       caller waiting for all of it; the bytes do not depend on how the calls cut the input */
    while (off < s->in_len) {
      const size_t n = s->in_len - off < ((size_t)4 << 20) ? s->in_len - off : ((size_t)4 << 20);
      if (brotli_amd_stream_write(s->stream, s->in_buf + off, n, BROTLI_AMD_OP_PROCESS, &out, &out_len) != BROTLI_AMD_OK) {
        if (s->env_verbose) fprintf(stderr, "brotli_amd: %s\n", brotli_amd_last_error(s->ctx));
        return 0;
      }
      if (!out_append(s, out, (size_t)out_len)) return 0;
      off += n;
    }
    s->submitted += s->in_len;
    s->in_len = 0;
    s->header_written = 1;
    return 1;
  } else {
    BrotliAmdJobParams p;
    BrotliAmdJobInfo info;
    uint64_t cap, n = 0;
    const size_t whole = (s->in_len / s->shard_bytes) * s->shard_bytes;
    if (whole == 0) return 1;
    memset(&p, 0, sizeof(p));
    p.quality = s->quality;
    p.lgwin = s->lgwin;
    p.size_hint = s->eff_hint;
    p.shard_size = s->shard_bytes;
    p.stream_base = (uint64_t)s->stream_offset + s->submitted;
    p.is_last = 0;
    if (s->header_written == 2 && p.stream_base == 0) p.flags |= BROTLI_AMD_FLAG_NO_HEADER;
    if (s->disable_ctx) p.flags |= BROTLI_AMD_FLAG_NO_LITERAL_CONTEXT;
    p.flags |= lgblock_flag(s);
    cap = brotli_amd_max_output(whole, &p);
    if (cap == 0) return 0;
    if (!sync_context_dictionaries(s)) return 0;
    if (s->out_pos == s->out_len) s->out_pos = s->out_len = 0;
    if (!grow(s, &s->out_buf, &s->out_cap, s->out_len, s->out_len + cap)) return 0;
    if (brotli_amd_encode_host(s->ctx, s->in_buf, whole, &p, s->out_buf + s->out_len, cap, &n, &info) != BROTLI_AMD_OK) {
      if (s->env_verbose) fprintf(stderr, "brotli_amd: %s\n", brotli_amd_last_error(s->ctx));
      return 0;
    }
    s->out_len += (size_t)n;
    s->submitted += whole;
    memmove(s->in_buf, s->in_buf + whole, s->in_len - whole);
    s->in_len -= whole;
    s->header_written = 1;
    return 1;
  }
}

static void push_output(BrotliEncoderState* s, size_t* available_out, uint8_t** next_out,
                        size_t* total_out) {
  size_t n = s->out_len - s->out_pos;
  if (available_out && *available_out < n) n = *available_out;
  if (!available_out) n = 0;
  if (n) {
    memcpy(*next_out, s->out_buf + s->out_pos, n);
    *next_out += n;
    *available_out -= n;
    s->out_pos += n;
    s->total_out += n;
  }
  if (total_out) *total_out = (size_t)s->total_out;
  /* CheckFlushComplete, encode.c:1417-1423 */
  if ((s->stream_state == ST_FLUSH_REQUESTED || s->stream_state == ST_METADATA) && s->out_pos == s->out_len)
    s->stream_state = ST_PROCESSING;
}

BROTLI_BOOL BrotliEncoderCompressStream(BrotliEncoderState* s, int op, size_t* available_in,
                                        const uint8_t** next_in, size_t* available_out,
                                        uint8_t** next_out, size_t* total_out) {
  if (!ensure_initialized(s)) return BROTLI_FALSE;
  if (op == OP_EMIT_METADATA) {
    /* ProcessMetadata / WriteMetadataHeader (encode.c:1223-1249, 1549-1617): the header
       continues the partial last byte of the data before it.  At quality 1 that byte is
       kept on the host (carry_*), so the block is assembled here; at the other qualities
       the partial byte lives in the device-side shard state and this operation is not
       offered. */
    const size_t n = *available_in;
    uint64_t bits;
    uint32_t nbits;
    uint8_t hdr[8];
    size_t hb, i;
    const int single5 = s->quality >= 2 && s->quality <= 9 && s->shard_bytes == 0;
    if (s->quality != 1 && !single5) {
      if (s->env_verbose) fprintf(stderr, "brotli_amd: BROTLI_OPERATION_EMIT_METADATA needs quality 1 or an "
                                     "unpartitioned quality-5 stream\n");
      return BROTLI_FALSE;
    }
    if (n > (1u << 24)) return BROTLI_FALSE;                        /* :1552 */
    if (s->stream_state == ST_METADATA) {
      /* the block is complete but not yet taken: the caller keeps calling with the same
         operation and no input until the output is drained (:1641-1645, 1583-1589) */
      if (n != 0) return BROTLI_FALSE;
      push_output(s, available_out, next_out, total_out);
      return BROTLI_TRUE;
    }
    if (s->stream_state != ST_PROCESSING) return BROTLI_FALSE;      /* :1558-1561 */
    if (single5) {
      /* the device stream flushes what it holds as a meta-block without the padding block
         and hands over the open byte */
      const uint8_t* o;
      uint64_t on;
      if (!s->hint_fixed && s->total_in != 0) {                    /* UpdateSizeHint(s, 0), :1647: a hint of 0 stays open */
        s->eff_hint = s->total_in >= (1u << 30) ? (1u << 30) : (uint32_t)s->total_in;
        s->hint_fixed = 1;
      }
      if (!s->stream && s->in_len == 0 && s->submitted == 0) {
        /* metadata before the first data byte: the stream header is still pending and goes out
           in front of the metadata block; the device stream will start without it */
        if (!s->header_written && s->stream_offset == 0) window_bits(s->lgwin, &s->carry_value, &s->carry_bits);
        s->header_written = 2;
      } else {
        if (!open_stream(s)) { s->failed = 1; return BROTLI_FALSE; }
        if (brotli_amd_stream_write(s->stream, s->in_buf, s->in_len, BROTLI_AMD_OP_FLUSH_OPEN, &o, &on) != BROTLI_AMD_OK ||
            brotli_amd_stream_take_partial(s->stream, &s->carry_bits, &s->carry_value) != BROTLI_AMD_OK) {
          if (s->env_verbose) fprintf(stderr, "brotli_amd: %s\n", brotli_amd_last_error(s->ctx));
          s->failed = 1;
          return BROTLI_FALSE;
        }
        s->submitted += s->in_len;
        s->in_len = 0;
        s->header_written = 1;
        if (!out_append(s, o, (size_t)on)) return BROTLI_FALSE;
      }
    } else if (!submit_fast(s, OP_PROCESS)) {                        /* data fed so far comes first */
      s->failed = 1;
      return BROTLI_FALSE;
    }
    bits = s->carry_value;
    nbits = s->carry_bits;
    /* ISLAST 0, MNIBBLES 11 (= 0 nibbles), reserved 0 */
    bits |= (uint64_t)0x6u << nbits;
    nbits += 4;
    if (n == 0) {
      nbits += 2;                                                   /* MSKIPBYTES 0 */
    } else {
      uint32_t lb = 1, nbytes;
      if (n > 1) { uint32_t v = (uint32_t)n - 1; lb = 0; while (v) { ++lb; v >>= 1; } }
      nbytes = (lb + 7) / 8;
      bits |= (uint64_t)nbytes << nbits;
      nbits += 2;
      bits |= (uint64_t)(n - 1) << nbits;
      nbits += 8 * nbytes;
    }
    hb = (nbits + 7) >> 3;
    for (i = 0; i < hb; ++i) hdr[i] = (uint8_t)(bits >> (8 * i));
    s->carry_bits = 0;
    s->carry_value = 0;
    if (!out_append(s, hdr, hb)) return BROTLI_FALSE;
    if (n && !out_append(s, *next_in, n)) return BROTLI_FALSE;
    *next_in += n;
    *available_in = 0;
    s->total_in += n;
    s->stream_state = ST_METADATA;
    push_output(s, available_out, next_out, total_out);
    return BROTLI_TRUE;
  }
  if (s->stream_state == ST_METADATA) return BROTLI_FALSE;          /* :1652-1655 */
  if (op < 0 || op > 2) return BROTLI_FALSE;
  if (s->stream_state != ST_PROCESSING && *available_in != 0) return BROTLI_FALSE;   /* encode.c:1657 */
  if (s->stream_state == ST_PROCESSING) {
    const size_t a = *available_in;
    /* UpdateSizeHint at the reference's first EncodeData (encode.c:1619-1632):
       when the first input block fills, or at the first op other than PROCESS. */
    if (!s->hint_fixed) {
      /* flint, or the input block size (quality.h:75-93) */
      const uint64_t threshold = s->stream_offset ? 2u : (uint64_t)1 << eff_lgblock(s);
      const uint64_t seen = s->total_in + a;
      if ((seen >= threshold || op != OP_PROCESS) && seen != 0) {   /* a hint of 0 stays open (:1620) */
        s->eff_hint = seen >= (1u << 30) ? (1u << 30) : (uint32_t)seen;
        s->hint_fixed = 1;
      }
    }
    if (s->quality == 1 && (a != 0 || op == OP_FINISH)) {
      /* every call is cut into its own fragments (block_size = min(1 << lgwin,
         *available_in), encode.c:1478); a FINISH without input is an empty
         fragment that carries ISLAST (:1479-1480) */
      if (s->ncalls == s->calls_cap) {
        const size_t nc = s->calls_cap ? 2 * s->calls_cap : 64;
        uint64_t* n = (uint64_t*)st_alloc(s, nc * sizeof(uint64_t));
        if (!n) return BROTLI_FALSE;
        if (s->ncalls) memcpy(n, s->calls, s->ncalls * sizeof(uint64_t));
        st_free(s, s->calls);
        s->calls = n;
        s->calls_cap = nc;
      }
      s->calls[s->ncalls++] = a;
    }
    if (a && op != OP_PROCESS && s->in_len == 0 && a >= ((size_t)1 << 16) && s->quality != 1 &&
        (s->shard_bytes != 0 || (op == OP_FINISH && s->submitted == 0 && !s->stream))) {
      /* A job (partition plan, the one-shard job of a one-shot call, the tiled stream of a long quality-5 one)
         whose whole input arrives with the FLUSH / FINISH: it reads the caller's buffer, and writes the caller's
         output buffer when nothing is waiting in front of it.  (Quality 5 without a plan went through the
         instance's own buffers until round 4: a GiB copied in, 2 GiB of output buffer grown and the result copied
         out were 250 of the 520 ms of the 1 GiB stock call, profiles/r04_g2_summary.txt.  Whatever submit() routes
         to the serial device stream reads the caller's buffer just the same.) */
      uint8_t* const own = s->in_buf;
      int ok;
      s->in_buf = (uint8_t*)(uintptr_t)*next_in;
      s->in_len = a;
      s->total_in += a;
      s->direct_out = (available_out && next_out && s->out_pos == s->out_len) ? *next_out : NULL;
      s->direct_cap = s->direct_out ? *available_out : 0;
      s->direct_n = 0;
      ok = submit(s, op);
      s->in_buf = own;
      s->in_len = 0;
      s->direct_out = NULL;
      if (!ok) { s->failed = 1; return BROTLI_FALSE; }
      *next_in += a;
      *available_in = 0;
      if (s->direct_n) {
        *next_out += s->direct_n;
        *available_out -= s->direct_n;
        s->total_out += s->direct_n;
        s->direct_n = 0;
      }
      s->stream_state = op == OP_FINISH ? ST_FINISHED : ST_FLUSH_REQUESTED;
      push_output(s, available_out, next_out, total_out);
      return BROTLI_TRUE;
    }
    if (a) {
      if (!grow_in(s, s->in_len + a)) return BROTLI_FALSE;
      memcpy(s->in_buf + s->in_len, *next_in, a);
      s->in_len += a;
      s->total_in += a;
      *next_in += a;
      *available_in = 0;
    }
    if (op != OP_PROCESS) {
      /* The reference encodes a block as soon as PROCESS calls have filled it (encode.c:1700-1712), not knowing that the
         stream ends there: a FINISH that brings nothing then finds the last block done with is_last = 0 (submit). */
      /* (behind a stream offset the two "flint" bytes are a block of their own, encode.c:1686-1694: no arithmetic here —
          such a stream goes to the serial device stream call by call, metablock_may_close) */
      /* (tail_finish chooses between the one-shard / tiled job and the serial stream, where nothing was submitted yet and
          in_len is the whole stream; the serial stream itself replays the calls whatever the alignment: empty_finish) */
      s->empty_finish = op == OP_FINISH && a == 0 && s->in_len != 0;
      s->tail_finish = s->empty_finish &&
                       (s->stream_offset != 0 || (s->in_len & (((size_t)1 << eff_lgblock(s)) - 1u)) == 0);
      if (!submit(s, op)) { s->failed = 1; return BROTLI_FALSE; }
      s->tail_finish = 0;
      s->empty_finish = 0;
      s->stream_state = op == OP_FINISH ? ST_FINISHED : ST_FLUSH_REQUESTED;
    } else if (!forward_pending_input(s, 0)) {
      s->failed = 1;
      return BROTLI_FALSE;
    }
  }
  push_output(s, available_out, next_out, total_out);
  return BROTLI_TRUE;
}

BROTLI_BOOL BrotliEncoderIsFinished(BrotliEncoderState* s) {
  return s->stream_state == ST_FINISHED && s->out_pos == s->out_len;
}

BROTLI_BOOL BrotliEncoderHasMoreOutput(BrotliEncoderState* s) { return s->out_pos != s->out_len; }

const uint8_t* BrotliEncoderTakeOutput(BrotliEncoderState* s, size_t* size) {
  size_t n = s->out_len - s->out_pos;
  const uint8_t* p = s->out_buf + s->out_pos;
  if (*size && *size < n) n = *size;
  if (n == 0) { *size = 0; return NULL; }
  s->out_pos += n;
  s->total_out += n;
  /* CheckFlushComplete (encode.c:1417-1423) ends a flush here; a metadata block is only
     closed by the next BrotliEncoderCompressStream(EMIT_METADATA) call (:1583-1589), as in
     the reference: any other operation before that is refused */
  if (s->stream_state == ST_FLUSH_REQUESTED && s->out_pos == s->out_len) s->stream_state = ST_PROCESSING;
  *size = n;
  return p;
}

/* Worst case of the raw fallback stream below (the bound the reference publishes,
   c/include/brotli/encode.h:388-404): two header bytes, one 4-byte meta-block header per 16 KiB
   started... the reference prices 4 bytes for every complete 16 KiB, plus 3 + 1 closing bytes. */
size_t BrotliEncoderMaxCompressedSize(size_t input_size) {
  size_t bound;
  if (input_size == 0) return 2;
  bound = input_size + 4 * (input_size >> 14) + 6;
  return bound < input_size ? 0 : bound;                      /* overflow: no bound exists */
}

/* The whole input as stored meta-blocks (RFC 7932 section 9.2: ISLAST 0, MNIBBLES, MLEN - 1,
   ISUNCOMPRESSED 1, byte aligned payload) behind a lgwin 24 header, closed by an empty last
   meta-block: the bytes BrotliEncoderCompress falls back to when compression does not pay
   (same stream as c/enc/encode.c:1264-1294). */
static size_t make_uncompressed_stream(const uint8_t* input, size_t input_size, uint8_t* output) {
  uint8_t* w = output;
  size_t left = input_size;
  if (input_size == 0) { *w = 0x06; return 1; }                /* WBITS 16, ISLAST, ISEMPTY */
  *w++ = 0x21;                                                  /* WBITS = 24 (0100001b) + ... */
  *w++ = 0x03;                                                  /* ... an empty metadata block pads the byte */
  while (left != 0) {
    const uint32_t mlen = left > (1u << 24) ? (1u << 24) : (uint32_t)left;
    const uint32_t nib = mlen > (1u << 20) ? 6u : mlen > (1u << 16) ? 5u : 4u;   /* nibbles of MLEN - 1 */
    /* bit 0 ISLAST = 0 | bits 1-2 MNIBBLES - 4 | MLEN - 1 | ISUNCOMPRESSED */
    const uint32_t head = ((nib - 4u) << 1) | ((mlen - 1u) << 3) | (1u << (3u + 4u * nib));
    const uint32_t head_bytes = nib == 6u ? 4u : 3u;
    uint32_t i;
    for (i = 0; i < head_bytes; ++i) *w++ = (uint8_t)(head >> (8u * i));
    memcpy(w, input, mlen);
    w += mlen;
    input += mlen;
    left -= mlen;
  }
  *w++ = 0x03;                                                  /* ISLAST, ISEMPTY */
  return (size_t)(w - output);
}

BROTLI_BOOL BrotliEncoderCompress(int quality, int lgwin, int mode, size_t input_size,
                                  const uint8_t* input_buffer, size_t* encoded_size,
                                  uint8_t* encoded_buffer) {
  BrotliEncoderState* s;
  const size_t out_size = *encoded_size;
  const size_t max_out_size = BrotliEncoderMaxCompressedSize(input_size);
  if (out_size == 0) return BROTLI_FALSE;
  if (input_size == 0) { *encoded_size = 1; *encoded_buffer = 6; return BROTLI_TRUE; }
  s = BrotliEncoderCreateInstance(0, 0, 0);
  if (!s) return BROTLI_FALSE;
  {
    size_t available_in = input_size, available_out = *encoded_size, total_out = 0;
    const uint8_t* next_in = input_buffer;
    uint8_t* next_out = encoded_buffer;
    BROTLI_BOOL result, unsupported;
    BrotliEncoderSetParameter(s, P_QUALITY, (uint32_t)quality);
    BrotliEncoderSetParameter(s, P_LGWIN, (uint32_t)lgwin);
    BrotliEncoderSetParameter(s, P_MODE, (uint32_t)mode);
    BrotliEncoderSetParameter(s, P_SIZE_HINT, (uint32_t)input_size);
    if (lgwin > 24) BrotliEncoderSetParameter(s, P_LARGE_WINDOW, 1);
    result = BrotliEncoderCompressStream(s, OP_FINISH, &available_in, &next_in, &available_out,
                                         &next_out, &total_out);
    unsupported = s->failed;
    if (!BrotliEncoderIsFinished(s)) result = 0;
    *encoded_size = total_out;
    BrotliEncoderDestroyInstance(s);
    if (unsupported) { *encoded_size = 0; return BROTLI_FALSE; }   /* fail loudly, no raw stream */
    if (result && !(max_out_size && *encoded_size > max_out_size)) return BROTLI_TRUE;
  }
  /* The compressed stream does not fit: store raw (encode.c:1345-1353). */
  *encoded_size = 0;
  if (!max_out_size) return BROTLI_FALSE;
  if (out_size >= max_out_size) {
    *encoded_size = make_uncompressed_stream(input_buffer, input_size, encoded_buffer);
    return BROTLI_TRUE;
  }
  return BROTLI_FALSE;
}

uint32_t BrotliEncoderVersion(void) { return 0x1002000; }   /* 1.2.0, c/common/version.h */

/* encode.h:531: the reference reports host memory; here the encoder state lives on the device, so
   this is the HOST side of one instance (staging of input and output) — the device workspace is
   sized by brotli_amd_max_output / DESIGN.md section 3 and is not host memory. */
size_t BrotliEncoderEstimatePeakMemoryUsage(int quality, int lgwin, size_t input_size) {
  (void)quality; (void)lgwin;
  return 2 * input_size + BrotliEncoderMaxCompressedSize(input_size) + (1u << 16);
}
/* encode.h:534-542: bytes this library holds for the prepared dictionary (the index; the raw bytes
   stay the caller's), 0 for anything that is not one of ours. */
size_t BrotliEncoderGetPreparedDictionarySize(const BrotliEncoderPreparedDictionary* dictionary) {
  const DictIndex* d = (const DictIndex*)dictionary;
  if (!d || d->magic != DICT_INDEX_MAGIC) return 0;
  return sizeof(*d) + (((size_t)1 << d->bucket_bits) + 1) * 4 + ((size_t)d->num_items + 1) * 4;
}

/* encode.h:318-346, encode.c:1756-1799.  Only BROTLI_SHARED_DICTIONARY_RAW (0): an LZ77 prefix.  The
   serialized form (1) is an experimental build option of the reference and answers NULL there too.
   The index is built on the host (dict_index.h) — once per dictionary, off the hot path — and goes
   to the device with the encoder instance it is attached to. */
BrotliEncoderPreparedDictionary* BrotliEncoderPrepareDictionary(
    int type, size_t data_size, const uint8_t* data, int quality,
    brotli_amd_alloc_func alloc_func, brotli_amd_free_func free_func, void* opaque) {
  DictIndex* d;
  (void)quality;
  if (type != 0 || data_size > DICT_INDEX_MAX_RAW) return NULL;
  if ((alloc_func == NULL) != (free_func == NULL)) return NULL;
  d = (DictIndex*)(alloc_func ? alloc_func(opaque, sizeof(*d)) : malloc(sizeof(*d)));
  if (!d) return NULL;
  memset(d, 0, sizeof(*d));
  d->alloc = alloc_func;
  d->free_ = free_func;
  d->opaque = opaque;
  if (!dict_index_build(d, data, data_size)) {
    if (free_func) free_func(opaque, d); else free(d);
    return NULL;
  }
  return (BrotliEncoderPreparedDictionary*)d;
}

void BrotliEncoderDestroyPreparedDictionary(BrotliEncoderPreparedDictionary* dictionary) {
  DictIndex* d = (DictIndex*)dictionary;
  brotli_amd_free_func free_func;
  void* opaque;
  if (!d || d->magic != DICT_INDEX_MAGIC) return;     /* encode.c:1806-1809: only what Prepare made */
  free_func = d->free_;
  opaque = d->opaque;
  dict_index_release(d);
  if (free_func) free_func(opaque, d); else free(d);
}

/* encode.c:1828-1850 + AttachPreparedDictionary (compound_dictionary.c:182-211): at most 15 chunks,
   2^31 - 1 bytes in total; allowed at any time for raw dictionaries, it then takes effect with the
   next input block.  Qualities 0 / 1 accept and ignore dictionaries (their compressors never look at
   params.dictionary).  In a partition plan every shard's instance has the dictionaries attached (the
   jobs of the context carry them, brotli_amd_ctx_set_dictionary): the bytes are the reference's driven
   with the same plan and the same Attach calls on every instance, and shards submitted after an Attach
   see the new dictionary. */
BROTLI_BOOL BrotliEncoderAttachPreparedDictionary(BrotliEncoderState* state,
                                                  const BrotliEncoderPreparedDictionary* dictionary) {
  const DictIndex* d = (const DictIndex*)dictionary;
  if (!state || !d || d->magic != DICT_INDEX_MAGIC) return BROTLI_FALSE;
  if (state->ndicts == 15) return BROTLI_FALSE;
  if (d->source_size > DICT_INDEX_MAX_RAW - state->dict_total) return BROTLI_FALSE;
  /* The reference has already parsed every complete input block it was given (encode.c:1700-1720) — without
     this dictionary.  Input still waiting on the host goes to the device first (complete blocks are parsed there,
     the partial one stays pending, as in the reference), then the dictionary joins. */
  if (state->initialized && !state->failed && state->ctx && !forward_pending_input(state, 1)) {
    state->failed = 1;
    return BROTLI_FALSE;
  }
  state->dicts[state->ndicts++] = d;
  state->dict_total += d->source_size;
  if (state->stream && !push_dictionaries(state)) {
    state->failed = 1;
    return BROTLI_FALSE;
  }
  return BROTLI_TRUE;
}
