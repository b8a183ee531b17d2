/* oracle/brotli_oracle.c — TEST INFRASTRUCTURE ONLY (see brotli_oracle.h).
 *
 * Scalar CPU restatement of the google/brotli encoder path used at qualities
 * 5..9 (no dictionaries, default distance parameters, lgwin 17..24):
 *   stream driver      c/enc/encode.c:642-700, 841-894, 905-971, 985-1221,
 *                      1356-1415, 1634-1722
 *   match search       c/enc/backward_references_inc.h:10-242,
 *                      c/enc/hash_longest_match64_simd_inc.h (H68),
 *                      c/enc/hash_longest_match_simd_inc.h (H58),
 *                      c/enc/hash_longest_match64_inc.h (H6),
 *                      c/enc/hash_longest_match_inc.h (H5), c/enc/hash.h:80-202
 *   meta-block builder c/enc/metablock.c:463-859, c/enc/metablock_inc.h,
 *                      c/enc/bit_cost.c:18-44, c/enc/encode.c:258-496
 *   entropy coder      c/enc/brotli_bit_stream.c:34-1114, 1321-1352,
 *                      c/enc/entropy_encode.c:20-497, c/enc/write_bits.h:33-54
 * It is pinned against the reference itself (oracle/_ref) by
 * tests/test_oracle.py; it is never linked into the product.
 */
#include "brotli_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int g_lgblock_param = 0;   /* oracle_set_lgblock */

/* ------------------------------------------------------------------------ */
/* Format data (loaded from the blob produced by tools/gen_tables.c).        */

static uint8_t g_context_lut[2048];
static uint8_t g_size_bits_by_length[32];
static uint32_t g_offsets_by_length[32];
static uint8_t* g_dict;
static uint32_t g_dict_size;
static uint16_t g_hash_words[32768];
static uint8_t g_hash_lengths[32768];
static double g_log2_small[256]; /* c/enc/fast_log.c:14: float literals */
static int g_ready;

int oracle_init(const char* path) {
  FILE* f;
  char magic[4];
  uint32_t ver;
  int i;
  if (g_ready) return 0;
  f = fopen(path, "rb");
  if (!f) return -1;
  if (fread(magic, 1, 4, f) != 4 || memcmp(magic, "BRTB", 4)) goto bad;
  if (fread(&ver, 4, 1, f) != 1 || ver != 1) goto bad;
  if (fread(g_context_lut, 1, 2048, f) != 2048) goto bad;
  if (fread(g_size_bits_by_length, 1, 32, f) != 32) goto bad;
  if (fread(g_offsets_by_length, 4, 32, f) != 32) goto bad;
  if (fread(&g_dict_size, 4, 1, f) != 1) goto bad;
  g_dict = (uint8_t*)malloc(((size_t)g_dict_size + 3u) & ~(size_t)3);
  if (fread(g_dict, 1, (g_dict_size + 3u) & ~3u, f) != ((g_dict_size + 3u) & ~3u))
    goto bad;
  if (fread(g_hash_words, 2, 32768, f) != 32768) goto bad;
  if (fread(g_hash_lengths, 1, 32768, f) != 32768) goto bad;
  fclose(f);
  g_log2_small[0] = 0.0;
  for (i = 1; i < 256; ++i) g_log2_small[i] = (double)(float)log2((double)i);
  g_ready = 1;
  return 0;
bad:
  fclose(f);
  return -2;
}

/* c/enc/fast_log.h:51-59 */
static double FastLog2(size_t v) {
  if (v < 256) return g_log2_small[v];
  return log2((double)v);
}

/* c/enc/fast_log.h:20-28 (BSR32 on the low 32 bits) */
static uint32_t Log2Floor(size_t n) {
  return 31u ^ (uint32_t)__builtin_clz((uint32_t)n);
}

static uint64_t Load64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static uint32_t Load32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }

/* ------------------------------------------------------------------------ */
/* Encoder instance.                                                         */

typedef OracleCommand Cmd;

typedef struct {
  uint32_t score_len; /* unused */
} Unused;

typedef struct Enc {
  int quality, lgwin, lgblock;
  uint32_t size_hint;
  size_t stream_offset;
  /* hasher (c/enc/quality.h:172-223) */
  int hasher_type; /* 5, 6, 58, 68; 2, 3, 4, 54 (HashLongestMatchQuickly) */
  int bucket_bits, block_bits, ndist;
  int sweep_bits, hash_len, use_dictionary; /* the quickly family (hash.h:251-279, 329-338) */
  /* the forgetful-chain family H40 / H41 / H42 (hash.h:296-326): addr / head per bucket, a tiny
     hash per 16-bit position, chain nodes in banks of recycled slots */
  int num_banks, bank_bits;
  size_t max_hops;
  uint32_t* fc_addr;
  uint16_t* fc_head;
  uint8_t* fc_tiny;
  uint16_t* fc_free;
  uint16_t* fc_delta;
  uint16_t* fc_next;
  int hasher_setup, hasher_prepared;
  uint16_t* num;
  uint8_t* tags;
  uint32_t* buckets;
  size_t dict_lookups, dict_matches;
  /* ring buffer (c/enc/ringbuffer.h) */
  uint8_t* rb_data;
  uint8_t* rb;
  uint32_t rb_size, rb_mask, rb_tail, rb_total, rb_cur, rb_pos;
  /* stream state (c/enc/state.h:49-110) */
  uint64_t input_pos, last_flush_pos, last_processed_pos;
  Cmd* cmds;
  size_t ncmds, cmd_cap, nlits, last_insert_len;
  int dist_cache[16];
  int saved_dist_cache[4];
  uint16_t last_bytes;
  uint8_t last_bytes_bits;
  int flint;
  uint8_t prev_byte, prev_byte2;
  int is_last_emitted;
  /* output accumulation */
  uint8_t* out;
  size_t out_len, out_cap;
  int overflow;
} Enc;

static OracleCommand* g_tap;
static size_t g_tap_cap;
static size_t* g_tap_n;
void oracle_set_command_tap(OracleCommand* cmds, size_t cap, size_t* ncmds) {
  g_tap = cmds; g_tap_cap = cap; g_tap_n = ncmds;
  if (ncmds) *ncmds = 0;
}

static void Emit(Enc* s, const uint8_t* p, size_t n) {
  if (s->out_len + n > s->out_cap) { s->overflow = 1; return; }
  memcpy(s->out + s->out_len, p, n);
  s->out_len += n;
}

/* c/enc/write_bits.h:33-54 */
static void WriteBits(size_t n_bits, uint64_t bits, size_t* pos, uint8_t* array) {
  uint8_t* p = &array[*pos >> 3];
  uint64_t v = (uint64_t)(*p);
  v |= bits << (*pos & 7);
  memcpy(p, &v, 8);
  *pos += n_bits;
}

/* ------------------------------------------------------------------------ */
/* Ring buffer, c/enc/ringbuffer.h:70-159.                                   */

static void RbInitBuffer(Enc* s, uint32_t buflen) {
  uint8_t* nd = (uint8_t*)malloc(2 + (size_t)buflen + 7);
  size_t i;
  if (s->rb_data) {
    memcpy(nd, s->rb_data, 2 + (size_t)s->rb_cur + 7);
    free(s->rb_data);
  }
  s->rb_data = nd;
  s->rb_cur = buflen;
  s->rb = nd + 2;
  s->rb[-2] = s->rb[-1] = 0;
  for (i = 0; i < 7; ++i) s->rb[s->rb_cur + i] = 0;
}

static void RbWrite(Enc* s, const uint8_t* bytes, size_t n) {
  if (s->rb_pos == 0 && n < s->rb_tail) {
    s->rb_pos = (uint32_t)n;
    RbInitBuffer(s, s->rb_pos);
    memcpy(s->rb, bytes, n);
    return;
  }
  if (s->rb_cur < s->rb_total) {
    RbInitBuffer(s, s->rb_total);
    s->rb[s->rb_size - 2] = 0;
    s->rb[s->rb_size - 1] = 0;
    s->rb[s->rb_size] = 241;
  }
  {
    const size_t masked_pos = s->rb_pos & s->rb_mask;
    if (masked_pos < s->rb_tail) {
      const size_t p = s->rb_size + masked_pos;
      size_t m = s->rb_tail - masked_pos;
      memcpy(&s->rb[p], bytes, n < m ? n : m);
    }
    if (masked_pos + n <= s->rb_size) {
      memcpy(&s->rb[masked_pos], bytes, n);
    } else {
      size_t m = s->rb_total - masked_pos;
      memcpy(&s->rb[masked_pos], bytes, n < m ? n : m);
      memcpy(&s->rb[0], bytes + (s->rb_size - masked_pos),
             n - (s->rb_size - masked_pos));
    }
  }
  {
    int not_first_lap = (s->rb_pos & (1u << 31)) != 0;
    uint32_t m31 = (1u << 31) - 1;
    s->rb[-2] = s->rb[s->rb_size - 2];
    s->rb[-1] = s->rb[s->rb_size - 1];
    s->rb_pos = (s->rb_pos & m31) + (uint32_t)(n & m31);
    if (not_first_lap) s->rb_pos |= 1u << 31;
  }
}

/* c/enc/encode.c:841-894 */
static void CopyInputToRingBuffer(Enc* s, size_t n, const uint8_t* in) {
  RbWrite(s, in, n);
  s->input_pos += n;
  if (s->rb_pos <= s->rb_mask) memset(s->rb + s->rb_pos, 0, 7);
}

/* c/enc/encode.c:127-135 */
static uint32_t WrapPosition(uint64_t position) {
  uint32_t result = (uint32_t)position;
  uint64_t gb = position >> 30;
  if (gb > 2) {
    result = (result & ((1u << 30) - 1)) | ((uint32_t)((gb - 1) & 1) + 1) << 30;
  }
  return result;
}

/* ------------------------------------------------------------------------ */
/* Hashers.                                                                  */

static const uint32_t kHashMul32 = 0x1E35A7BD;
static const uint64_t kHashMul64 = 0x1FE35A7BD3579BD3ull;

static int HasherQuick(const Enc* s) { return s->hasher_type < 5 || s->hasher_type == 54; }
static int HasherChain(const Enc* s) { return s->hasher_type >= 40 && s->hasher_type <= 42; }
static int HasherTagged(const Enc* s) { return s->hasher_type >= 58; }
static int Hasher64(const Enc* s) { return s->hasher_type == 6 || s->hasher_type == 68; }
/* HashTypeLength == StoreLookahead: 8 for the 64-bit and the quickly hashers
   (hash_longest_match_quickly_inc.h:17-18) */
static size_t HashTypeLength(const Enc* s) { return (Hasher64(s) || HasherQuick(s)) ? 8 : 4; }
/* HashBytes of the quickly family, hash_longest_match_quickly_inc.h:23-29 */
static uint32_t QuickKey(const Enc* s, const uint8_t* p) {
  const uint64_t h = (Load64(p) << (64 - 8 * s->hash_len)) * kHashMul64;
  return (uint32_t)(h >> (64 - s->bucket_bits));
}

/* Returns key | tag<<32 (tag only for H58/H68).
   H68: hash_longest_match64_simd_inc.h:26-32; H58: ..._simd_inc.h:18-24;
   H6: hash_longest_match64_inc.h:23-29; H5: hash_longest_match_inc.h. */
static uint64_t HashKeyTag(const Enc* s, const uint8_t* p) {
  if (s->hasher_type == 68) {
    uint64_t h = (Load64(p) * (kHashMul64 << 24)) >> (64 - 15 - 8);
    return (h >> 8) | ((h & 0xFF) << 32);
  } else if (s->hasher_type == 6) {
    return (Load64(p) * (kHashMul64 << 24)) >> (64 - 15);
  } else if (s->hasher_type == 58) {
    uint32_t h = (uint32_t)(Load32(p) * kHashMul32) >> (32 - s->bucket_bits - 8);
    return (uint64_t)(h >> 8) | ((uint64_t)(h & 0xFF) << 32);
  } else {
    return (uint32_t)(Load32(p) * kHashMul32) >> (32 - s->bucket_bits);
  }
}

/* c/enc/quality.h:172-223 (x86-64 build: BROTLI_MAX_RECOMMENDED_SIMD_QUALITY
   is 6, c/common/platform.h:666-668). */
static int ChooseHasher(Enc* s) {
  int q = s->quality;
  if (q >= 2 && q <= 4 && s->lgwin >= 10 && s->lgwin <= 24) {
    /* quality.h:176-179 and the template parameters of hash.h:251-279, 329-338 */
    s->hasher_type = (q == 4 && s->size_hint >= (1u << 20)) ? 54 : q;
    s->bucket_bits = s->hasher_type == 54 ? 20 : s->hasher_type == 4 ? 17 : 16;
    s->sweep_bits = s->hasher_type == 2 ? 0 : s->hasher_type == 3 ? 1 : 2;
    s->hash_len = s->hasher_type == 54 ? 7 : 5;
    s->use_dictionary = s->hasher_type == 2 || s->hasher_type == 4;
    s->block_bits = 0;
    s->ndist = 4;
    return 1;
  }
  if (q < 5 || q > 9 || s->lgwin < 10 || s->lgwin > 24) return 0;
  if (s->lgwin <= 16) {
    /* quality.h:180-181; hash.h:296-326 (15-bit buckets; one bank of 65536 slots, 512 banks of 512
       at quality 9); max_hops: hash_forgetful_chain_inc.h:87 */
    s->hasher_type = q < 7 ? 40 : q < 9 ? 41 : 42;
    s->bucket_bits = 15;
    s->block_bits = 0;
    s->ndist = q < 7 ? 4 : q < 9 ? 10 : 16;
    s->num_banks = q < 9 ? 1 : 512;
    s->bank_bits = q < 9 ? 16 : 9;
    s->max_hops = (size_t)(q > 6 ? 7u : 8u) << (q - 4);
    return 1;
  }
  if (s->size_hint >= (1u << 20) && s->lgwin >= 19) {
    s->hasher_type = q <= 6 ? 68 : 6;
    s->bucket_bits = 15;
  } else {
    s->hasher_type = q <= 6 ? 58 : 5;
    s->bucket_bits = q < 7 ? 14 : 15;
  }
  s->block_bits = q - 1;
  s->ndist = q < 7 ? 4 : q < 9 ? 10 : 16;
  return 1;
}

/* HasherSetup, c/enc/hash.h:450-503; Prepare: ..64_simd_inc.h:81-98 (0xFFFF,
   counting down) / ..64_inc.h:76-91 (0, counting up).  The sparse one-shot
   branch initialises exactly the keys that can be touched, so a full fill is
   equivalent. */
static void HasherSetup(Enc* s) {
  if (HasherChain(s)) {
    /* Prepare, hash_forgetful_chain_inc.h:90-118 (the partial branch touches exactly the buckets
       that can be reached) */
    const size_t nb = (size_t)1 << s->bucket_bits;
    if (!s->hasher_setup) {
      s->fc_addr = (uint32_t*)malloc(nb * 4);
      s->fc_head = (uint16_t*)malloc(nb * 2);
      s->fc_tiny = (uint8_t*)malloc(65536);
      s->fc_free = (uint16_t*)malloc((size_t)s->num_banks * 2);
      s->fc_delta = (uint16_t*)calloc((size_t)s->num_banks << s->bank_bits, 2);
      s->fc_next = (uint16_t*)calloc((size_t)s->num_banks << s->bank_bits, 2);
      s->dict_lookups = s->dict_matches = 0;
      s->hasher_setup = 1;
      s->hasher_prepared = 0;
    }
    if (!s->hasher_prepared) {
      memset(s->fc_addr, 0xCC, nb * 4);
      memset(s->fc_head, 0, nb * 2);
      memset(s->fc_tiny, 0, 65536);
      memset(s->fc_free, 0, (size_t)s->num_banks * 2);
      s->hasher_prepared = 1;
    }
    return;
  }
  if (HasherQuick(s)) {
    /* Prepare, hash_longest_match_quickly_inc.h:49-77: all zero (the sparse branch clears exactly
       the slots that can be touched) */
    const size_t nb = (size_t)1 << s->bucket_bits;
    if (!s->hasher_setup) {
      s->buckets = (uint32_t*)malloc(nb * 4);
      s->dict_lookups = s->dict_matches = 0;
      s->hasher_setup = 1;
      s->hasher_prepared = 0;
    }
    if (!s->hasher_prepared) {
      memset(s->buckets, 0, nb * 4);
      s->hasher_prepared = 1;
    }
    return;
  }
  if (!s->hasher_setup) {
    size_t nb = (size_t)1 << s->bucket_bits, bs = (size_t)1 << s->block_bits;
    s->num = (uint16_t*)malloc(nb * 2);
    s->buckets = (uint32_t*)malloc(nb * bs * 4);
    s->tags = HasherTagged(s) ? (uint8_t*)calloc(nb * bs, 1) : NULL;
    memset(s->buckets, 0, nb * bs * 4);
    s->dict_lookups = s->dict_matches = 0;
    s->hasher_setup = 1;
    s->hasher_prepared = 0;
  }
  if (!s->hasher_prepared) {
    size_t nb = (size_t)1 << s->bucket_bits;
    memset(s->num, HasherTagged(s) ? 255 : 0, nb * 2);
    s->hasher_prepared = 1;
  }
}

/* Store, ..64_simd_inc.h:114-128 / ..64_inc.h:105-115 */
static void HStore(Enc* s, size_t ix) {
  uint64_t kt;
  if (HasherChain(s)) {
    /* hash_forgetful_chain_inc.h:133-149: the node goes to the next slot of the key's bank,
       whatever lived there is forgotten */
    const uint32_t key = (uint32_t)(Load32(&s->rb[ix & s->rb_mask]) * kHashMul32) >> (32 - 15);
    const size_t bank = key & (uint32_t)(s->num_banks - 1);
    const size_t idx = (size_t)(s->fc_free[bank]++ & ((1u << s->bank_bits) - 1u)) + (bank << s->bank_bits);
    size_t delta = ix - s->fc_addr[key];
    s->fc_tiny[(uint16_t)ix] = (uint8_t)key;
    if (delta > 0xFFFF) delta = 0xFFFF;
    s->fc_delta[idx] = (uint16_t)delta;
    s->fc_next[idx] = s->fc_head[key];
    s->fc_addr[key] = (uint32_t)ix;
    s->fc_head[key] = (uint16_t)(idx & ((1u << s->bank_bits) - 1u));
    return;
  }
  if (HasherQuick(s)) {
    /* hash_longest_match_quickly_inc.h:93-104: the slot is wiggled by bits 3.. of the position */
    const uint32_t key = QuickKey(s, &s->rb[ix & s->rb_mask]);
    const uint32_t off = (uint32_t)ix & (((1u << s->sweep_bits) - 1u) << 3);
    s->buckets[(key + off) & ((1u << s->bucket_bits) - 1u)] = (uint32_t)ix;
    return;
  }
  kt = HashKeyTag(s, &s->rb[ix & s->rb_mask]);
  size_t key = (size_t)(kt & 0xFFFFFFFFu);
  uint32_t bmask = (1u << s->block_bits) - 1;
  size_t off = (s->num[key] & bmask) + (key << s->block_bits);
  if (HasherTagged(s)) {
    --s->num[key];
    s->tags[off] = (uint8_t)(kt >> 32);
  } else {
    ++s->num[key];
  }
  s->buckets[off] = (uint32_t)ix;
}

/* c/enc/find_match_length.h:19-40 */
static size_t FindMatchLength(const uint8_t* s1, const uint8_t* s2, size_t limit) {
  size_t n = 0;
  while (limit >= 8) {
    uint64_t x = Load64(s2 + n) ^ Load64(s1 + n);
    if (x) return n + ((size_t)__builtin_ctzll(x) >> 3);
    n += 8; limit -= 8;
  }
  while (limit && s1[n] == s2[n]) { ++n; --limit; }
  return n;
}

typedef struct { size_t len, distance, score; int len_code_delta; } SearchResult;

#define SCORE_BASE (30 * 8 * sizeof(size_t))

/* c/enc/hash.h:123-138 */
static size_t ScoreNormal(size_t len, size_t backward) {
  return SCORE_BASE + 135 * len - 30 * Log2Floor(backward);
}
static size_t ScoreLast(size_t len) { return 135 * len + SCORE_BASE + 15; }
static size_t PenaltyLast(size_t i) { return 39 + ((0x1CA10 >> (i & 0xE)) & 0xE); }

/* c/enc/hash.h:140-202 with the default dictionary (c/enc/encoder_dict.c). */
static void SearchInStaticDictionary(Enc* s, const uint8_t* data,
    size_t max_length, size_t max_backward, size_t max_distance,
    SearchResult* out) {
  size_t key, i;
  const size_t nprobes = HasherQuick(s) ? 1 : 2;   /* `shallow`, hash.h:187 */
  if (s->dict_matches < (s->dict_lookups >> 7)) return;
  key = ((uint32_t)(Load32(data) * kHashMul32) >> (32 - 14)) << 1;
  for (i = 0; i < nprobes; ++i, ++key) {
    size_t len = g_hash_lengths[key];
    s->dict_lookups++;
    if (len != 0) {
      size_t word_idx = g_hash_words[key];
      size_t offset = g_offsets_by_length[len] + len * word_idx;
      size_t matchlen, backward, score, cut, transform_id;
      if (len > max_length) continue;
      matchlen = FindMatchLength(data, &g_dict[offset], len);
      if (matchlen + 10 <= len || matchlen == 0) continue;
      cut = len - matchlen;
      transform_id = (cut << 2) +
          (size_t)((0x071B520ADA2D3200ull >> (cut * 6)) & 0x3F);
      backward = max_backward + 1 + word_idx +
          (transform_id << g_size_bits_by_length[len]);
      if (backward > max_distance) continue;
      score = ScoreNormal(matchlen, backward);
      if (score < out->score) continue;
      out->len = matchlen;
      out->len_code_delta = (int)len - (int)matchlen;
      out->distance = backward;
      out->score = score;
      s->dict_matches++;
    }
  }
}

/* FindLongestMatch of the quickly family, hash_longest_match_quickly_inc.h:141-262: the last
   distance, then the bucket sweep in slot order; a candidate is looked at only if it agrees with
   the input at offset best_len (so the outcome depends on the order), out->len comes in from the
   caller (backward_references_inc.h:127-128). */
static void FindLongestMatchQuick(Enc* s, size_t cur_ix, size_t max_length,
    size_t max_backward, size_t dictionary_distance, size_t max_distance,
    SearchResult* out) {
  const uint8_t* data = s->rb;
  const size_t mask = s->rb_mask;
  const size_t cur_ix_masked = cur_ix & mask;
  const size_t best_len_in = out->len;
  const uint32_t bucket_mask = (1u << s->bucket_bits) - 1u;
  const uint32_t sweep = 1u << s->sweep_bits;
  int compare_char = data[cur_ix_masked + best_len_in];
  const uint32_t key = QuickKey(s, &data[cur_ix_masked]);
  const size_t min_score = out->score;
  size_t best_score = out->score;
  size_t best_len = best_len_in;
  const size_t cached_backward = (size_t)s->dist_cache[0];
  size_t prev_ix = cur_ix - cached_backward;
  uint32_t i;
  out->len_code_delta = 0;
  if (prev_ix < cur_ix && cached_backward <= max_backward) {
    prev_ix &= (uint32_t)mask;
    if (compare_char == data[prev_ix + best_len]) {
      const size_t len = FindMatchLength(&data[prev_ix], &data[cur_ix_masked], max_length);
      if (len >= 4) {
        const size_t score = ScoreLast(len);
        if (best_score < score) {
          out->len = len; out->distance = cached_backward; out->score = score;
          if (sweep == 1) { s->buckets[key] = (uint32_t)cur_ix; return; }
          best_len = len; best_score = score;
          compare_char = data[cur_ix_masked + len];
        }
      }
    }
  }
  if (sweep == 1) {
    size_t backward, len;
    prev_ix = s->buckets[key];
    s->buckets[key] = (uint32_t)cur_ix;
    backward = cur_ix - prev_ix;
    prev_ix &= (uint32_t)mask;
    if (compare_char != data[prev_ix + best_len_in]) return;
    if (backward == 0 || backward > max_backward) return;
    len = FindMatchLength(&data[prev_ix], &data[cur_ix_masked], max_length);
    if (len >= 4) {
      const size_t score = ScoreNormal(len, backward);
      if (best_score < score) { out->len = len; out->distance = backward; out->score = score; return; }
    }
  } else {
    for (i = 0; i < sweep; ++i) {
      size_t backward, len;
      prev_ix = s->buckets[(key + (i << 3)) & bucket_mask];
      backward = cur_ix - prev_ix;
      prev_ix &= (uint32_t)mask;
      if (compare_char != data[prev_ix + best_len]) continue;
      if (backward == 0 || backward > max_backward) continue;
      len = FindMatchLength(&data[prev_ix], &data[cur_ix_masked], max_length);
      if (len >= 4) {
        const size_t score = ScoreNormal(len, backward);
        if (best_score < score) {
          best_len = len; best_score = score;
          compare_char = data[cur_ix_masked + len];
          out->len = len; out->score = score; out->distance = backward;
        }
      }
    }
  }
  if (s->use_dictionary && min_score == out->score) {
    SearchInStaticDictionary(s, &data[cur_ix_masked], max_length, dictionary_distance, max_distance, out);
  }
  if (sweep != 1) {
    s->buckets[(key + ((uint32_t)cur_ix & ((sweep - 1u) << 3))) & bucket_mask] = (uint32_t)cur_ix;
  }
}

/* FindLongestMatch of the forgetful-chain family, hash_forgetful_chain_inc.h:190-298. */
static void FindLongestMatchChain(Enc* s, size_t cur_ix, size_t max_length,
    size_t max_backward, size_t dictionary_distance, size_t max_distance,
    SearchResult* out) {
  const uint8_t* data = s->rb;
  const size_t mask = s->rb_mask;
  const size_t cur_ix_masked = cur_ix & mask;
  const size_t min_score = out->score;
  size_t best_score = out->score;
  size_t best_len = out->len;
  const uint32_t key = (uint32_t)(Load32(&data[cur_ix_masked]) * kHashMul32) >> (32 - 15);
  const uint8_t tiny_hash = (uint8_t)key;
  size_t i;
  out->len = 0;
  out->len_code_delta = 0;
  for (i = 0; i < (size_t)s->ndist; ++i) {
    const size_t backward = (size_t)s->dist_cache[i];
    size_t prev_ix = cur_ix - backward;
    if (i > 0 && s->fc_tiny[(uint16_t)prev_ix] != tiny_hash) continue;
    if (prev_ix >= cur_ix || backward > max_backward) continue;
    prev_ix &= mask;
    {
      const size_t len = FindMatchLength(&data[prev_ix], &data[cur_ix_masked], max_length);
      if (len >= 2) {
        size_t score = ScoreLast(len);
        if (best_score < score) {
          if (i != 0) score -= PenaltyLast(i);
          if (best_score < score) {
            best_score = score; best_len = len;
            out->len = len; out->distance = backward; out->score = score;
          }
        }
      }
    }
  }
  if (best_len < 3) best_len = 3;
  {
    const size_t bank = key & (uint32_t)(s->num_banks - 1);
    size_t backward = 0;
    size_t hops = s->max_hops;
    size_t delta = cur_ix - s->fc_addr[key];
    size_t slot = s->fc_head[key];
    while (hops--) {
      size_t prev_ix;
      const size_t last = slot + (bank << s->bank_bits);
      backward += delta;
      if (backward > max_backward) break;
      prev_ix = (cur_ix - backward) & mask;
      slot = s->fc_next[last];
      delta = s->fc_delta[last];
      if (cur_ix_masked + best_len > mask || prev_ix + best_len > mask ||
          Load32(&data[cur_ix_masked + best_len - 3]) != Load32(&data[prev_ix + best_len - 3])) continue;
      {
        const size_t len = FindMatchLength(&data[prev_ix], &data[cur_ix_masked], max_length);
        if (len >= 4) {
          const size_t score = ScoreNormal(len, backward);
          if (best_score < score) {
            best_score = score; best_len = len;
            out->len = len; out->distance = backward; out->score = score;
          }
        }
      }
    }
    HStore(s, cur_ix);
  }
  if (out->score == min_score) {
    SearchInStaticDictionary(s, &data[cur_ix_masked], max_length, dictionary_distance, max_distance, out);
  }
}

/* FindLongestMatch for all four hashers: ..64_simd_inc.h:170-302,
   .._simd_inc.h:140-277, ..64_inc.h:157-277, .._inc.h. */
static void FindLongestMatch(Enc* s, size_t cur_ix, size_t max_length,
    size_t max_backward, size_t dictionary_distance, size_t max_distance,
    SearchResult* out) {
  const uint8_t* data = s->rb;
  const size_t mask = s->rb_mask;
  const size_t cur_ix_masked = cur_ix & mask;
  const size_t min_score = out->score;
  size_t best_score = out->score;
  size_t best_len = out->len;
  const uint64_t kt = HashKeyTag(s, &data[cur_ix_masked]);
  const size_t key = (size_t)(kt & 0xFFFFFFFFu);
  const uint32_t bsize = 1u << s->block_bits, bmask = bsize - 1;
  uint32_t* bucket = &s->buckets[key << s->block_bits];
  const int is64 = Hasher64(s);
  size_t i;
  if (HasherQuick(s)) {
    FindLongestMatchQuick(s, cur_ix, max_length, max_backward, dictionary_distance, max_distance, out);
    return;
  }
  if (HasherChain(s)) {
    FindLongestMatchChain(s, cur_ix, max_length, max_backward, dictionary_distance, max_distance, out);
    return;
  }
  out->len = 0;
  out->len_code_delta = 0;
  for (i = 0; i < (size_t)s->ndist; ++i) {
    const size_t backward = (size_t)s->dist_cache[i];
    size_t prev_ix = cur_ix - backward;
    if (prev_ix >= cur_ix) continue;
    if (backward > max_backward) continue;
    prev_ix &= mask;
    if (cur_ix_masked + best_len > mask) break;
    if (prev_ix + best_len > mask ||
        data[cur_ix_masked + best_len] != data[prev_ix + best_len]) continue;
    {
      const size_t len = FindMatchLength(&data[prev_ix], &data[cur_ix_masked], max_length);
      if (len >= 3 || (len == 2 && i < 2)) {
        size_t score = ScoreLast(len);
        if (best_score < score) {
          if (i != 0) score -= PenaltyLast(i);
          if (best_score < score) {
            best_score = score; best_len = len;
            out->len = len; out->distance = backward; out->score = score;
          }
        }
      }
    }
  }
  /* All four hashers raise best_len to 3 so the 4-byte gate is always valid
     (..64_simd_inc.h:243-245, ..64_inc.h:219-221, .._inc.h). */
  if (best_len < 3) best_len = 3;
  if (HasherTagged(s)) {
    const uint8_t tag = (uint8_t)(kt >> 32);
    const uint8_t* tag_bucket = &s->tags[key << s->block_bits];
    const size_t head = (s->num[key] + 1u) & bmask;
    const uint16_t n = (uint16_t)(65535 - s->num[key]);
    const uint32_t first4 = Load32(data + cur_ix_masked);
    size_t t;
    for (t = 0; t < bsize; ++t) {
      /* bit t of the rotated tag mask (matching_tag_mask.h:16-65) masked by
         the "unused slots" mask (:250-257). */
      const size_t rb_index = (head + t) & bmask;
      size_t prev_ix, backward, len, score;
      if (bsize > n && t >= n) break;
      if (tag_bucket[rb_index] != tag) continue;
      prev_ix = bucket[rb_index];
      backward = cur_ix - prev_ix;
      if (backward > max_backward) break;
      prev_ix &= mask;
      if (cur_ix_masked + best_len > mask) break;
      if (prev_ix + best_len > mask ||
          Load32(&data[cur_ix_masked + best_len - 3]) !=
          Load32(&data[prev_ix + best_len - 3])) continue;
      if (is64) {
        if (first4 != Load32(data + prev_ix)) continue;
        len = FindMatchLength(&data[prev_ix + 4], &data[cur_ix_masked + 4],
                              max_length - 4) + 4;
      } else {
        len = FindMatchLength(&data[prev_ix], &data[cur_ix_masked], max_length);
        if (len < 4) continue;
      }
      score = ScoreNormal(len, backward);
      if (best_score < score) {
        best_score = score; best_len = len;
        out->len = len; out->distance = backward; out->score = score;
      }
    }
    bucket[s->num[key] & bmask] = (uint32_t)cur_ix;
    s->tags[(key << s->block_bits) + (s->num[key] & bmask)] = tag;
    --s->num[key];
  } else {
    const size_t down = (s->num[key] > bsize) ? (s->num[key] - bsize) : 0u;
    const uint32_t first4 = Load32(data + cur_ix_masked);
    for (i = s->num[key]; i > down;) {
      size_t prev_ix = bucket[--i & bmask];
      const size_t backward = cur_ix - prev_ix;
      size_t len, score;
      if (backward > max_backward) break;
      prev_ix &= mask;
      if (cur_ix_masked + best_len > mask) break;
      if (prev_ix + best_len > mask ||
          Load32(&data[cur_ix_masked + best_len - 3]) !=
          Load32(&data[prev_ix + best_len - 3])) continue;
      if (is64) {
        if (first4 != Load32(data + prev_ix)) continue;
        len = FindMatchLength(&data[prev_ix + 4], &data[cur_ix_masked + 4],
                              max_length - 4) + 4;
      } else {
        len = FindMatchLength(&data[prev_ix], &data[cur_ix_masked], max_length);
        if (len < 4) continue;
      }
      score = ScoreNormal(len, backward);
      if (best_score < score) {
        best_score = score; best_len = len;
        out->len = len; out->distance = backward; out->score = score;
      }
    }
    bucket[s->num[key] & bmask] = (uint32_t)cur_ix;
    ++s->num[key];
  }
  if (min_score == out->score) {
    SearchInStaticDictionary(s, &data[cur_ix_masked], max_length,
                             dictionary_distance, max_distance, out);
  }
}

/* c/enc/hash.h:80-100 */
static void PrepareDistanceCache(int* dc, int n) {
  if (n > 4) {
    int last = dc[0];
    dc[4] = last - 1; dc[5] = last + 1; dc[6] = last - 2;
    dc[7] = last + 2; dc[8] = last - 3; dc[9] = last + 3;
    if (n > 10) {
      int nl = dc[1];
      dc[10] = nl - 1; dc[11] = nl + 1; dc[12] = nl - 2;
      dc[13] = nl + 2; dc[14] = nl - 3; dc[15] = nl + 3;
    }
  }
}

/* c/enc/backward_references.c:87-109 */
static size_t ComputeDistanceCode(size_t distance, size_t max_distance, const int* dc) {
  if (distance <= max_distance) {
    size_t dp3 = distance + 3;
    size_t o0 = dp3 - (size_t)dc[0];
    size_t o1 = dp3 - (size_t)dc[1];
    if (distance == (size_t)dc[0]) return 0;
    if (distance == (size_t)dc[1]) return 1;
    if (o0 < 7) return (0x9750468 >> (4 * o0)) & 0xF;
    if (o1 < 7) return (0xFDB1ACE >> (4 * o1)) & 0xF;
    if (distance == (size_t)dc[2]) return 2;
    if (distance == (size_t)dc[3]) return 3;
  }
  return distance + 16 - 1;
}

/* c/enc/command.h:31-88 */
static uint16_t InsertLengthCode(size_t n) {
  if (n < 6) return (uint16_t)n;
  if (n < 130) { uint32_t nb = Log2Floor(n - 2) - 1u; return (uint16_t)((nb << 1) + ((n - 2) >> nb) + 2); }
  if (n < 2114) return (uint16_t)(Log2Floor(n - 66) + 10);
  if (n < 6210) return 21u;
  if (n < 22594) return 22u;
  return 23u;
}
static uint16_t CopyLengthCode(size_t n) {
  if (n < 10) return (uint16_t)(n - 2);
  if (n < 134) { uint32_t nb = Log2Floor(n - 6) - 1u; return (uint16_t)((nb << 1) + ((n - 6) >> nb) + 4); }
  if (n < 2118) return (uint16_t)(Log2Floor(n - 70) + 12);
  return 23u;
}
static uint16_t CombineLengthCodes(uint16_t ins, uint16_t cpy, int use_last) {
  uint16_t bits64 = (uint16_t)((cpy & 7u) | ((ins & 7u) << 3u));
  if (use_last && ins < 8u && cpy < 16u) {
    return (cpy < 8u) ? bits64 : (uint16_t)(bits64 | 64u);
  } else {
    uint32_t offset = 2u * ((cpy >> 3u) + 3u * (ins >> 3u));
    offset = (offset << 5u) + 0x40u + ((0x520D40u >> offset) & 0xC0u);
    return (uint16_t)(offset | bits64);
  }
}
static const uint32_t kInsBase[24] = {0, 1, 2, 3, 4, 5, 6, 8, 10, 14, 18, 26,
    34, 50, 66, 98, 130, 194, 322, 578, 1090, 2114, 6210, 22594};
static const uint32_t kInsExtra[24] = {0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4,
    5, 5, 6, 7, 8, 9, 10, 12, 14, 24};
static const uint32_t kCopyBase[24] = {2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 18,
    22, 30, 38, 54, 70, 102, 134, 198, 326, 582, 1094, 2118};
static const uint32_t kCopyExtra[24] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3,
    4, 4, 5, 5, 6, 7, 8, 9, 10, 24};

/* c/enc/prefix.h:23-46 with NPOSTFIX = NDIRECT = 0 */
static void PrefixEncodeCopyDistance(size_t distance_code, uint16_t* code, uint32_t* extra) {
  if (distance_code < 16) { *code = (uint16_t)distance_code; *extra = 0; return; }
  {
    size_t dist = ((size_t)1 << 2) + (distance_code - 16);
    size_t bucket = Log2Floor(dist) - 1;
    size_t prefix = (dist >> bucket) & 1;
    size_t offset = (2 + prefix) << bucket;
    size_t nbits = bucket;
    *code = (uint16_t)((nbits << 10) | (16 + ((2 * (nbits - 1) + prefix))));
    *extra = (uint32_t)(dist - offset);
  }
}

/* c/enc/command.h:118-143 */
static void InitCommand(Cmd* c, size_t insertlen, size_t copylen, int delta, size_t distance_code) {
  uint32_t d = (uint8_t)((int8_t)delta);
  c->insert_len = (uint32_t)insertlen;
  c->copy_len = (uint32_t)(copylen | (d << 25));
  PrefixEncodeCopyDistance(distance_code, &c->dist_prefix, &c->dist_extra);
  c->cmd_prefix = CombineLengthCodes(InsertLengthCode(insertlen),
      CopyLengthCode((size_t)((int)copylen + delta)), (c->dist_prefix & 0x3FF) == 0);
}
static void InitInsertCommand(Cmd* c, size_t insertlen) {
  c->insert_len = (uint32_t)insertlen;
  c->copy_len = 4u << 25;
  c->dist_extra = 0;
  c->dist_prefix = 16;
  c->cmd_prefix = CombineLengthCodes(InsertLengthCode(insertlen), CopyLengthCode(4), 0);
}
static uint32_t CmdCopyLen(const Cmd* c) { return c->copy_len & 0x1FFFFFF; }
static uint32_t CmdCopyLenCode(const Cmd* c) {
  uint32_t m = c->copy_len >> 25;
  int32_t delta = (int8_t)((uint8_t)(m | ((m & 0x40) << 1)));
  return (uint32_t)((int32_t)(c->copy_len & 0x1FFFFFF) + delta);
}

/* c/enc/backward_references_inc.h:10-242 (base64 mode and compound
   dictionaries are off: encode.h:69, ENABLE_COMPOUND_DICTIONARY 0). */

/* ------------------------------------------------------------------------ */
/* Attached (compound) dictionaries: raw LZ77 prefixes prepared by
   BrotliEncoderPrepareDictionary(BROTLI_SHARED_DICTIONARY_RAW) and attached with
   BrotliEncoderAttachPreparedDictionary (c/enc/encode.c:1756-1880).            */

typedef struct {
  uint32_t source_size, bucket_bits, slot_bits, hash_bits, num_items;
  uint32_t* slot_offsets;   /* [1 << slot_bits] */
  uint16_t* heads;          /* [1 << bucket_bits] */
  uint32_t* items;          /* [num_items], bit 31 ends a chain */
  const uint8_t* source;
} PDict;

static struct {
  size_t num_chunks, total_size;
  PDict chunks[16];
  size_t chunk_offsets[17];
} g_cd;

static const uint64_t kPDictMul = 0x1FE35A7BD3579BD3ull; /* compound_dictionary.h:31-32 */

/* c/enc/compound_dictionary.c:13-153 (CreatePreparedDictionaryWithParams) with the parameters
   of CreatePreparedDictionary (:155-173). */
static int PDictCreate(PDict* d, const uint8_t* source, size_t source_size) {
  uint32_t bucket_bits = 17, slot_bits = 7;
  const uint32_t hash_bits = 40;
  const uint16_t bucket_limit = 32;
  size_t volume = (size_t)16 << bucket_bits;
  uint32_t num_slots, num_buckets, hash_shift, slot_mask, i, total_items = 0;
  uint64_t hash_mask;
  uint16_t* num;
  uint32_t *bucket_heads, *next_bucket, *slot_size, *slot_limit;
  while (volume < source_size && bucket_bits < 22) { bucket_bits++; slot_bits++; volume <<= 1; }
  num_slots = 1u << slot_bits;
  num_buckets = 1u << bucket_bits;
  hash_shift = 64u - bucket_bits;
  hash_mask = (~(uint64_t)0) >> (64 - hash_bits);
  slot_mask = num_slots - 1;
  slot_size = (uint32_t*)calloc(num_slots, 4);
  slot_limit = (uint32_t*)calloc(num_slots, 4);
  num = (uint16_t*)calloc(num_buckets, 2);
  bucket_heads = (uint32_t*)calloc(num_buckets, 4);
  next_bucket = (uint32_t*)calloc(source_size + 1, 4);
  for (i = 0; (size_t)i + 7 < source_size; ++i) {
    const uint64_t h = (Load64(&source[i]) & hash_mask) * kPDictMul;
    const uint32_t key = (uint32_t)(h >> hash_shift);
    uint16_t count = num[key];
    next_bucket[i] = count == 0 ? (uint32_t)-1 : bucket_heads[key];
    bucket_heads[key] = i;
    count++;
    if (count > bucket_limit) count = bucket_limit;
    num[key] = count;
  }
  for (i = 0; i < num_slots; ++i) {
    slot_limit[i] = bucket_limit;
    for (;;) {
      uint32_t limit = slot_limit[i], count = 0;
      size_t j;
      int overflow = 0;
      for (j = i; j < num_buckets; j += num_slots) {
        uint32_t size = num[j];
        if (count >= 0xFFFF) { overflow = 1; break; }
        if (size > limit) size = limit;
        count += size;
      }
      if (!overflow) { slot_size[i] = count; total_items += count; break; }
      slot_limit[i]--;
    }
  }
  d->source_size = (uint32_t)source_size;
  d->bucket_bits = bucket_bits;
  d->slot_bits = slot_bits;
  d->hash_bits = hash_bits;
  d->num_items = total_items;
  d->slot_offsets = (uint32_t*)calloc(num_slots, 4);
  d->heads = (uint16_t*)calloc(num_buckets, 2);
  d->items = (uint32_t*)calloc(total_items + 1, 4);
  d->source = source;
  total_items = 0;
  for (i = 0; i < num_slots; ++i) {
    d->slot_offsets[i] = total_items;
    total_items += slot_size[i];
    slot_size[i] = 0;
  }
  for (i = 0; i < num_buckets; ++i) {
    uint32_t slot = i & slot_mask, count = num[i], pos;
    size_t j, cursor = slot_size[slot];
    if (count > slot_limit[slot]) count = slot_limit[slot];
    if (count == 0) { d->heads[i] = 0xFFFF; continue; }
    d->heads[i] = (uint16_t)cursor;
    cursor += d->slot_offsets[slot];
    slot_size[slot] += count;
    pos = bucket_heads[i];
    for (j = 0; j < count; j++) { d->items[cursor++] = pos; pos = next_bucket[pos]; }
    d->items[cursor - 1] |= 0x80000000u;
  }
  free(slot_size); free(slot_limit); free(num); free(bucket_heads); free(next_bucket);
  return 1;
}

/* AttachPreparedDictionary, compound_dictionary.c:182-211; n == 0 detaches everything.  The
   sources must stay alive while encodes run (the "lean" form references them). */
/* BROTLI_PARAM_LGBLOCK of every encoder instance created from now on (0 = the default), encode.h:190-197. */
void oracle_set_lgblock(int lgblock) { g_lgblock_param = lgblock; }

int oracle_set_dictionary(const uint8_t* const* sources, const size_t* sizes, size_t n) {
  size_t i;
  for (i = 0; i < g_cd.num_chunks; ++i) {
    free(g_cd.chunks[i].slot_offsets); free(g_cd.chunks[i].heads); free(g_cd.chunks[i].items);
  }
  memset(&g_cd, 0, sizeof(g_cd));
  if (n > 15) return 0;
  for (i = 0; i < n; ++i) {
    if (sizes[i] > (size_t)0x7FFFFFFF - g_cd.total_size) return 0;   /* SHARED_BROTLI_MAX_RAW_DICT_SIZE */
    PDictCreate(&g_cd.chunks[i], sources[i], sizes[i]);
    g_cd.total_size += sizes[i];
    g_cd.chunk_offsets[i + 1] = g_cd.total_size;
    g_cd.num_chunks++;
  }
  return 1;
}

/* The hashers that have a compound-dictionary variant (backward_references.c:194-243, 256-282);
   H2 and H54 fall through to the plain variant, where only `gap` applies. */
static int HasherLooksUpCompound(int t) {
  return t == 3 || t == 4 || t == 5 || t == 6 || t == 40 || t == 41 || t == 42 || t == 58 || t == 68;
}

/* FindCompoundDictionaryMatch, c/enc/hash.h:526-634 */
static void FindCompoundDictionaryMatch(const PDict* self, const uint8_t* data, size_t ring_buffer_mask,
    const int* distance_cache, size_t cur_ix, size_t max_length, size_t distance_offset,
    size_t max_distance, SearchResult* out) {
  const uint32_t source_size = self->source_size;
  const size_t boundary = distance_offset - source_size;
  const uint32_t hash_shift = 64u - self->bucket_bits;
  const uint32_t slot_mask = (~(uint32_t)0) >> (32 - self->slot_bits);
  const uint64_t hash_mask = (~(uint64_t)0) >> (64 - self->hash_bits);
  const uint8_t* source = self->source;
  const size_t cur_ix_masked = cur_ix & ring_buffer_mask;
  size_t best_score = out->score, best_len = out->len, i;
  const uint64_t h = (Load64(&data[cur_ix_masked]) & hash_mask) * kPDictMul;
  const uint32_t key = (uint32_t)(h >> hash_shift);
  const uint32_t slot = key & slot_mask;
  const uint32_t head = self->heads[key];
  const uint32_t* chain = &self->items[self->slot_offsets[slot] + head];
  uint32_t item = head == 0xFFFF ? 1 : 0;
  for (i = 0; i < 4; ++i) {
    const size_t distance = (size_t)distance_cache[i];
    size_t offset, limit, len;
    if (distance <= boundary || distance > distance_offset) continue;
    offset = distance_offset - distance;
    limit = source_size - offset;
    limit = limit > max_length ? max_length : limit;
    len = FindMatchLength(&source[offset], &data[cur_ix_masked], limit);
    if (len >= 2) {
      size_t score = ScoreLast(len);
      if (best_score < score) {
        if (i != 0) score -= PenaltyLast(i);
        if (best_score < score) {
          best_score = score;
          if (len > best_len) best_len = len;
          out->len = len; out->len_code_delta = 0; out->distance = distance; out->score = best_score;
        }
      }
    }
  }
  if (best_len < 3) best_len = 3;
  while (item == 0) {
    size_t offset, distance, limit;
    item = *chain++;
    offset = item & 0x7FFFFFFF;
    item &= 0x80000000u;
    distance = distance_offset - offset;
    limit = source_size - offset;
    limit = limit > max_length ? max_length : limit;
    if (distance > max_distance) continue;
    if (cur_ix_masked + best_len > ring_buffer_mask || best_len >= limit ||
        Load32(&data[cur_ix_masked + best_len - 3]) != Load32(&source[offset + best_len - 3])) continue;
    {
      const size_t len = FindMatchLength(&source[offset], &data[cur_ix_masked], limit);
      if (len >= 4) {
        size_t score = ScoreNormal(len, distance);
        if (best_score < score) {
          best_score = score; best_len = len;
          out->len = best_len; out->len_code_delta = 0; out->distance = distance; out->score = best_score;
        }
      }
    }
  }
}

/* LookupCompoundDictionaryMatch, c/enc/hash.h:703-717 */
static void LookupCompoundDictionaryMatch(Enc* s, size_t cur_ix, size_t max_length,
    size_t max_ring_buffer_distance, size_t max_distance, SearchResult* sr) {
  size_t base_offset = max_ring_buffer_distance + 1 + g_cd.total_size - 1, d;
  if (!HasherLooksUpCompound(s->hasher_type)) return;
  for (d = 0; d < g_cd.num_chunks; ++d) {
    FindCompoundDictionaryMatch(&g_cd.chunks[d], s->rb, s->rb_mask, s->dist_cache, cur_ix, max_length,
        base_offset - g_cd.chunk_offsets[d], max_distance, sr);
  }
}

static void CreateBackwardReferences(Enc* s, size_t num_bytes, size_t position) {
  const size_t max_backward_limit = ((size_t)1 << s->lgwin) - 16;
  const size_t position_offset = s->stream_offset;
  const size_t htl = HashTypeLength(s); /* == StoreLookahead for all four */
  Cmd* commands = s->cmds + s->ncmds;
  const Cmd* const orig = commands;
  size_t insert_length = s->last_insert_len;
  const size_t pos_end = position + num_bytes;
  const size_t store_end = num_bytes >= htl ? position + num_bytes - htl + 1 : position;
  const size_t window = s->quality < 9 ? 64 : 512; /* quality.h:116-119 */
  size_t apply_random_heuristics = position + window;
  const size_t kMinScore = SCORE_BASE + 100;
  const size_t dist_max_distance = 0x3FFFFFC; /* metablock.c:190-191 */
  int* dc = s->dist_cache;
  const size_t gap = g_cd.total_size;   /* backward_references_inc.h:31 */
  PrepareDistanceCache(dc, s->ndist);
  while (position + htl < pos_end) {
    size_t max_length = pos_end - position;
    size_t max_distance = position < max_backward_limit ? position : max_backward_limit;
    size_t dictionary_start = position + position_offset < max_backward_limit ?
        position + position_offset : max_backward_limit;
    SearchResult sr;
    sr.len = 0; sr.len_code_delta = 0; sr.distance = 0; sr.score = kMinScore;
    FindLongestMatch(s, position, max_length, max_distance, dictionary_start + gap,
                     dist_max_distance, &sr);
    if (g_cd.num_chunks) LookupCompoundDictionaryMatch(s, position, max_length, dictionary_start,
                                                       dist_max_distance, &sr);
    if (sr.score > kMinScore) {
      int delayed = 0;
      --max_length;
      for (;; --max_length) {
        SearchResult sr2;
        sr2.len = s->quality < 5 ? (sr.len - 1 < max_length ? sr.len - 1 : max_length) : 0; /* :127-128 */
        sr2.len_code_delta = 0; sr2.distance = 0; sr2.score = kMinScore;
        max_distance = position + 1 < max_backward_limit ? position + 1 : max_backward_limit;
        dictionary_start = position + 1 + position_offset < max_backward_limit ?
            position + 1 + position_offset : max_backward_limit;
        FindLongestMatch(s, position + 1, max_length, max_distance,
                         dictionary_start + gap, dist_max_distance, &sr2);
        if (g_cd.num_chunks) LookupCompoundDictionaryMatch(s, position + 1, max_length, dictionary_start,
                                                           dist_max_distance, &sr2);
        if (sr2.score >= sr.score + 175) {
          ++position; ++insert_length; sr = sr2;
          if (++delayed < 4 && position + htl < pos_end) continue;
        }
        break;
      }
      apply_random_heuristics = position + 2 * sr.len + window;
      dictionary_start = position + position_offset < max_backward_limit ?
          position + position_offset : max_backward_limit;
      {
        size_t distance_code = ComputeDistanceCode(sr.distance, dictionary_start + gap, dc);
        if (sr.distance <= dictionary_start + gap && distance_code > 0) {
          dc[3] = dc[2]; dc[2] = dc[1]; dc[1] = dc[0]; dc[0] = (int)sr.distance;
          PrepareDistanceCache(dc, s->ndist);
        }
        InitCommand(commands++, insert_length, sr.len, sr.len_code_delta, distance_code);
      }
      s->nlits += insert_length;
      insert_length = 0;
      {
        size_t range_start = position + 2;
        size_t range_end = position + sr.len < store_end ? position + sr.len : store_end;
        size_t i;
        if (sr.distance < (sr.len >> 2)) {
          size_t a = position + sr.len - (sr.distance << 2);
          size_t m = range_start > a ? range_start : a;
          range_start = range_end < m ? range_end : m;
        }
        for (i = range_start; i < range_end; ++i) HStore(s, i);
      }
      position += sr.len;
    } else {
      ++insert_length;
      ++position;
      if (position > apply_random_heuristics) {
        if (position > apply_random_heuristics + 4 * window) {
          const size_t kMargin = htl - 1 > 4 ? htl - 1 : 4;
          size_t pos_jump = position + 16 < pos_end - kMargin ? position + 16 : pos_end - kMargin;
          for (; position < pos_jump; position += 4) { HStore(s, position); insert_length += 4; }
        } else {
          const size_t kMargin = htl - 1 > 2 ? htl - 1 : 2;
          size_t pos_jump = position + 8 < pos_end - kMargin ? position + 8 : pos_end - kMargin;
          for (; position < pos_jump; position += 2) { HStore(s, position); insert_length += 2; }
        }
      }
    }
  }
  insert_length += pos_end - position;
  s->last_insert_len = insert_length;
  s->ncmds += (size_t)(commands - orig);
}

/* ------------------------------------------------------------------------ */
/* Entropy primitives.                                                       */

/* c/enc/bit_cost.c:18-44 (summation order kept) */
static double BitsEntropy(const uint32_t* population, size_t size) {
  size_t sum = 0, i;
  double retval = 0;
  for (i = 0; i < size; ++i) {
    size_t p = population[i];
    sum += p;
    retval -= (double)p * FastLog2(p);
  }
  if (sum) retval += (double)sum * FastLog2(sum);
  if (retval < (double)sum) retval = (double)sum;
  return retval;
}

typedef struct { uint32_t total_count; int16_t left; int16_t right_or_value; } HTree;

/* c/enc/entropy_encode.c:20-42 */
static int SetDepth(int p0, HTree* pool, uint8_t* depth, int max_depth) {
  int stack[16];
  int level = 0;
  int p = p0;
  stack[0] = -1;
  for (;;) {
    if (pool[p].left >= 0) {
      level++;
      if (level > max_depth) return 0;
      stack[level] = pool[p].right_or_value;
      p = pool[p].left;
      continue;
    } else {
      depth[pool[p].right_or_value] = (uint8_t)level;
    }
    while (level >= 0 && stack[level] == -1) level--;
    if (level < 0) return 1;
    p = stack[level];
    stack[level] = -1;
  }
}

static int TreeLess(const HTree* a, const HTree* b) {
  if (a->total_count != b->total_count) return a->total_count < b->total_count;
  return a->right_or_value > b->right_or_value;
}

/* c/enc/entropy_encode.h:82-115 */
static void SortTree(HTree* items, size_t n) {
  static const size_t gaps[] = {132, 57, 23, 10, 4, 1};
  if (n < 13) {
    size_t i;
    for (i = 1; i < n; ++i) {
      HTree tmp = items[i];
      size_t k = i, j = i - 1;
      while (TreeLess(&tmp, &items[j])) {
        items[k] = items[j];
        k = j;
        if (!j--) break;
      }
      items[k] = tmp;
    }
  } else {
    int g = n < 57 ? 2 : 0;
    for (; g < 6; ++g) {
      size_t gap = gaps[g], i;
      for (i = gap; i < n; ++i) {
        size_t j = i;
        HTree tmp = items[i];
        for (; j >= gap && TreeLess(&tmp, &items[j - gap]); j -= gap) items[j] = items[j - gap];
        items[j] = tmp;
      }
    }
  }
}

/* c/enc/entropy_encode.c:68-147 */
static void CreateHuffmanTree(const uint32_t* data, size_t length, int tree_limit,
                              HTree* tree, uint8_t* depth) {
  uint32_t count_limit;
  HTree sentinel;
  sentinel.total_count = 0xFFFFFFFFu; sentinel.left = -1; sentinel.right_or_value = -1;
  for (count_limit = 1;; count_limit *= 2) {
    size_t n = 0, i, j, k;
    for (i = length; i != 0;) {
      --i;
      if (data[i]) {
        uint32_t count = data[i] > count_limit ? data[i] : count_limit;
        tree[n].total_count = count; tree[n].left = -1; tree[n].right_or_value = (int16_t)i;
        ++n;
      }
    }
    if (n == 1) { depth[tree[0].right_or_value] = 1; break; }
    SortTree(tree, n);
    tree[n] = sentinel;
    tree[n + 1] = sentinel;
    i = 0; j = n + 1;
    for (k = n - 1; k != 0; --k) {
      size_t left, right;
      if (tree[i].total_count <= tree[j].total_count) { left = i; ++i; } else { left = j; ++j; }
      if (tree[i].total_count <= tree[j].total_count) { right = i; ++i; } else { right = j; ++j; }
      {
        size_t j_end = 2 * n - k;
        tree[j_end].total_count = tree[left].total_count + tree[right].total_count;
        tree[j_end].left = (int16_t)left;
        tree[j_end].right_or_value = (int16_t)right;
        tree[j_end + 1] = sentinel;
      }
    }
    if (SetDepth((int)(2 * n - 1), tree, depth, tree_limit)) break;
  }
}

static void Reverse(uint8_t* v, size_t start, size_t end) {
  --end;
  while (start < end) { uint8_t t = v[start]; v[start] = v[end]; v[end] = t; ++start; --end; }
}

/* c/enc/entropy_encode.c:160-239 */
static void WriteTreeReps(uint8_t prev, uint8_t value, size_t reps, size_t* n,
                          uint8_t* tree, uint8_t* extra) {
  if (prev != value) { tree[*n] = value; extra[*n] = 0; ++*n; --reps; }
  if (reps == 7) { tree[*n] = value; extra[*n] = 0; ++*n; --reps; }
  if (reps < 3) {
    size_t i;
    for (i = 0; i < reps; ++i) { tree[*n] = value; extra[*n] = 0; ++*n; }
  } else {
    size_t start = *n;
    reps -= 3;
    for (;;) {
      tree[*n] = 16; extra[*n] = reps & 3; ++*n;
      reps >>= 2;
      if (reps == 0) break;
      --reps;
    }
    Reverse(tree, start, *n);
    Reverse(extra, start, *n);
  }
}
static void WriteTreeRepsZeros(size_t reps, size_t* n, uint8_t* tree, uint8_t* extra) {
  if (reps == 11) { tree[*n] = 0; extra[*n] = 0; ++*n; --reps; }
  if (reps < 3) {
    size_t i;
    for (i = 0; i < reps; ++i) { tree[*n] = 0; extra[*n] = 0; ++*n; }
  } else {
    size_t start = *n;
    reps -= 3;
    for (;;) {
      tree[*n] = 17; extra[*n] = reps & 7; ++*n;
      reps >>= 3;
      if (reps == 0) break;
      --reps;
    }
    Reverse(tree, start, *n);
    Reverse(extra, start, *n);
  }
}

/* c/enc/entropy_encode.c:241-370 */
static void OptimizeHuffmanCountsForRle(size_t length, uint32_t* counts, uint8_t* good_for_rle) {
  size_t nonzero_count = 0, stride, limit, sum, i;
  const size_t streak_limit = 1240;
  for (i = 0; i < length; i++) if (counts[i]) ++nonzero_count;
  if (nonzero_count < 16) return;
  while (length != 0 && counts[length - 1] == 0) --length;
  if (length == 0) return;
  {
    size_t nonzeros = 0;
    uint32_t smallest_nonzero = 1 << 30;
    for (i = 0; i < length; ++i) {
      if (counts[i] != 0) {
        ++nonzeros;
        if (smallest_nonzero > counts[i]) smallest_nonzero = counts[i];
      }
    }
    if (nonzeros < 5) return;
    if (smallest_nonzero < 4) {
      size_t zeros = length - nonzeros;
      if (zeros < 6) {
        for (i = 1; i < length - 1; ++i) {
          if (counts[i - 1] != 0 && counts[i] == 0 && counts[i + 1] != 0) counts[i] = 1;
        }
      }
    }
    if (nonzeros < 28) return;
  }
  memset(good_for_rle, 0, length);
  {
    uint32_t symbol = counts[0];
    size_t step = 0;
    for (i = 0; i <= length; ++i) {
      if (i == length || counts[i] != symbol) {
        if ((symbol == 0 && step >= 5) || (symbol != 0 && step >= 7)) {
          size_t k;
          for (k = 0; k < step; ++k) good_for_rle[i - k - 1] = 1;
        }
        step = 1;
        if (i != length) symbol = counts[i];
      } else {
        ++step;
      }
    }
  }
  stride = 0;
  limit = 256 * (counts[0] + counts[1] + counts[2]) / 3 + 420;
  sum = 0;
  for (i = 0; i <= length; ++i) {
    if (i == length || good_for_rle[i] || (i != 0 && good_for_rle[i - 1]) ||
        (256 * counts[i] - limit + streak_limit) >= 2 * streak_limit) {
      if (stride >= 4 || (stride >= 3 && sum == 0)) {
        size_t k;
        size_t count = (sum + stride / 2) / stride;
        if (count == 0) count = 1;
        if (sum == 0) count = 0;
        for (k = 0; k < stride; ++k) counts[i - k - 1] = (uint32_t)count;
      }
      stride = 0;
      sum = 0;
      if (i < length - 2) {
        limit = 256 * (counts[i] + counts[i + 1] + counts[i + 2]) / 3 + 420;
      } else if (i < length) {
        limit = 256 * counts[i];
      } else {
        limit = 0;
      }
    }
    ++stride;
    if (i != length) {
      sum += counts[i];
      if (stride >= 4) limit = (256 * sum + stride / 2) / stride;
      if (stride == 4) limit += 120;
    }
  }
}

/* c/enc/entropy_encode.c:372-452 */
static void WriteHuffmanTree(const uint8_t* depth, size_t length, size_t* tree_size,
                             uint8_t* tree, uint8_t* extra) {
  uint8_t previous_value = 8;
  size_t i;
  int rle_nz = 0, rle_z = 0;
  size_t new_length = length;
  for (i = 0; i < length; ++i) {
    if (depth[length - i - 1] == 0) --new_length; else break;
  }
  if (length > 50) {
    size_t total_z = 0, total_nz = 0, cnt_z = 1, cnt_nz = 1;
    for (i = 0; i < new_length;) {
      const uint8_t value = depth[i];
      size_t reps = 1, k;
      for (k = i + 1; k < new_length && depth[k] == value; ++k) ++reps;
      if (reps >= 3 && value == 0) { total_z += reps; ++cnt_z; }
      if (reps >= 4 && value != 0) { total_nz += reps; ++cnt_nz; }
      i += reps;
    }
    rle_nz = total_nz > cnt_nz * 2;
    rle_z = total_z > cnt_z * 2;
  }
  for (i = 0; i < new_length;) {
    const uint8_t value = depth[i];
    size_t reps = 1;
    if ((value != 0 && rle_nz) || (value == 0 && rle_z)) {
      size_t k;
      for (k = i + 1; k < new_length && depth[k] == value; ++k) ++reps;
    }
    if (value == 0) {
      WriteTreeRepsZeros(reps, tree_size, tree, extra);
    } else {
      WriteTreeReps(previous_value, value, reps, tree_size, tree, extra);
      previous_value = value;
    }
    i += reps;
  }
}

/* c/enc/entropy_encode.c:454-497 */
static uint16_t ReverseBits(size_t num_bits, uint16_t bits) {
  static const size_t kLut[16] = {0x00, 0x08, 0x04, 0x0C, 0x02, 0x0A, 0x06, 0x0E,
      0x01, 0x09, 0x05, 0x0D, 0x03, 0x0B, 0x07, 0x0F};
  size_t retval = kLut[bits & 0x0F];
  size_t i;
  for (i = 4; i < num_bits; i += 4) {
    retval <<= 4;
    bits = (uint16_t)(bits >> 4);
    retval |= kLut[bits & 0x0F];
  }
  retval >>= ((0 - num_bits) & 0x03);
  return (uint16_t)retval;
}
static void ConvertBitDepthsToSymbols(const uint8_t* depth, size_t len, uint16_t* bits) {
  uint16_t bl_count[16] = {0};
  uint16_t next_code[16];
  size_t i;
  int code = 0;
  for (i = 0; i < len; ++i) ++bl_count[depth[i]];
  bl_count[0] = 0;
  next_code[0] = 0;
  for (i = 1; i < 16; ++i) {
    code = (code + bl_count[i - 1]) << 1;
    next_code[i] = (uint16_t)code;
  }
  for (i = 0; i < len; ++i) if (depth[i]) bits[i] = ReverseBits(depth[i], next_code[depth[i]]++);
}

/* c/enc/brotli_bit_stream.c:163-345 */
static void StoreHuffmanTree(const uint8_t* depths, size_t num, HTree* tree,
                             size_t* ix, uint8_t* storage) {
  static const uint8_t kStorageOrder[18] = {1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15};
  static const uint8_t kSym[6] = {0, 7, 3, 2, 1, 15};
  static const uint8_t kLen[6] = {2, 4, 3, 2, 2, 4};
  uint8_t huffman_tree[704];
  uint8_t extra_bits[704];
  size_t huffman_tree_size = 0;
  uint8_t cl_depth[18] = {0};
  uint16_t cl_bits[18];
  uint32_t histogram[18] = {0};
  size_t i;
  int num_codes = 0;
  size_t code = 0;
  WriteHuffmanTree(depths, num, &huffman_tree_size, huffman_tree, extra_bits);
  for (i = 0; i < huffman_tree_size; ++i) ++histogram[huffman_tree[i]];
  for (i = 0; i < 18; ++i) {
    if (histogram[i]) {
      if (num_codes == 0) { code = i; num_codes = 1; }
      else if (num_codes == 1) { num_codes = 2; break; }
    }
  }
  CreateHuffmanTree(histogram, 18, 5, tree, cl_depth);
  ConvertBitDepthsToSymbols(cl_depth, 18, cl_bits);
  {
    size_t skip_some = 0;
    size_t codes_to_store = 18;
    if (num_codes > 1) {
      for (; codes_to_store > 0; --codes_to_store) {
        if (cl_depth[kStorageOrder[codes_to_store - 1]] != 0) break;
      }
    }
    if (cl_depth[kStorageOrder[0]] == 0 && cl_depth[kStorageOrder[1]] == 0) {
      skip_some = 2;
      if (cl_depth[kStorageOrder[2]] == 0) skip_some = 3;
    }
    WriteBits(2, skip_some, ix, storage);
    for (i = skip_some; i < codes_to_store; ++i) {
      size_t l = cl_depth[kStorageOrder[i]];
      WriteBits(kLen[l], kSym[l], ix, storage);
    }
  }
  if (num_codes == 1) cl_depth[code] = 0;
  for (i = 0; i < huffman_tree_size; ++i) {
    size_t v = huffman_tree[i];
    WriteBits(cl_depth[v], cl_bits[v], ix, storage);
    if (v == 16) WriteBits(2, extra_bits[i], ix, storage);
    else if (v == 17) WriteBits(3, extra_bits[i], ix, storage);
  }
}

/* c/enc/brotli_bit_stream.c:242-279, 349-397 */
static void BuildAndStoreHuffmanTree(const uint32_t* histogram, size_t histogram_length,
    size_t alphabet_size, HTree* tree, uint8_t* depth, uint16_t* bits,
    size_t* ix, uint8_t* storage) {
  size_t count = 0, s4[4] = {0}, i, max_bits = 0;
  for (i = 0; i < histogram_length; i++) {
    if (histogram[i]) {
      if (count < 4) s4[count] = i; else if (count > 4) break;
      count++;
    }
  }
  {
    size_t c = alphabet_size - 1;
    while (c) { c >>= 1; ++max_bits; }
  }
  if (count <= 1) {
    WriteBits(4, 1, ix, storage);
    WriteBits(max_bits, s4[0], ix, storage);
    depth[s4[0]] = 0;
    bits[s4[0]] = 0;
    return;
  }
  memset(depth, 0, histogram_length);
  CreateHuffmanTree(histogram, histogram_length, 15, tree, depth);
  ConvertBitDepthsToSymbols(depth, histogram_length, bits);
  if (count <= 4) {
    size_t j;
    WriteBits(2, 1, ix, storage);
    WriteBits(2, count - 1, ix, storage);
    for (i = 0; i < count; i++) {
      for (j = i + 1; j < count; j++) {
        if (depth[s4[j]] < depth[s4[i]]) { size_t t = s4[j]; s4[j] = s4[i]; s4[i] = t; }
      }
    }
    for (i = 0; i < count; ++i) WriteBits(max_bits, s4[i], ix, storage);
    if (count == 4) WriteBits(1, depth[s4[0]] == 1 ? 1 : 0, ix, storage);
  } else {
    StoreHuffmanTree(depth, histogram_length, tree, ix, storage);
  }
}

/* ------------------------------------------------------------------------ */
/* Greedy meta-block builder.                                                */

typedef struct {
  size_t num_types, num_blocks;
  uint8_t* types;
  uint32_t* lengths;
} Split;

typedef struct {
  size_t alphabet, num_contexts, max_block_types, min_block_size;
  double split_threshold;
  size_t num_blocks;
  Split* split;
  uint32_t* histograms; /* [histograms_size][alphabet] */
  size_t histograms_size;
  size_t target_block_size, block_size, curr_ix, last_ix[2];
  double last_entropy[2 * 13];
  size_t merge_last_count;
} Splitter;

/* InitBlockSplitter (metablock_inc.h:48-82) / InitContextBlockSplitter
   (metablock.c:499-541); num_contexts == 1 gives the plain splitter
   (BROTLI_MAX_NUMBER_OF_BLOCK_TYPES / 1 == 256). */
static void SplitterInit(Splitter* b, size_t alphabet, size_t num_contexts,
    size_t min_block_size, double threshold, size_t num_symbols, Split* split) {
  size_t max_num_blocks = num_symbols / min_block_size + 1;
  size_t max_num_types;
  memset(b, 0, sizeof(*b));
  b->alphabet = alphabet;
  b->num_contexts = num_contexts;
  b->max_block_types = 256 / num_contexts;
  b->min_block_size = min_block_size;
  b->split_threshold = threshold;
  b->split = split;
  b->target_block_size = min_block_size;
  max_num_types = max_num_blocks < b->max_block_types + 1 ? max_num_blocks : b->max_block_types + 1;
  split->types = (uint8_t*)malloc(max_num_blocks);
  split->lengths = (uint32_t*)malloc(max_num_blocks * 4);
  split->num_blocks = max_num_blocks;
  split->num_types = 0;
  b->histograms_size = max_num_types * num_contexts;
  b->histograms = (uint32_t*)calloc(b->histograms_size * alphabet, 4);
}

#define HISTO(b, i) ((b)->histograms + (size_t)(i) * (b)->alphabet)

/* metablock_inc.h:87-173 and metablock.c:543-661 in one routine (they are the
   same algorithm; the context version sums the entropy deltas over contexts). */
static void SplitterFinishBlock(Splitter* b, int is_final) {
  Split* split = b->split;
  const size_t nc = b->num_contexts, A = b->alphabet;
  double* last_entropy = b->last_entropy;
  size_t i;
  if (b->block_size < b->min_block_size) b->block_size = b->min_block_size;
  if (b->num_blocks == 0) {
    split->lengths[0] = (uint32_t)b->block_size;
    split->types[0] = 0;
    for (i = 0; i < nc; ++i) {
      last_entropy[i] = BitsEntropy(HISTO(b, i), A);
      last_entropy[nc + i] = last_entropy[i];
    }
    ++b->num_blocks;
    ++split->num_types;
    b->curr_ix += nc;
    if (b->curr_ix < b->histograms_size) memset(HISTO(b, b->curr_ix), 0, nc * A * 4);
    b->block_size = 0;
  } else if (b->block_size > 0) {
    double entropy[13];
    uint32_t* combined = (uint32_t*)malloc(2 * nc * A * 4);
    double combined_entropy[2 * 13];
    double diff[2] = {0.0, 0.0};
    for (i = 0; i < nc; ++i) {
      size_t cur = b->curr_ix + i, j, k;
      entropy[i] = BitsEntropy(HISTO(b, cur), A);
      for (j = 0; j < 2; ++j) {
        size_t jx = j * nc + i;
        size_t last = b->last_ix[j] + i;
        uint32_t* c = combined + jx * A;
        for (k = 0; k < A; ++k) c[k] = HISTO(b, cur)[k] + HISTO(b, last)[k];
        combined_entropy[jx] = BitsEntropy(c, A);
        diff[j] += combined_entropy[jx] - entropy[i] - last_entropy[jx];
      }
    }
    if (getenv("ORACLE_DEBUG2")) fprintf(stderr, "oracle A=%zu nb=%zu bs=%zu e=%f c0=%f c1=%f l0=%f l1=%f d0=%f d1=%f\n", A, b->num_blocks, b->block_size, entropy[0], combined_entropy[0], combined_entropy[nc], last_entropy[0], last_entropy[nc], diff[0], diff[1]);
    if (split->num_types < b->max_block_types &&
        diff[0] > b->split_threshold && diff[1] > b->split_threshold) {
      split->lengths[b->num_blocks] = (uint32_t)b->block_size;
      split->types[b->num_blocks] = (uint8_t)split->num_types;
      b->last_ix[1] = b->last_ix[0];
      b->last_ix[0] = split->num_types * nc;
      for (i = 0; i < nc; ++i) {
        last_entropy[nc + i] = last_entropy[i];
        last_entropy[i] = entropy[i];
      }
      ++b->num_blocks;
      ++split->num_types;
      b->curr_ix += nc;
      if (b->curr_ix < b->histograms_size) memset(HISTO(b, b->curr_ix), 0, nc * A * 4);
      b->block_size = 0;
      b->merge_last_count = 0;
      b->target_block_size = b->min_block_size;
    } else if (diff[1] < diff[0] - 20.0) {
      size_t t;
      split->lengths[b->num_blocks] = (uint32_t)b->block_size;
      split->types[b->num_blocks] = split->types[b->num_blocks - 2];
      t = b->last_ix[0]; b->last_ix[0] = b->last_ix[1]; b->last_ix[1] = t;
      for (i = 0; i < nc; ++i) {
        memcpy(HISTO(b, b->last_ix[0] + i), combined + (nc + i) * A, A * 4);
        last_entropy[nc + i] = last_entropy[i];
        last_entropy[i] = combined_entropy[nc + i];
        memset(HISTO(b, b->curr_ix + i), 0, A * 4);
      }
      ++b->num_blocks;
      b->block_size = 0;
      b->merge_last_count = 0;
      b->target_block_size = b->min_block_size;
    } else {
      split->lengths[b->num_blocks - 1] += (uint32_t)b->block_size;
      for (i = 0; i < nc; ++i) {
        memcpy(HISTO(b, b->last_ix[0] + i), combined + i * A, A * 4);
        last_entropy[i] = combined_entropy[i];
        if (split->num_types == 1) last_entropy[nc + i] = last_entropy[i];
        memset(HISTO(b, b->curr_ix + i), 0, A * 4);
      }
      b->block_size = 0;
      if (++b->merge_last_count > 1) b->target_block_size += b->min_block_size;
    }
    free(combined);
  }
  if (is_final) {
    b->histograms_size = split->num_types * nc;
    split->num_blocks = b->num_blocks;
  }
}

static void SplitterAdd(Splitter* b, size_t symbol, size_t context) {
  ++HISTO(b, b->curr_ix + context)[symbol];
  ++b->block_size;
  if (b->block_size == b->target_block_size) SplitterFinishBlock(b, 0);
}

/* c/enc/encode.c:258-269 */
static double EstimateEntropy(const uint32_t* population, size_t size) {
  size_t total = 0, i;
  double result = 0;
  for (i = 0; i < size; ++i) {
    uint32_t p = population[i];
    total += p;
    result += (double)p * FastLog2(p);
  }
  result = (double)total * FastLog2(total) - result;
  return result;
}

static const uint32_t kMapContinuation[64] = {1, 1, 2, 2};
static const uint32_t kMapSimpleUTF8[64] = {0, 0, 1, 1};
static const uint32_t kMapComplexUTF8[64] = {
    11, 11, 12, 12, 0, 0, 0, 0, 1, 1, 9, 9, 2, 2, 2, 2, 1, 1, 1, 1, 8, 3, 3, 3,
    1, 1, 1, 1, 2, 2, 2, 2, 8, 4, 4, 4, 8, 7, 4, 4, 8, 0, 0, 0, 3, 3, 3, 3,
    5, 5, 10, 5, 5, 5, 10, 5, 6, 6, 6, 6, 6, 6, 6, 6};

/* c/enc/encode.c:278-455 */
static void DecideOverLiteralContextModeling(const uint8_t* input, size_t start_pos,
    size_t length, size_t mask, int quality, size_t size_hint,
    size_t* num_literal_contexts, const uint32_t** literal_context_map) {
  if (quality < 5 || length < 64) return;
  if (size_hint >= (1u << 20)) {
    /* ShouldUseComplexStaticContextMap, :342-420 */
    const size_t end_pos = start_pos + length;
    uint32_t arena[32 * 14];
    uint32_t* combined_histo = arena;
    uint32_t* context_histo = arena + 32;
    uint32_t total = 0;
    double entropy[3];
    size_t i, sp = start_pos;
    const uint8_t* lut = &g_context_lut[2 << 9];
    memset(arena, 0, sizeof(arena));
    for (; sp + 64 <= end_pos; sp += 4096) {
      const size_t stride_end_pos = sp + 64;
      uint8_t prev2 = input[sp & mask];
      uint8_t prev1 = input[(sp + 1) & mask];
      size_t pos;
      for (pos = sp + 2; pos < stride_end_pos; ++pos) {
        const uint8_t literal = input[pos & mask];
        const uint8_t context = (uint8_t)kMapComplexUTF8[lut[prev1] | lut[256 + prev2]];
        ++total;
        ++combined_histo[literal >> 3];
        ++context_histo[(context << 5) + (literal >> 3)];
        prev2 = prev1;
        prev1 = literal;
      }
    }
    entropy[1] = EstimateEntropy(combined_histo, 32);
    entropy[2] = 0;
    for (i = 0; i < 13; ++i) entropy[2] += EstimateEntropy(context_histo + (i << 5), 32);
    entropy[0] = 1.0 / (double)total;
    entropy[1] *= entropy[0];
    entropy[2] *= entropy[0];
    if (!(entropy[2] > 3.0 || entropy[1] - entropy[2] < 0.2)) {
      *num_literal_contexts = 13;
      *literal_context_map = kMapComplexUTF8;
      return;
    }
  }
  {
    const size_t end_pos = start_pos + length;
    uint32_t bigram[9] = {0};
    static const int lut[4] = {0, 0, 1, 2};
    uint32_t monogram[3] = {0}, two_prefix[6] = {0};
    size_t total, i;
    double entropy[4];
    for (; start_pos + 64 <= end_pos; start_pos += 4096) {
      const size_t stride_end_pos = start_pos + 64;
      int prev = lut[input[start_pos & mask] >> 6] * 3;
      size_t pos;
      for (pos = start_pos + 1; pos < stride_end_pos; ++pos) {
        const uint8_t literal = input[pos & mask];
        ++bigram[prev + lut[literal >> 6]];
        prev = lut[literal >> 6] * 3;
      }
    }
    /* ChooseContextMap, :278-338 */
    for (i = 0; i < 9; ++i) {
      monogram[i % 3] += bigram[i];
      two_prefix[i % 6] += bigram[i];
    }
    entropy[1] = EstimateEntropy(monogram, 3);
    entropy[2] = EstimateEntropy(two_prefix, 3) + EstimateEntropy(two_prefix + 3, 3);
    entropy[3] = 0;
    for (i = 0; i < 3; ++i) entropy[3] += EstimateEntropy(bigram + 3 * i, 3);
    total = monogram[0] + monogram[1] + monogram[2];
    entropy[0] = 1.0 / (double)total;
    entropy[1] *= entropy[0];
    entropy[2] *= entropy[0];
    entropy[3] *= entropy[0];
    if (quality < 7) entropy[3] = entropy[1] * 10;
    if (entropy[1] - entropy[2] < 0.2 && entropy[1] - entropy[3] < 0.2) {
      *num_literal_contexts = 1;
    } else if (entropy[2] - entropy[3] < 0.02) {
      *num_literal_contexts = 2;
      *literal_context_map = kMapSimpleUTF8;
    } else {
      *num_literal_contexts = 3;
      *literal_context_map = kMapContinuation;
    }
  }
}

/* c/enc/encode.c:457-483 */
static int ShouldCompress(const uint8_t* data, size_t mask, uint64_t last_flush_pos,
    size_t bytes, size_t num_literals, size_t num_commands) {
  if (bytes <= 2) return 0;
  if (num_commands < (bytes >> 8) + 2) {
    if ((double)num_literals > 0.99 * (double)bytes) {
      uint32_t literal_histo[256] = {0};
      const double bit_cost_threshold = (double)bytes * 7.92 * (1.0 / 13.0);
      size_t t = (bytes + 13 - 1) / 13;
      uint32_t pos = (uint32_t)last_flush_pos;
      size_t i;
      for (i = 0; i < t; i++) {
        ++literal_histo[data[pos & mask]];
        pos += 13;
      }
      if (BitsEntropy(literal_histo, 256) > bit_cost_threshold) return 0;
    }
  }
  return 1;
}

/* ------------------------------------------------------------------------ */
/* Meta-block storage.                                                       */

static const struct { uint16_t offset; uint8_t nbits; } kBlockLen[26] = {
    {1, 2}, {5, 2}, {9, 2}, {13, 2}, {17, 3}, {25, 3}, {33, 3}, {41, 3},
    {49, 4}, {65, 4}, {81, 4}, {97, 4}, {113, 5}, {145, 5}, {177, 5},
    {209, 5}, {241, 6}, {305, 6}, {369, 7}, {497, 8}, {753, 9}, {1265, 10},
    {2289, 11}, {4337, 12}, {8433, 13}, {16625, 24}};

/* c/enc/brotli_bit_stream.c:34-46 */
static uint32_t BlockLengthPrefixCode(uint32_t len) {
  uint32_t code = (len >= 177) ? (len >= 753 ? 20 : 14) : (len >= 41 ? 7 : 0);
  while (code < 25 && len >= kBlockLen[code + 1].offset) ++code;
  return code;
}

typedef struct { size_t last_type, second_last_type; } TypeCalc;
static size_t NextBlockTypeCode(TypeCalc* c, uint8_t type) {
  size_t code = (type == c->last_type + 1) ? 1u : (type == c->second_last_type) ? 0u : type + 2u;
  c->second_last_type = c->last_type;
  c->last_type = type;
  return code;
}

typedef struct {
  TypeCalc calc;
  uint8_t type_depths[258];
  uint16_t type_bits[258];
  uint8_t length_depths[26];
  uint16_t length_bits[26];
} SplitCode;

typedef struct {
  size_t histogram_length, num_block_types;
  const uint8_t* block_types;
  const uint32_t* block_lengths;
  size_t num_blocks;
  SplitCode code;
  size_t block_ix, block_len, entropy_ix;
  uint8_t* depths;
  uint16_t* bits;
} BlockEnc;

static void StoreVarLenUint8(size_t n, size_t* ix, uint8_t* storage) {
  if (n == 0) {
    WriteBits(1, 0, ix, storage);
  } else {
    size_t nbits = Log2Floor(n);
    WriteBits(1, 1, ix, storage);
    WriteBits(3, nbits, ix, storage);
    WriteBits(nbits, n - ((size_t)1 << nbits), ix, storage);
  }
}

/* :737-756 */
static void StoreBlockSwitch(SplitCode* code, uint32_t block_len, uint8_t block_type,
                             int is_first, size_t* ix, uint8_t* storage) {
  size_t typecode = NextBlockTypeCode(&code->calc, block_type);
  uint32_t lencode = BlockLengthPrefixCode(block_len);
  if (!is_first) WriteBits(code->type_depths[typecode], code->type_bits[typecode], ix, storage);
  WriteBits(code->length_depths[lencode], code->length_bits[lencode], ix, storage);
  WriteBits(kBlockLen[lencode].nbits, block_len - kBlockLen[lencode].offset, ix, storage);
}

/* :760-791 */
static void BuildAndStoreBlockSplitCode(BlockEnc* e, HTree* tree, size_t* ix, uint8_t* storage) {
  uint32_t type_histo[258];
  uint32_t length_histo[26];
  size_t i;
  TypeCalc calc = {1, 0};
  memset(type_histo, 0, (e->num_block_types + 2) * 4);
  memset(length_histo, 0, sizeof(length_histo));
  for (i = 0; i < e->num_blocks; ++i) {
    size_t type_code = NextBlockTypeCode(&calc, e->block_types[i]);
    if (i != 0) ++type_histo[type_code];
    ++length_histo[BlockLengthPrefixCode(e->block_lengths[i])];
  }
  StoreVarLenUint8(e->num_block_types - 1, ix, storage);
  if (e->num_block_types > 1) {
    BuildAndStoreHuffmanTree(type_histo, e->num_block_types + 2, e->num_block_types + 2,
        tree, e->code.type_depths, e->code.type_bits, ix, storage);
    BuildAndStoreHuffmanTree(length_histo, 26, 26, tree, e->code.length_depths,
        e->code.length_bits, ix, storage);
    StoreBlockSwitch(&e->code, e->block_lengths[0], e->block_types[0], 1, ix, storage);
  }
}

static void BlockEncInit(BlockEnc* e, size_t histogram_length, const Split* sp) {
  memset(e, 0, sizeof(*e));
  e->histogram_length = histogram_length;
  e->num_block_types = sp->num_types;
  e->block_types = sp->types;
  e->block_lengths = sp->lengths;
  e->num_blocks = sp->num_blocks;
  e->code.calc.last_type = 1;
  e->code.calc.second_last_type = 0;
  e->block_len = sp->num_blocks == 0 ? 0 : sp->lengths[0];
}

/* :879-918 */
static void StoreSymbol(BlockEnc* e, size_t symbol, size_t* ix, uint8_t* storage) {
  if (e->block_len == 0) {
    size_t block_ix = ++e->block_ix;
    uint32_t block_len = e->block_lengths[block_ix];
    uint8_t block_type = e->block_types[block_ix];
    e->block_len = block_len;
    e->entropy_ix = block_type * e->histogram_length;
    StoreBlockSwitch(&e->code, block_len, block_type, 0, ix, storage);
  }
  --e->block_len;
  WriteBits(e->depths[e->entropy_ix + symbol], e->bits[e->entropy_ix + symbol], ix, storage);
}
static void StoreSymbolWithContext(BlockEnc* e, size_t symbol, size_t context,
    const uint32_t* context_map, size_t* ix, uint8_t* storage, size_t context_bits) {
  if (e->block_len == 0) {
    size_t block_ix = ++e->block_ix;
    uint32_t block_len = e->block_lengths[block_ix];
    uint8_t block_type = e->block_types[block_ix];
    e->block_len = block_len;
    e->entropy_ix = (size_t)block_type << context_bits;
    StoreBlockSwitch(&e->code, block_len, block_type, 0, ix, storage);
  }
  --e->block_len;
  {
    size_t histo_ix = context_map[e->entropy_ix + context];
    size_t k = histo_ix * e->histogram_length + symbol;
    WriteBits(e->depths[k], e->bits[k], ix, storage);
  }
}

/* :794-830 */
static void StoreTrivialContextMap(size_t num_types, size_t context_bits, HTree* tree,
                                   size_t* ix, uint8_t* storage) {
  StoreVarLenUint8(num_types - 1, ix, storage);
  if (num_types > 1) {
    size_t repeat_code = context_bits - 1u;
    size_t repeat_bits = (1u << repeat_code) - 1u;
    size_t alphabet_size = num_types + repeat_code;
    uint32_t histogram[272];
    uint8_t depths[272];
    uint16_t bits[272];
    size_t i;
    memset(histogram, 0, alphabet_size * 4);
    WriteBits(1, 1, ix, storage);
    WriteBits(4, repeat_code - 1, ix, storage);
    histogram[repeat_code] = (uint32_t)num_types;
    histogram[0] = 1;
    for (i = context_bits; i < alphabet_size; ++i) histogram[i] = 1;
    BuildAndStoreHuffmanTree(histogram, alphabet_size, alphabet_size, tree, depths, bits, ix, storage);
    for (i = 0; i < num_types; ++i) {
      size_t code = (i == 0 ? 0 : i + context_bits - 1);
      WriteBits(depths[code], bits[code], ix, storage);
      WriteBits(depths[repeat_code], bits[repeat_code], ix, storage);
      WriteBits(repeat_code, repeat_bits, ix, storage);
    }
    WriteBits(1, 1, ix, storage);
  }
}

/* :574-734 */
static void EncodeContextMap(const uint32_t* context_map, size_t context_map_size,
    size_t num_clusters, HTree* tree, size_t* ix, uint8_t* storage) {
  size_t i;
  uint32_t* rle;
  uint32_t max_run_length_prefix = 6;
  size_t num_rle = 0;
  uint32_t histogram[272];
  uint8_t depths[272];
  uint16_t bits[272];
  StoreVarLenUint8(num_clusters - 1, ix, storage);
  if (num_clusters == 1) return;
  rle = (uint32_t*)malloc(context_map_size * 4);
  { /* MoveToFrontTransform */
    uint8_t mtf[256];
    uint32_t max_value = context_map[0];
    size_t mtf_size;
    for (i = 1; i < context_map_size; ++i) if (context_map[i] > max_value) max_value = context_map[i];
    for (i = 0; i <= max_value; ++i) mtf[i] = (uint8_t)i;
    mtf_size = max_value + 1;
    for (i = 0; i < context_map_size; ++i) {
      size_t index = 0, k;
      uint8_t value;
      for (; index < mtf_size; ++index) if (mtf[index] == (uint8_t)context_map[i]) break;
      rle[i] = (uint32_t)index;
      value = mtf[index];
      for (k = index; k != 0; --k) mtf[k] = mtf[k - 1];
      mtf[0] = value;
    }
  }
  { /* RunLengthCodeZeros */
    uint32_t max_reps = 0, max_prefix;
    size_t in_size = context_map_size;
    for (i = 0; i < in_size;) {
      uint32_t reps = 0;
      for (; i < in_size && rle[i] != 0; ++i) ;
      for (; i < in_size && rle[i] == 0; ++i) ++reps;
      if (reps > max_reps) max_reps = reps;
    }
    max_prefix = max_reps > 0 ? Log2Floor(max_reps) : 0;
    if (max_prefix > max_run_length_prefix) max_prefix = max_run_length_prefix;
    max_run_length_prefix = max_prefix;
    for (i = 0; i < in_size;) {
      if (rle[i] != 0) {
        rle[num_rle++] = rle[i] + max_run_length_prefix;
        ++i;
      } else {
        uint32_t reps = 1;
        size_t k;
        for (k = i + 1; k < in_size && rle[k] == 0; ++k) ++reps;
        i += reps;
        while (reps != 0) {
          if (reps < (2u << max_prefix)) {
            uint32_t p = Log2Floor(reps);
            rle[num_rle++] = p + ((reps - (1u << p)) << 9);
            break;
          } else {
            rle[num_rle++] = max_prefix + (((1u << max_prefix) - 1u) << 9);
            reps -= (2u << max_prefix) - 1u;
          }
        }
      }
    }
  }
  memset(histogram, 0, sizeof(histogram));
  for (i = 0; i < num_rle; ++i) ++histogram[rle[i] & 511];
  {
    int use_rle = max_run_length_prefix > 0;
    WriteBits(1, (uint64_t)use_rle, ix, storage);
    if (use_rle) WriteBits(4, max_run_length_prefix - 1, ix, storage);
  }
  BuildAndStoreHuffmanTree(histogram, num_clusters + max_run_length_prefix,
      num_clusters + max_run_length_prefix, tree, depths, bits, ix, storage);
  for (i = 0; i < num_rle; ++i) {
    const uint32_t sym = rle[i] & 511;
    const uint32_t extra = rle[i] >> 9;
    WriteBits(depths[sym], bits[sym], ix, storage);
    if (sym > 0 && sym <= max_run_length_prefix) WriteBits(sym, extra, ix, storage);
  }
  WriteBits(1, 1, ix, storage);
  free(rle);
}

/* c/enc/brotli_bit_stream.c:1321-1352 */
static void StoreUncompressedMetaBlock(int is_final, const uint8_t* input, size_t position,
    size_t mask, size_t len, size_t* ix, uint8_t* storage) {
  size_t masked_pos = position & mask;
  {
    size_t lg = (len == 1) ? 1 : Log2Floor((uint32_t)(len - 1)) + 1;
    size_t mnibbles = (lg < 16 ? 16 : (lg + 3)) / 4;
    WriteBits(1, 0, ix, storage);
    WriteBits(2, mnibbles - 4, ix, storage);
    WriteBits(mnibbles * 4, len - 1, ix, storage);
    WriteBits(1, 1, ix, storage);
  }
  *ix = (*ix + 7u) & ~(size_t)7u;
  storage[*ix >> 3] = 0;
  if (masked_pos + len > mask + 1) {
    size_t len1 = mask + 1 - masked_pos;
    memcpy(&storage[*ix >> 3], &input[masked_pos], len1);
    *ix += len1 << 3;
    len -= len1;
    masked_pos = 0;
  }
  memcpy(&storage[*ix >> 3], &input[masked_pos], len);
  *ix += len << 3;
  storage[*ix >> 3] = 0;
  if (is_final) {
    WriteBits(1, 1, ix, storage);
    WriteBits(1, 1, ix, storage);
    *ix = (*ix + 7u) & ~(size_t)7u;
    storage[*ix >> 3] = 0;
  }
}

/* WriteMetaBlockInternal (encode.c:498-614) for 4 <= quality < 10:
   greedy builder (metablock.c:708-839), BrotliOptimizeHistograms
   (metablock.c:841-859), BrotliStoreMetaBlock (brotli_bit_stream.c:947-1114). */
static void BuildAndStoreHuffmanTreeFast(HTree* tree, const uint32_t* histogram,
    size_t histogram_total, size_t max_bits, uint8_t* depth, uint16_t* bits,
    size_t* ix, uint8_t* storage);
static void ConvertBitDepthsToSymbols(const uint8_t* depth, size_t len, uint16_t* bits);
static void FastStaticInit(void);

/* Qualities 2 and 3: one prefix code per category, no block splitting.
   BrotliStoreMetaBlockTrivial (brotli_bit_stream.c:1196-1240, quality 3) and
   BrotliStoreMetaBlockFast (:1242-1314, quality 2: count-only trees; up to 128 commands the
   command and distance codes are the static ones of entropy_encode_static.h — 448 command
   symbols of 9 bits + 256 of 11, 64 distance symbols of 6 bits, canonical — whose serialised
   forms are the constants of :524-541). */
static void WriteMetaBlockSimple(Enc* s, size_t bytes, int is_last, size_t* ix, uint8_t* storage) {
  const uint8_t* data = s->rb;
  const size_t mask = s->rb_mask;
  size_t pos = (size_t)s->last_flush_pos, i;
  uint32_t lit_histo[256], cmd_histo[704], dist_histo[140];
  uint8_t lit_depth[256], cmd_depth[704], dist_depth[140];
  uint16_t lit_bits[256], cmd_bits[704], dist_bits[140];
  size_t nlit = 0, ncmd = 0, ndist = 0;
  HTree* tree = (HTree*)malloc(sizeof(HTree) * (2 * 704 + 1));
  { /* StoreCompressedMetaBlockHeader :120-143 */
    size_t lg = (bytes == 1) ? 1 : Log2Floor((uint32_t)(bytes - 1)) + 1;
    size_t mnibbles = (lg < 16 ? 16 : (lg + 3)) / 4;
    WriteBits(1, (uint64_t)is_last, ix, storage);
    if (is_last) WriteBits(1, 0, ix, storage);
    WriteBits(2, mnibbles - 4, ix, storage);
    WriteBits(mnibbles * 4, bytes - 1, ix, storage);
    if (!is_last) WriteBits(1, 0, ix, storage);
  }
  WriteBits(13, 0, ix, storage);
  FastStaticInit();
  memset(lit_histo, 0, sizeof(lit_histo));
  memset(cmd_histo, 0, sizeof(cmd_histo));
  memset(dist_histo, 0, sizeof(dist_histo));
  memset(lit_depth, 0, sizeof(lit_depth));
  memset(cmd_depth, 0, sizeof(cmd_depth));
  memset(dist_depth, 0, sizeof(dist_depth));
  memset(lit_bits, 0, sizeof(lit_bits));
  memset(cmd_bits, 0, sizeof(cmd_bits));
  memset(dist_bits, 0, sizeof(dist_bits));
  for (i = 0; i < s->ncmds; ++i) { /* BuildHistograms :1152-1172 */
    const Cmd c = s->cmds[i];
    size_t j;
    ++cmd_histo[c.cmd_prefix]; ++ncmd;
    for (j = c.insert_len; j != 0; --j) { ++lit_histo[data[pos & mask]]; ++nlit; ++pos; }
    pos += CmdCopyLen(&c);
    if (CmdCopyLen(&c) && c.cmd_prefix >= 128) { ++dist_histo[c.dist_prefix & 0x3FF]; ++ndist; }
  }
  if (s->quality == 3) {
    BuildAndStoreHuffmanTree(lit_histo, 256, 256, tree, lit_depth, lit_bits, ix, storage);
    BuildAndStoreHuffmanTree(cmd_histo, 704, 704, tree, cmd_depth, cmd_bits, ix, storage);
    BuildAndStoreHuffmanTree(dist_histo, 140, 64, tree, dist_depth, dist_bits, ix, storage);
  } else if (s->ncmds <= 128) {
    BuildAndStoreHuffmanTreeFast(tree, lit_histo, nlit, 8, lit_depth, lit_bits, ix, storage);
    for (i = 0; i < 704; ++i) cmd_depth[i] = i < 448 ? 9 : 11;
    for (i = 0; i < 64; ++i) dist_depth[i] = 6;
    ConvertBitDepthsToSymbols(cmd_depth, 704, cmd_bits);
    ConvertBitDepthsToSymbols(dist_depth, 64, dist_bits);
    WriteBits(56, ((uint64_t)0x926244u << 32) | 0x16307003u, ix, storage);
    WriteBits(3, 0, ix, storage);
    WriteBits(28, 0x0369DC03u, ix, storage);
  } else {
    BuildAndStoreHuffmanTreeFast(tree, lit_histo, nlit, 8, lit_depth, lit_bits, ix, storage);
    BuildAndStoreHuffmanTreeFast(tree, cmd_histo, ncmd, 10, cmd_depth, cmd_bits, ix, storage);
    BuildAndStoreHuffmanTreeFast(tree, dist_histo, ndist, 6 /* Log2Floor(64 - 1) + 1 */, dist_depth, dist_bits, ix, storage);
  }
  free(tree);
  pos = (size_t)s->last_flush_pos;
  for (i = 0; i < s->ncmds; ++i) { /* StoreDataWithHuffmanCodes :1174-1207 */
    const Cmd c = s->cmds[i];
    size_t j;
    WriteBits(cmd_depth[c.cmd_prefix], cmd_bits[c.cmd_prefix], ix, storage);
    {
      uint32_t copylen_code = CmdCopyLenCode(&c);
      uint16_t inscode = InsertLengthCode(c.insert_len);
      uint16_t copycode = CopyLengthCode(copylen_code);
      uint32_t insnumextra = kInsExtra[inscode];
      uint64_t insextraval = c.insert_len - kInsBase[inscode];
      uint64_t copyextraval = copylen_code - kCopyBase[copycode];
      WriteBits(insnumextra + kCopyExtra[copycode], (copyextraval << insnumextra) | insextraval, ix, storage);
    }
    for (j = c.insert_len; j != 0; --j) {
      const uint8_t literal = data[pos & mask];
      WriteBits(lit_depth[literal], lit_bits[literal], ix, storage);
      ++pos;
    }
    pos += CmdCopyLen(&c);
    if (CmdCopyLen(&c) && c.cmd_prefix >= 128) {
      const size_t dist_code = c.dist_prefix & 0x3FF;
      WriteBits(dist_depth[dist_code], dist_bits[dist_code], ix, storage);
      WriteBits(c.dist_prefix >> 10, c.dist_extra, ix, storage);
    }
  }
  if (is_last) { *ix = (*ix + 7u) & ~(size_t)7u; storage[*ix >> 3] = 0; }
}

static void WriteMetaBlock(Enc* s, size_t bytes, int is_last, size_t* ix, uint8_t* storage) {
  const uint8_t* data = s->rb;
  const size_t mask = s->rb_mask;
  const uint64_t last_flush_pos = s->last_flush_pos;
  uint16_t last_bytes;
  uint8_t last_bytes_bits;
  const uint8_t* lut = &g_context_lut[2 << 9]; /* CONTEXT_UTF8, encode.c:486-496 */
  size_t num_contexts = 1;
  const uint32_t* static_map = NULL;
  Split lit_split, cmd_split, dist_split;
  Splitter lit, cmd, dist;
  uint32_t* literal_context_map = NULL;
  size_t literal_context_map_size = 0;
  size_t i;

  if (g_tap) { /* parity tap: the final command list of this meta-block */
    size_t k;
    for (k = 0; k < s->ncmds; ++k) {
      if (*g_tap_n < g_tap_cap) g_tap[*g_tap_n] = s->cmds[k];
      ++*g_tap_n;
    }
  }
  if (bytes == 0) {
    WriteBits(2, 3, ix, storage);
    *ix = (*ix + 7u) & ~(size_t)7u;
    return;
  }
  if (!ShouldCompress(data, mask, last_flush_pos, bytes, s->nlits, s->ncmds)) {
    memcpy(s->dist_cache, s->saved_dist_cache, 4 * sizeof(int));
    StoreUncompressedMetaBlock(is_last, data, (size_t)last_flush_pos, mask, bytes, ix, storage);
    return;
  }
  last_bytes = (uint16_t)((storage[1] << 8) | storage[0]);
  last_bytes_bits = (uint8_t)(*ix);
  if (s->quality < 4) { /* encode.c:543-555 */
    WriteMetaBlockSimple(s, bytes, is_last, ix, storage);
    if (bytes + 4 < (*ix >> 3)) {
      memcpy(s->dist_cache, s->saved_dist_cache, 4 * sizeof(int));
      storage[0] = (uint8_t)last_bytes;
      storage[1] = (uint8_t)(last_bytes >> 8);
      *ix = last_bytes_bits;
      StoreUncompressedMetaBlock(is_last, data, (size_t)last_flush_pos, mask, bytes, ix, storage);
    }
    return;
  }

  DecideOverLiteralContextModeling(data, (size_t)last_flush_pos, bytes, mask,
      s->quality, s->size_hint, &num_contexts, &static_map);

  { /* BrotliBuildMetaBlockGreedyInternal */
    size_t pos = (size_t)last_flush_pos;
    size_t num_literals = 0;
    uint8_t prev_byte = s->prev_byte, prev_byte2 = s->prev_byte2;
    for (i = 0; i < s->ncmds; ++i) num_literals += s->cmds[i].insert_len;
    SplitterInit(&lit, 256, num_contexts, 512, 400.0, num_literals, &lit_split);
    SplitterInit(&cmd, 704, 1, 1024, 500.0, s->ncmds, &cmd_split);
    SplitterInit(&dist, 64, 1, 512, 100.0, s->ncmds, &dist_split);
    for (i = 0; i < s->ncmds; ++i) {
      const Cmd c = s->cmds[i];
      size_t j;
      SplitterAdd(&cmd, c.cmd_prefix, 0);
      for (j = c.insert_len; j != 0; --j) {
        uint8_t literal = data[pos & mask];
        if (num_contexts == 1) {
          SplitterAdd(&lit, literal, 0);
        } else {
          size_t context = lut[prev_byte] | lut[256 + prev_byte2];
          SplitterAdd(&lit, literal, static_map[context]);
        }
        prev_byte2 = prev_byte;
        prev_byte = literal;
        ++pos;
      }
      pos += CmdCopyLen(&c);
      if (CmdCopyLen(&c)) {
        prev_byte2 = data[(pos - 2) & mask];
        prev_byte = data[(pos - 1) & mask];
        if (c.cmd_prefix >= 128) SplitterAdd(&dist, c.dist_prefix & 0x3FF, 0);
      }
    }
    SplitterFinishBlock(&lit, 1);
    SplitterFinishBlock(&cmd, 1);
    SplitterFinishBlock(&dist, 1);
    if (num_contexts > 1) { /* MapStaticContexts, metablock.c:677-697 */
      literal_context_map_size = lit_split.num_types << 6;
      literal_context_map = (uint32_t*)malloc(literal_context_map_size * 4);
      for (i = 0; i < lit_split.num_types; ++i) {
        uint32_t offset = (uint32_t)(i * num_contexts);
        size_t j;
        for (j = 0; j < 64; ++j) literal_context_map[(i << 6) + j] = offset + static_map[j];
      }
    }
  }
  if (getenv("ORACLE_DEBUG")) {
    const Split* sp[3] = {&lit_split, &cmd_split, &dist_split};
    int c;
    fprintf(stderr, "oracle nc=%zu ncmds=%zu\n", num_contexts, s->ncmds);
    for (c = 0; c < 3; ++c) {
      size_t b;
      fprintf(stderr, "  cat %d types=%zu blocks=%zu:", c, sp[c]->num_types, sp[c]->num_blocks);
      for (b = 0; b < sp[c]->num_blocks && b < 40; ++b) fprintf(stderr, " %u:%u", sp[c]->types[b], sp[c]->lengths[b]);
      fprintf(stderr, "\n");
    }
  }
  { /* BrotliOptimizeHistograms */
    uint8_t good_for_rle[704];
    for (i = 0; i < lit.histograms_size; ++i) OptimizeHuffmanCountsForRle(256, HISTO(&lit, i), good_for_rle);
    for (i = 0; i < cmd.histograms_size; ++i) OptimizeHuffmanCountsForRle(704, HISTO(&cmd, i), good_for_rle);
    for (i = 0; i < dist.histograms_size; ++i) OptimizeHuffmanCountsForRle(64, HISTO(&dist, i), good_for_rle);
  }
  { /* BrotliStoreMetaBlock */
    size_t pos = (size_t)last_flush_pos;
    HTree* tree = (HTree*)malloc(sizeof(HTree) * (2 * 704 + 1));
    BlockEnc le, ce, de;
    uint8_t prev_byte = s->prev_byte, prev_byte2 = s->prev_byte2;
    { /* StoreCompressedMetaBlockHeader :120-143 */
      size_t lg = (bytes == 1) ? 1 : Log2Floor((uint32_t)(bytes - 1)) + 1;
      size_t mnibbles = (lg < 16 ? 16 : (lg + 3)) / 4;
      WriteBits(1, (uint64_t)is_last, ix, storage);
      if (is_last) WriteBits(1, 0, ix, storage);
      WriteBits(2, mnibbles - 4, ix, storage);
      WriteBits(mnibbles * 4, bytes - 1, ix, storage);
      if (!is_last) WriteBits(1, 0, ix, storage);
    }
    BlockEncInit(&le, 256, &lit_split);
    BlockEncInit(&ce, 704, &cmd_split);
    BlockEncInit(&de, 64, &dist_split);
    BuildAndStoreBlockSplitCode(&le, tree, ix, storage);
    BuildAndStoreBlockSplitCode(&ce, tree, ix, storage);
    BuildAndStoreBlockSplitCode(&de, tree, ix, storage);
    WriteBits(2, 0, ix, storage); /* NPOSTFIX */
    WriteBits(4, 0, ix, storage); /* NDIRECT >> NPOSTFIX */
    for (i = 0; i < lit_split.num_types; ++i) WriteBits(2, 2 /* CONTEXT_UTF8 */, ix, storage);
    if (literal_context_map_size == 0) {
      StoreTrivialContextMap(lit.histograms_size, 6, tree, ix, storage);
    } else {
      EncodeContextMap(literal_context_map, literal_context_map_size, lit.histograms_size, tree, ix, storage);
    }
    StoreTrivialContextMap(dist.histograms_size, 2, tree, ix, storage);
    le.depths = (uint8_t*)malloc(lit.histograms_size * 256);
    le.bits = (uint16_t*)malloc(lit.histograms_size * 256 * 2);
    for (i = 0; i < lit.histograms_size; ++i)
      BuildAndStoreHuffmanTree(HISTO(&lit, i), 256, 256, tree, &le.depths[i * 256], &le.bits[i * 256], ix, storage);
    ce.depths = (uint8_t*)malloc(cmd.histograms_size * 704);
    ce.bits = (uint16_t*)malloc(cmd.histograms_size * 704 * 2);
    for (i = 0; i < cmd.histograms_size; ++i)
      BuildAndStoreHuffmanTree(HISTO(&cmd, i), 704, 704, tree, &ce.depths[i * 704], &ce.bits[i * 704], ix, storage);
    de.depths = (uint8_t*)malloc(dist.histograms_size * 64);
    de.bits = (uint16_t*)malloc(dist.histograms_size * 64 * 2);
    for (i = 0; i < dist.histograms_size; ++i)
      BuildAndStoreHuffmanTree(HISTO(&dist, i), 64, 64, tree, &de.depths[i * 64], &de.bits[i * 64], ix, storage);
    free(tree);
    for (i = 0; i < s->ncmds; ++i) {
      const Cmd c = s->cmds[i];
      size_t j;
      StoreSymbol(&ce, c.cmd_prefix, ix, storage);
      { /* StoreCommandExtra :82-93 */
        uint32_t copylen_code = CmdCopyLenCode(&c);
        uint16_t inscode = InsertLengthCode(c.insert_len);
        uint16_t copycode = CopyLengthCode(copylen_code);
        uint32_t insnumextra = kInsExtra[inscode];
        uint64_t insextraval = c.insert_len - kInsBase[inscode];
        uint64_t copyextraval = copylen_code - kCopyBase[copycode];
        WriteBits(insnumextra + kCopyExtra[copycode], (copyextraval << insnumextra) | insextraval, ix, storage);
      }
      if (literal_context_map_size == 0) {
        for (j = c.insert_len; j != 0; --j) { StoreSymbol(&le, data[pos & mask], ix, storage); ++pos; }
      } else {
        for (j = c.insert_len; j != 0; --j) {
          size_t context = lut[prev_byte] | lut[256 + prev_byte2];
          uint8_t literal = data[pos & mask];
          StoreSymbolWithContext(&le, literal, context, literal_context_map, ix, storage, 6);
          prev_byte2 = prev_byte;
          prev_byte = literal;
          ++pos;
        }
      }
      pos += CmdCopyLen(&c);
      if (CmdCopyLen(&c)) {
        prev_byte2 = data[(pos - 2) & mask];
        prev_byte = data[(pos - 1) & mask];
        if (c.cmd_prefix >= 128) {
          StoreSymbol(&de, c.dist_prefix & 0x3FF, ix, storage);
          WriteBits(c.dist_prefix >> 10, c.dist_extra, ix, storage);
        }
      }
    }
    free(le.depths); free(le.bits); free(ce.depths); free(ce.bits); free(de.depths); free(de.bits);
    if (is_last) { *ix = (*ix + 7u) & ~(size_t)7u; storage[*ix >> 3] = 0; }
  }
  free(lit_split.types); free(lit_split.lengths); free(lit.histograms);
  free(cmd_split.types); free(cmd_split.lengths); free(cmd.histograms);
  free(dist_split.types); free(dist_split.lengths); free(dist.histograms);
  free(literal_context_map);
  if (bytes + 4 < (*ix >> 3)) {
    memcpy(s->dist_cache, s->saved_dist_cache, 4 * sizeof(int));
    storage[0] = (uint8_t)last_bytes;
    storage[1] = (uint8_t)(last_bytes >> 8);
    *ix = last_bytes_bits;
    StoreUncompressedMetaBlock(is_last, data, (size_t)last_flush_pos, mask, bytes, ix, storage);
  }
}

/* ------------------------------------------------------------------------ */
/* Stream driver.                                                            */

/* c/enc/encode.c:905-971 (no compound dictionary) */
static void ExtendLastCommand(Enc* s, uint32_t* bytes, uint32_t* wrapped_pos) {
  Cmd* last = &s->cmds[s->ncmds - 1];
  const uint8_t* data = s->rb;
  const uint32_t mask = s->rb_mask;
  uint64_t max_backward_distance = (((uint64_t)1) << s->lgwin) - 16;
  uint64_t last_copy_len = last->copy_len & 0x1FFFFFF;
  uint64_t last_processed_pos = s->last_processed_pos - last_copy_len;
  uint64_t max_distance = last_processed_pos < max_backward_distance ? last_processed_pos : max_backward_distance;
  uint64_t cmd_dist = (uint64_t)s->dist_cache[0];
  uint32_t distance_code; /* CommandRestoreDistanceCode, command.h:145-164 */
  {
    uint32_t dcode = last->dist_prefix & 0x3FFu;
    if (dcode < 16) {
      distance_code = dcode;
    } else {
      uint32_t nbits = last->dist_prefix >> 10;
      uint32_t hcode = dcode - 16;
      uint32_t offset = ((2U + (hcode & 1U)) << nbits) - 4U;
      distance_code = offset + last->dist_extra + 16;
    }
  }
  if (distance_code < 16 || distance_code - 15 == cmd_dist) {
    if (cmd_dist <= max_distance) {
      while (*bytes != 0 && data[*wrapped_pos & mask] == data[(*wrapped_pos - cmd_dist) & mask]) {
        last->copy_len++;
        (*bytes)--;
        (*wrapped_pos)++;
      }
    } else if ((cmd_dist - max_distance - 1) < g_cd.total_size && last_copy_len < cmd_dist - max_distance) {
      /* encode.c:930-961: the copy continues inside the attached dictionary, chunk after chunk */
      size_t address = g_cd.total_size - (size_t)(cmd_dist - max_distance) + (size_t)last_copy_len;
      size_t br_index = 0, br_offset, chunk_length;
      const uint8_t* chunk;
      while (address >= g_cd.chunk_offsets[br_index + 1]) br_index++;
      br_offset = address - g_cd.chunk_offsets[br_index];
      chunk = g_cd.chunks[br_index].source;
      chunk_length = g_cd.chunk_offsets[br_index + 1] - g_cd.chunk_offsets[br_index];
      while (*bytes != 0 && data[*wrapped_pos & mask] == chunk[br_offset]) {
        last->copy_len++;
        (*bytes)--;
        (*wrapped_pos)++;
        if (++br_offset == chunk_length) {
          br_index++;
          br_offset = 0;
          if (br_index != g_cd.num_chunks) {
            chunk = g_cd.chunks[br_index].source;
            chunk_length = g_cd.chunk_offsets[br_index + 1] - g_cd.chunk_offsets[br_index];
          } else {
            break;
          }
        }
      }
    }
    last->cmd_prefix = CombineLengthCodes(InsertLengthCode(last->insert_len),
        CopyLengthCode((size_t)((int)(last->copy_len & 0x1FFFFFF) + (int)(last->copy_len >> 25))),
        (last->dist_prefix & 0x3FF) == 0);
  }
}

/* c/enc/encode.c:898-903 */
static int UpdateLastProcessedPos(Enc* s) {
  uint32_t a = WrapPosition(s->last_processed_pos);
  uint32_t b = WrapPosition(s->input_pos);
  s->last_processed_pos = s->input_pos;
  return b < a;
}

/* c/enc/encode.c:985-1221; returns 0 on failure. Output appended to s->out. */
static int EncodeData(Enc* s, int is_last, int force_flush) {
  const uint64_t delta = s->input_pos - s->last_processed_pos;
  uint32_t bytes = (uint32_t)delta;
  uint32_t wrapped_last_processed_pos = WrapPosition(s->last_processed_pos);
  if (delta == 0) {
    if (!s->rb) {
      if (is_last) {
        uint8_t b[2];
        s->last_bytes |= (uint16_t)(3u << s->last_bytes_bits);
        s->last_bytes_bits = (uint8_t)(s->last_bytes_bits + 2u);
        b[0] = (uint8_t)s->last_bytes;
        b[1] = (uint8_t)(s->last_bytes >> 8);
        Emit(s, b, (s->last_bytes_bits + 7u) >> 3u);
        return 1;
      }
      return 1;
    } else if (!is_last && !force_flush) {
      return 1;
    }
  }
  if (s->is_last_emitted) return 0;
  if (is_last) s->is_last_emitted = 1;
  if (delta > ((size_t)1 << s->lgblock)) return 0;
  {
    size_t newsize = s->ncmds + bytes / 2 + 1;
    if (newsize > s->cmd_cap) {
      newsize += (bytes / 4) + 16;
      s->cmd_cap = newsize;
      s->cmds = (Cmd*)realloc(s->cmds, sizeof(Cmd) * newsize);
    }
  }
  /* InitOrStitchToPreviousBlock, hash.h:505-522 */
  HasherSetup(s);
  if (bytes >= HashTypeLength(s) - 1 && wrapped_last_processed_pos >= 3) {
    HStore(s, wrapped_last_processed_pos - 3);
    HStore(s, wrapped_last_processed_pos - 2);
    HStore(s, wrapped_last_processed_pos - 1);
  }
  if (s->ncmds && s->last_insert_len == 0) ExtendLastCommand(s, &bytes, &wrapped_last_processed_pos);
  CreateBackwardReferences(s, bytes, wrapped_last_processed_pos);
  {
    const int rb_bits = 1 + (s->lgwin > s->lgblock ? s->lgwin : s->lgblock);
    const size_t max_length = (size_t)1 << (rb_bits < 24 ? rb_bits : 24);
    const size_t max_literals = max_length / 8, max_commands = max_length / 8;
    const size_t processed_bytes = (size_t)(s->input_pos - s->last_flush_pos);
    const int next_fits = processed_bytes + ((size_t)1 << s->lgblock) <= max_length;
    /* no block splitting: flush once 0x2FFF symbols have gathered (encode.c:1150-1153) */
    const int should_flush = s->quality < 4 && s->nlits + s->ncmds >= 0x2FFF;
    if (!is_last && !force_flush && !should_flush && next_fits && s->nlits < max_literals && s->ncmds < max_commands) {
      if (UpdateLastProcessedPos(s)) s->hasher_prepared = 0;
      return 1;
    }
  }
  if (s->last_insert_len > 0) {
    InitInsertCommand(&s->cmds[s->ncmds++], s->last_insert_len);
    s->nlits += s->last_insert_len;
    s->last_insert_len = 0;
  }
  if (!is_last && s->input_pos == s->last_flush_pos) return 1;
  {
    const uint32_t metablock_size = (uint32_t)(s->input_pos - s->last_flush_pos);
    uint8_t* storage = (uint8_t*)malloc(2 * (size_t)metablock_size + 503 + 16);
    size_t ix = s->last_bytes_bits;
    storage[0] = (uint8_t)s->last_bytes;
    storage[1] = (uint8_t)(s->last_bytes >> 8);
    WriteMetaBlock(s, metablock_size, is_last, &ix, storage);
    s->last_bytes = (uint16_t)(storage[ix >> 3]);
    s->last_bytes_bits = ix & 7u;
    s->last_flush_pos = s->input_pos;
    if (UpdateLastProcessedPos(s)) s->hasher_prepared = 0;
    if (s->last_flush_pos > 0) s->prev_byte = s->rb[((uint32_t)s->last_flush_pos - 1) & s->rb_mask];
    if (s->last_flush_pos > 1) s->prev_byte2 = s->rb[(uint32_t)(s->last_flush_pos - 2) & s->rb_mask];
    s->ncmds = 0;
    s->nlits = 0;
    memcpy(s->saved_dist_cache, s->dist_cache, sizeof(s->saved_dist_cache));
    Emit(s, storage, ix >> 3);
    free(storage);
    return 1;
  }
}

size_t oracle_encode_shard(const uint8_t* in, size_t len, int quality, int lgwin,
    uint32_t size_hint, uint32_t stream_offset, int is_last_shard,
    uint8_t* out, size_t out_cap) {
  Enc e;
  Enc* s = &e;
  size_t avail = len;
  const uint8_t* next = in;
  int state = 0; /* 0 processing, 1 flush requested, 2 finished */
  if (!g_ready) return 0;
  memset(s, 0, sizeof(*s));
  s->quality = quality;
  s->lgwin = lgwin;
  s->size_hint = size_hint;
  s->stream_offset = stream_offset;
  s->out = out;
  s->out_cap = out_cap;
  s->dist_cache[0] = 4; s->dist_cache[1] = 11; s->dist_cache[2] = 15; s->dist_cache[3] = 16;
  memcpy(s->saved_dist_cache, s->dist_cache, sizeof(s->saved_dist_cache));
  /* EnsureInitialized, encode.c:642-700 */
  s->flint = -2;
  s->lgblock = quality < 4 ? 14 : 16;   /* ComputeLgBlock, quality.h:75-93 */
  if (quality >= 9 && lgwin > 16) s->lgblock = lgwin < 18 ? lgwin : 18;
  /* BROTLI_PARAM_LGBLOCK (oracle_set_lgblock): from quality 4 on, clamped to [16, 24] (quality.h:85-89) */
  if (quality >= 4 && g_lgblock_param != 0) s->lgblock = g_lgblock_param < 16 ? 16 : g_lgblock_param > 24 ? 24 : g_lgblock_param;
  if (stream_offset != 0) {
    s->flint = 2;
    s->dist_cache[0] = s->dist_cache[1] = s->dist_cache[2] = s->dist_cache[3] = -16;
    memcpy(s->saved_dist_cache, s->dist_cache, sizeof(s->saved_dist_cache));
  }
  {
    int rb_bits = 1 + (lgwin > s->lgblock ? lgwin : s->lgblock);
    s->rb_size = 1u << rb_bits;
    s->rb_mask = s->rb_size - 1;
    s->rb_tail = 1u << s->lgblock;
    s->rb_total = s->rb_size + s->rb_tail;
  }
  if (stream_offset == 0) {
    /* EncodeWindowBits, encode.c:191-211 */
    if (lgwin == 16) { s->last_bytes = 0; s->last_bytes_bits = 1; }
    else if (lgwin == 17) { s->last_bytes = 1; s->last_bytes_bits = 7; }
    else if (lgwin > 17) { s->last_bytes = (uint16_t)(((lgwin - 17) << 1) | 1); s->last_bytes_bits = 4; }
    else { s->last_bytes = (uint16_t)(((lgwin - 8) << 4) | 1); s->last_bytes_bits = 7; }
  } else {
    size_t lim = ((size_t)1 << lgwin) - 16;
    if (s->stream_offset > lim) s->stream_offset = lim;
  }
  /* BrotliEncoderCompressStream main loop, encode.c:1665-1719, with op =
     FINISH (last shard) or FLUSH and an unbounded output buffer. */
  for (;;) {
    size_t block = (size_t)1 << s->lgblock;
    uint64_t d = s->input_pos - s->last_processed_pos;
    size_t remaining = d >= block ? 0 : block - (size_t)d;
    if (s->flint >= 0 && remaining > (size_t)s->flint) remaining = (size_t)s->flint;
    if (remaining != 0 && avail != 0) {
      size_t n = remaining < avail ? remaining : avail;
      CopyInputToRingBuffer(s, n, next);
      next += n; avail -= n;
      if (s->flint > 0) s->flint = (int)(s->flint - (int)n);
      continue;
    }
    if (state == 1) {
      /* InjectFlushOrPushOutput / InjectBytePaddingBlock, :1356-1415 */
      if (s->last_bytes_bits != 0) {
        uint32_t seal = s->last_bytes;
        size_t seal_bits = s->last_bytes_bits;
        uint8_t b[3];
        s->last_bytes = 0;
        s->last_bytes_bits = 0;
        seal |= 0x6u << seal_bits;
        seal_bits += 6;
        b[0] = (uint8_t)seal; b[1] = (uint8_t)(seal >> 8); b[2] = (uint8_t)(seal >> 16);
        Emit(s, b, (seal_bits + 7) >> 3);
      }
      state = 0; /* CheckFlushComplete */
      if (s->flint == -1) { s->flint = -2; continue; }
      break; /* flush requested by the caller is complete */
    }
    if (state == 0) {
      if (remaining == 0 || 1 /* op != PROCESS */) {
        int is_last = (avail == 0) && is_last_shard;
        int force_flush = (avail == 0) && !is_last_shard;
        if (!is_last && s->flint == 0) { s->flint = -1; force_flush = 1; }
        if (remaining != 0 && avail != 0) { /* unreachable */ }
        if (s->size_hint == 0) { /* UpdateSizeHint, :1619-1632 */
          uint64_t dd = s->input_pos - s->last_processed_pos;
          uint64_t tot = dd + avail;
          s->size_hint = tot >= (1u << 30) ? (1u << 30) : (uint32_t)tot;
        }
        if (!s->hasher_setup && !ChooseHasher(s)) { s->overflow = 1; break; }
        if (!EncodeData(s, is_last, force_flush)) { s->overflow = 1; break; }
        if (force_flush) state = 1;
        if (is_last) { state = 2; break; }
        continue;
      }
    }
    break;
  }
  free(s->rb_data); free(s->num); free(s->tags); free(s->buckets); free(s->cmds);
  free(s->fc_addr); free(s->fc_head); free(s->fc_tiny); free(s->fc_free); free(s->fc_delta); free(s->fc_next);
  if (s->overflow) return 0;
  return s->out_len;
}

size_t oracle_encode_plan(const uint8_t* in, size_t len, int quality, int lgwin,
    size_t shard_size, uint8_t* out, size_t out_cap, uint64_t* shard_sizes_out) {
  size_t nshards, k, total = 0;
  uint32_t size_hint = len >= (1u << 30) ? (1u << 30) : (uint32_t)len;
  if (len == 0) { if (out_cap < 1) return 0; out[0] = 6; return 1; }
  if (shard_size == 0 || shard_size >= len) shard_size = len;
  nshards = (len + shard_size - 1) / shard_size;
  for (k = 0; k < nshards; ++k) {
    size_t off = k * shard_size;
    size_t n = len - off < shard_size ? len - off : shard_size;
    uint32_t so = off >= (1u << 30) ? (1u << 30) : (uint32_t)off;
    size_t w = oracle_encode_shard(in + off, n, quality, lgwin, size_hint, so,
                                   k + 1 == nshards, out + total, out_cap - total);
    if (w == 0) return 0;
    if (shard_sizes_out) shard_sizes_out[k] = w;
    total += w;
  }
  return total;
}

/* ======================================================================== */
/* Quality 1: the two-pass fragment compressor.        */
/* Restates c/enc/compress_fragment_two_pass.c, the fast tree writer         */
/* c/enc/brotli_bit_stream.c:404-573 and the driver                           */
/* BrotliEncoderCompressStreamFast c/enc/encode.c:1425-1547.                 */

#define F_BLOCK ((size_t)1 << 17)          /* kCompressFragmentTwoPassBlockSize */
#define F_MAX_DISTANCE ((long)(((size_t)1 << 18) - 16)) /* compress_fragment_two_pass.c:29 */

/* Static code-length code of the fast tree writer, generated instead of
   tabulated (entropy_encode_static.h:20-22, 82-90): depths 4 for symbols
   0..12, 16, 17; 5 for 13, 14; 15 unused. */
static uint8_t f_cl_depth[18];
static uint16_t f_cl_bits[18];
static uint64_t f_zero_bits[704], f_nonzero_bits[704];
static uint8_t f_zero_depth[704], f_nonzero_depth[704];
static int f_static_ready;

static void FastStaticInit(void) {
  size_t reps, i;
  if (f_static_ready) return;
  for (i = 0; i < 18; ++i) f_cl_depth[i] = 4;
  f_cl_depth[13] = f_cl_depth[14] = 5;
  f_cl_depth[15] = 0;
  ConvertBitDepthsToSymbols(f_cl_depth, 18, f_cl_bits);
  /* A run of `reps` zeros / `reps + 3` repeats of the previous non-zero length as
     the bit string the tree serialiser would emit for it (entropy_encode.c:160-239
     without the "7" / value special cases the fast writer handles itself). */
  for (reps = 0; reps < 704; ++reps) {
    uint8_t tree[16], extra[16];
    size_t n = 0, k;
    uint64_t bits = 0; unsigned nb = 0;
    WriteTreeRepsZeros(reps, &n, tree, extra);
    for (k = 0; k < n; ++k) {
      bits |= (uint64_t)f_cl_bits[tree[k]] << nb; nb += f_cl_depth[tree[k]];
      if (tree[k] == 17) { bits |= (uint64_t)extra[k] << nb; nb += 3; }
    }
    f_zero_bits[reps] = bits; f_zero_depth[reps] = (uint8_t)nb;
    {
      size_t r = reps, start = 0;
      n = 0; bits = 0; nb = 0;
      for (;;) {
        tree[n] = 16; extra[n] = (uint8_t)(r & 3); ++n;
        r >>= 2;
        if (r == 0) break;
        --r;
      }
      Reverse(tree, start, n);
      Reverse(extra, start, n);
      for (k = 0; k < n; ++k) {
        bits |= (uint64_t)f_cl_bits[16] << nb; nb += f_cl_depth[16];
        bits |= (uint64_t)extra[k] << nb; nb += 2;
      }
      f_nonzero_bits[reps] = bits; f_nonzero_depth[reps] = (uint8_t)nb;
    }
  }
  f_static_ready = 1;
}

/* test hook: lets tests/ compare the generated tables with the reference's */
void oracle_fast_static_tables(const uint64_t** zb, const uint8_t** zd,
                               const uint64_t** nzb, const uint8_t** nzd) {
  FastStaticInit();
  *zb = f_zero_bits; *zd = f_zero_depth; *nzb = f_nonzero_bits; *nzd = f_nonzero_depth;
}

/* entropy_encode.h:82-115 with the count-only comparator of
   brotli_bit_stream.c:399-402 */
static void SortTreeByCount(HTree* items, size_t n) {
  static const size_t gaps[] = {132, 57, 23, 10, 4, 1};
  if (n < 13) {
    size_t i;
    for (i = 1; i < n; ++i) {
      HTree tmp = items[i];
      size_t k = i, j = i - 1;
      while (tmp.total_count < items[j].total_count) {
        items[k] = items[j];
        k = j;
        if (!j--) break;
      }
      items[k] = tmp;
    }
  } else {
    int g = n < 57 ? 2 : 0;
    for (; g < 6; ++g) {
      size_t gap = gaps[g], i;
      for (i = gap; i < n; ++i) {
        size_t j = i;
        HTree tmp = items[i];
        for (; j >= gap && tmp.total_count < items[j - gap].total_count; j -= gap) items[j] = items[j - gap];
        items[j] = tmp;
      }
    }
  }
}

/* brotli_bit_stream.c:404-573 */
static void BuildAndStoreHuffmanTreeFast(HTree* tree, const uint32_t* histogram,
    size_t histogram_total, size_t max_bits, uint8_t* depth, uint16_t* bits,
    size_t* ix, uint8_t* storage) {
  size_t count = 0, symbols[4] = {0}, length = 0, total = histogram_total;
  while (total != 0) {
    if (histogram[length]) {
      if (count < 4) symbols[count] = length;
      ++count;
      total -= histogram[length];
    }
    ++length;
  }
  if (count <= 1) {
    WriteBits(4, 1, ix, storage);
    WriteBits(max_bits, symbols[0], ix, storage);
    depth[symbols[0]] = 0;
    bits[symbols[0]] = 0;
    return;
  }
  memset(depth, 0, length);
  {
    uint32_t count_limit;
    HTree sentinel;
    sentinel.total_count = 0xFFFFFFFFu; sentinel.left = -1; sentinel.right_or_value = -1;
    for (count_limit = 1;; count_limit *= 2) {
      size_t n = 0, l;
      int i = 0, j, k;
      for (l = length; l != 0;) {
        --l;
        if (histogram[l]) {
          tree[n].total_count = histogram[l] >= count_limit ? histogram[l] : count_limit;
          tree[n].left = -1; tree[n].right_or_value = (int16_t)l;
          ++n;
        }
      }
      SortTreeByCount(tree, n);
      tree[n] = sentinel;
      tree[n + 1] = sentinel;
      j = (int)n + 1;
      for (k = (int)n - 1; k > 0; --k) {
        int left, right;
        size_t end = 2 * n - (size_t)k;
        if (tree[i].total_count <= tree[j].total_count) { left = i; ++i; } else { left = j; ++j; }
        if (tree[i].total_count <= tree[j].total_count) { right = i; ++i; } else { right = j; ++j; }
        tree[end].total_count = tree[left].total_count + tree[right].total_count;
        tree[end].left = (int16_t)left;
        tree[end].right_or_value = (int16_t)right;
        tree[end + 1] = sentinel;
      }
      if (SetDepth((int)(2 * n - 1), tree, depth, 14)) break;
    }
  }
  ConvertBitDepthsToSymbols(depth, length, bits);
  if (count <= 4) {
    size_t i, j;
    WriteBits(2, 1, ix, storage);
    WriteBits(2, count - 1, ix, storage);
    for (i = 0; i < count; i++) {
      for (j = i + 1; j < count; j++) {
        if (depth[symbols[j]] < depth[symbols[i]]) { size_t t = symbols[j]; symbols[j] = symbols[i]; symbols[i] = t; }
      }
    }
    for (i = 0; i < count; ++i) WriteBits(max_bits, symbols[i], ix, storage);
    if (count == 4) WriteBits(1, depth[symbols[0]] == 1 ? 1 : 0, ix, storage);
  } else {
    uint8_t previous_value = 8;
    size_t i;
    WriteBits(40, ((uint64_t)0xFFu << 32) | 0x55555554u, ix, storage);
    for (i = 0; i < length;) {
      const uint8_t value = depth[i];
      size_t reps = 1, k;
      for (k = i + 1; k < length && depth[k] == value; ++k) ++reps;
      i += reps;
      if (value == 0) {
        WriteBits(f_zero_depth[reps], f_zero_bits[reps], ix, storage);
      } else {
        if (previous_value != value) { WriteBits(f_cl_depth[value], f_cl_bits[value], ix, storage); --reps; }
        if (reps < 3) {
          while (reps != 0) { reps--; WriteBits(f_cl_depth[value], f_cl_bits[value], ix, storage); }
        } else {
          reps -= 3;
          WriteBits(f_nonzero_depth[reps], f_nonzero_bits[reps], ix, storage);
        }
        previous_value = value;
      }
    }
  }
}

/* compress_fragment_two_pass.c:31-52 */
static uint32_t FHash(const uint8_t* p, size_t shift, size_t length) {
  const uint64_t h = (Load64(p) << ((8 - length) * 8)) * 0x1E35A7BDu;
  return (uint32_t)(h >> shift);
}
static uint32_t FHashAt(uint64_t v, size_t offset, size_t shift, size_t length) {
  const uint64_t h = ((v >> (8 * offset)) << ((8 - length) * 8)) * 0x1E35A7BDu;
  return (uint32_t)(h >> shift);
}
static int FIsMatch(const uint8_t* p1, const uint8_t* p2, size_t length) {
  if (Load32(p1) != Load32(p2)) return 0;
  if (length == 4) return 1;
  return p1[4] == p2[4] && p1[5] == p2[5];
}

/* Two-pass command words: low byte = code in the 128-symbol working alphabet
   (0..23 insert, 24..39 copy after a repeated distance, 40..63 copy, 64..127
   distance), upper 24 bits = extra-bit value.  :106-214 */
static uint32_t FInsertLen(uint32_t insertlen) {
  if (insertlen < 6) return insertlen;
  if (insertlen < 130) {
    const uint32_t tail = insertlen - 2, nbits = Log2Floor(tail) - 1u, prefix = tail >> nbits;
    return ((nbits << 1) + prefix + 2) | ((tail - (prefix << nbits)) << 8);
  }
  if (insertlen < 2114) {
    const uint32_t tail = insertlen - 66, nbits = Log2Floor(tail);
    return (nbits + 10) | ((tail - (1u << nbits)) << 8);
  }
  if (insertlen < 6210) return 21 | ((insertlen - 2114) << 8);
  if (insertlen < 22594) return 22 | ((insertlen - 6210) << 8);
  return 23 | ((insertlen - 22594) << 8);
}
static uint32_t FCopyLen(size_t copylen) {
  if (copylen < 10) return (uint32_t)(copylen + 38);
  if (copylen < 134) {
    const size_t tail = copylen - 6, nbits = Log2Floor(tail) - 1, prefix = tail >> nbits;
    return (uint32_t)(((nbits << 1) + prefix + 44) | ((tail - (prefix << nbits)) << 8));
  }
  if (copylen < 2118) {
    const size_t tail = copylen - 70, nbits = Log2Floor(tail);
    return (uint32_t)((nbits + 52) | ((tail - ((size_t)1 << nbits)) << 8));
  }
  return (uint32_t)(63 | ((copylen - 2118) << 8));
}
/* returns the number of words written (1 or 2) */
static int FCopyLenLastDistance(size_t copylen, uint32_t* w) {
  if (copylen < 12) { w[0] = (uint32_t)(copylen + 20); return 1; }
  if (copylen < 72) {
    const size_t tail = copylen - 8, nbits = Log2Floor(tail) - 1, prefix = tail >> nbits;
    w[0] = (uint32_t)(((nbits << 1) + prefix + 28) | ((tail - (prefix << nbits)) << 8));
    return 1;
  }
  if (copylen < 136) {
    const size_t tail = copylen - 8;
    w[0] = (uint32_t)(((tail >> 5) + 54) | ((tail & 31) << 8));
  } else if (copylen < 2120) {
    const size_t tail = copylen - 72, nbits = Log2Floor(tail);
    w[0] = (uint32_t)((nbits + 52) | ((tail - ((size_t)1 << nbits)) << 8));
  } else {
    w[0] = (uint32_t)(63 | ((copylen - 2120) << 8));
  }
  w[1] = 64;
  return 2;
}
static uint32_t FDistance(uint32_t distance) {
  const uint32_t d = distance + 3, nbits = Log2Floor(d) - 1, prefix = (d >> nbits) & 1;
  const uint32_t offset = (2 + prefix) << nbits;
  return (2 * (nbits - 1) + prefix + 80) | ((d - offset) << 8);
}

/* :216-232 */
static void FStoreMetaBlockHeader(size_t len, int is_uncompressed, size_t* ix, uint8_t* storage) {
  size_t nibbles = 6;
  WriteBits(1, 0, ix, storage);
  if (len <= (1u << 16)) nibbles = 4; else if (len <= (1u << 20)) nibbles = 5;
  WriteBits(2, nibbles - 4, ix, storage);
  WriteBits(nibbles * 4, len - 1, ix, storage);
  WriteBits(1, (uint64_t)is_uncompressed, ix, storage);
}

/* Hash-table refresh after a copy (:354-386 and :410-442).  `first` selects the
   variant used right after the first match of a scan, which for 4-byte hashes
   keys the third store with offset 0 again (:362-363). */
static uint32_t FAfterCopy(const uint8_t* ip, const uint8_t* base_ip, int* table, size_t shift,
                           size_t min_match, int first) {
  uint64_t v; uint32_t cur;
  if (min_match == 4) {
    v = Load64(ip - 3);
    cur = FHashAt(v, 3, shift, 4);
    table[FHashAt(v, 0, shift, 4)] = (int)(ip - base_ip - 3);
    table[FHashAt(v, 1, shift, 4)] = (int)(ip - base_ip - 2);
    table[FHashAt(v, first ? 0 : 2, shift, 4)] = (int)(ip - base_ip - 1);
  } else {
    v = Load64(ip - 5);
    table[FHashAt(v, 0, shift, 6)] = (int)(ip - base_ip - 5);
    table[FHashAt(v, 1, shift, 6)] = (int)(ip - base_ip - 4);
    table[FHashAt(v, 2, shift, 6)] = (int)(ip - base_ip - 3);
    v = Load64(ip - 2);
    cur = FHashAt(v, 2, shift, 6);
    table[FHashAt(v, 0, shift, 6)] = (int)(ip - base_ip - 2);
    table[FHashAt(v, 1, shift, 6)] = (int)(ip - base_ip - 1);
  }
  return cur;
}

/* :234-459 */
static void FCreateCommands(const uint8_t* input, size_t block_size, size_t input_size,
    const uint8_t* base_ip, int* table, size_t table_bits, size_t min_match,
    uint8_t** literals, uint32_t** commands) {
  const uint8_t* ip = input;
  const size_t shift = 64u - table_bits;
  const uint8_t* ip_end = input + block_size;
  const uint8_t* next_emit = input;
  int last_distance = -1;
  if (block_size >= 16) {
    const size_t a = block_size - min_match, b = input_size - 16;
    const uint8_t* ip_limit = input + (a < b ? a : b);
    uint32_t next_hash;
    for (next_hash = FHash(++ip, shift, min_match);;) {
      uint32_t skip = 32;
      const uint8_t* next_ip = ip;
      const uint8_t* candidate;
    trawl:
      do {
        uint32_t hash = next_hash;
        uint32_t step = skip++ >> 5;
        ip = next_ip;
        next_ip = ip + step;
        if (next_ip > ip_limit) goto emit_remainder;
        next_hash = FHash(next_ip, shift, min_match);
        candidate = ip - last_distance;
        if (FIsMatch(ip, candidate, min_match)) {
          if (candidate < ip) { table[hash] = (int)(ip - base_ip); break; }
        }
        candidate = base_ip + table[hash];
        table[hash] = (int)(ip - base_ip);
      } while (!FIsMatch(ip, candidate, min_match));
      if (ip - candidate > F_MAX_DISTANCE) goto trawl;
      {
        const uint8_t* base = ip;
        size_t matched = min_match + FindMatchLength(candidate + min_match, ip + min_match,
                                                     (size_t)(ip_end - ip) - min_match);
        int distance = (int)(base - candidate);
        int insert = (int)(base - next_emit);
        ip += matched;
        *(*commands)++ = FInsertLen((uint32_t)insert);
        memcpy(*literals, next_emit, (size_t)insert);
        *literals += insert;
        if (distance == last_distance) {
          *(*commands)++ = 64;
        } else {
          *(*commands)++ = FDistance((uint32_t)distance);
          last_distance = distance;
        }
        *commands += FCopyLenLastDistance(matched, *commands);
        next_emit = ip;
        if (ip >= ip_limit) goto emit_remainder;
        {
          uint32_t cur = FAfterCopy(ip, base_ip, table, shift, min_match, 1);
          candidate = base_ip + table[cur];
          table[cur] = (int)(ip - base_ip);
        }
      }
      while (ip - candidate <= F_MAX_DISTANCE && FIsMatch(ip, candidate, min_match)) {
        const uint8_t* base = ip;
        size_t matched = min_match + FindMatchLength(candidate + min_match, ip + min_match,
                                                     (size_t)(ip_end - ip) - min_match);
        ip += matched;
        last_distance = (int)(base - candidate);
        *(*commands)++ = FCopyLen(matched);
        *(*commands)++ = FDistance((uint32_t)last_distance);
        next_emit = ip;
        if (ip >= ip_limit) goto emit_remainder;
        {
          uint32_t cur = FAfterCopy(ip, base_ip, table, shift, min_match, 0);
          candidate = base_ip + table[cur];
          table[cur] = (int)(ip - base_ip);
        }
      }
      next_hash = FHash(++ip, shift, min_match);
    }
  }
emit_remainder:
  if (next_emit < ip_end) {
    const uint32_t insert = (uint32_t)(ip_end - next_emit);
    *(*commands)++ = FInsertLen(insert);
    memcpy(*literals, next_emit, insert);
    *literals += insert;
  }
}

/* Working-alphabet code -> the order in which BuildAndStoreCommandPrefixCode
   (:56-104) lines the 64 command codes up for canonical code assignment, and the
   full 704-symbol alphabet slot each occupies. */
static const uint8_t kFOrder[64] = {
  24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47,
  0, 1, 2, 3, 4, 5, 6, 7, 48, 49, 50, 51, 52, 53, 54, 55, 8, 9, 10, 11, 12, 13, 14, 15,
  56, 57, 58, 59, 60, 61, 62, 63, 16, 17, 18, 19, 20, 21, 22, 23};

typedef struct {
  uint32_t lit_histo[256];
  uint8_t lit_depth[256];
  uint16_t lit_bits[256];
  uint32_t cmd_histo[128];
  uint8_t cmd_depth[128];
  uint16_t cmd_bits[128];
  HTree tree[2 * 704 + 1];
} FArena;

static void FBuildAndStoreCommandPrefixCode(FArena* s, size_t* ix, uint8_t* storage) {
  uint8_t tmp_depth[704];
  uint16_t tmp_bits[64];
  size_t i;
  CreateHuffmanTree(s->cmd_histo, 64, 15, s->tree, s->cmd_depth);
  CreateHuffmanTree(&s->cmd_histo[64], 64, 14, s->tree, &s->cmd_depth[64]);
  for (i = 0; i < 64; ++i) tmp_depth[i] = s->cmd_depth[kFOrder[i]];
  memset(tmp_bits, 0, sizeof(tmp_bits));
  ConvertBitDepthsToSymbols(tmp_depth, 64, tmp_bits);
  for (i = 0; i < 64; ++i) s->cmd_bits[kFOrder[i]] = tmp_bits[i];
  ConvertBitDepthsToSymbols(&s->cmd_depth[64], 64, &s->cmd_bits[64]);
  memset(tmp_depth, 0, sizeof(tmp_depth));
  for (i = 0; i < 8; ++i) {
    tmp_depth[i] = s->cmd_depth[24 + i];
    tmp_depth[64 + i] = s->cmd_depth[32 + i];
    tmp_depth[128 + i] = s->cmd_depth[40 + i];
    tmp_depth[192 + i] = s->cmd_depth[48 + i];
    tmp_depth[384 + i] = s->cmd_depth[56 + i];
  }
  for (i = 0; i < 8; ++i) {
    tmp_depth[128 + 8 * i] = s->cmd_depth[i];
    tmp_depth[256 + 8 * i] = s->cmd_depth[8 + i];
    tmp_depth[448 + 8 * i] = s->cmd_depth[16 + i];
  }
  StoreHuffmanTree(tmp_depth, 704, s->tree, ix, storage);
  StoreHuffmanTree(&s->cmd_depth[64], 64, s->tree, ix, storage);
}

static uint32_t FNumExtraBits(uint32_t code) {
  static const uint8_t ins[24] = {0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 12, 14, 24};
  static const uint8_t cpy[24] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 24};
  if (code < 24) return ins[code];
  if (code < 40) return cpy[code - 24];
  if (code < 64) return cpy[code - 40];
  if (code < 80) return 0;
  return (code - 80) / 2 + 1;
}
static const uint32_t kFInsertOffset[24] = {0, 1, 2, 3, 4, 5, 6, 8, 10, 14, 18, 26, 34, 50, 66, 98,
    130, 194, 322, 578, 1090, 2114, 6210, 22594};

/* :461-522 */
static void FStoreCommands(FArena* s, const uint8_t* literals, size_t num_literals,
    const uint32_t* commands, size_t num_commands, size_t* ix, uint8_t* storage) {
  size_t i;
  memset(s->lit_histo, 0, sizeof(s->lit_histo));
  memset(s->cmd_depth, 0, sizeof(s->cmd_depth));
  memset(s->cmd_bits, 0, sizeof(s->cmd_bits));
  memset(s->cmd_histo, 0, sizeof(s->cmd_histo));
  for (i = 0; i < num_literals; ++i) ++s->lit_histo[literals[i]];
  BuildAndStoreHuffmanTreeFast(s->tree, s->lit_histo, num_literals, 8, s->lit_depth, s->lit_bits, ix, storage);
  for (i = 0; i < num_commands; ++i) ++s->cmd_histo[commands[i] & 0xFF];
  s->cmd_histo[1] += 1;
  s->cmd_histo[2] += 1;
  s->cmd_histo[64] += 1;
  s->cmd_histo[84] += 1;
  FBuildAndStoreCommandPrefixCode(s, ix, storage);
  for (i = 0; i < num_commands; ++i) {
    const uint32_t cmd = commands[i], code = cmd & 0xFF, extra = cmd >> 8;
    WriteBits(s->cmd_depth[code], s->cmd_bits[code], ix, storage);
    WriteBits(FNumExtraBits(code), extra, ix, storage);
    if (code < 24) {
      uint32_t insert = kFInsertOffset[code] + extra, j;
      for (j = 0; j < insert; ++j) {
        const uint8_t lit = *literals++;
        WriteBits(s->lit_depth[lit], s->lit_bits[lit], ix, storage);
      }
    }
  }
}

/* :524-544 */
static int FShouldCompress(FArena* s, const uint8_t* input, size_t input_size, size_t num_literals) {
  double corpus_size = (double)input_size;
  if ((double)num_literals < 0.98 * corpus_size) return 1;
  {
    const double max_total_bit_cost = corpus_size * 8 * 0.98 / 43;
    size_t i;
    memset(s->lit_histo, 0, sizeof(s->lit_histo));
    for (i = 0; i < input_size; i += 43) ++s->lit_histo[input[i]];
    return BitsEntropy(s->lit_histo, 256) < max_total_bit_cost;
  }
}

/* :554-562 */
static void FEmitUncompressedMetaBlock(const uint8_t* input, size_t input_size, size_t* ix, uint8_t* storage) {
  FStoreMetaBlockHeader(input_size, 1, ix, storage);
  *ix = (*ix + 7u) & ~(size_t)7u;
  memcpy(&storage[*ix >> 3], input, input_size);
  *ix += input_size << 3;
  storage[*ix >> 3] = 0;
}

/* BrotliCompressFragmentTwoPass :564-641 */
static void FCompressFragmentTwoPass(FArena* s, const uint8_t* input, size_t input_size, int is_last,
    uint32_t* command_buf, uint8_t* literal_buf, int* table, size_t table_size,
    size_t* ix, uint8_t* storage) {
  const size_t initial_ix = *ix;
  const size_t table_bits = Log2Floor(table_size);
  const size_t min_match = table_bits <= 15 ? 4 : 6;
  const uint8_t* base_ip = input;
  const uint8_t* in = input;
  size_t left = input_size;
  while (left > 0) {
    size_t block_size = left < F_BLOCK ? left : F_BLOCK;
    uint32_t* commands = command_buf;
    uint8_t* literals = literal_buf;
    size_t num_literals;
    FCreateCommands(in, block_size, left, base_ip, table, table_bits, min_match, &literals, &commands);
    num_literals = (size_t)(literals - literal_buf);
    if (FShouldCompress(s, in, block_size, num_literals)) {
      FStoreMetaBlockHeader(block_size, 0, ix, storage);
      WriteBits(13, 0, ix, storage);
      FStoreCommands(s, literal_buf, num_literals, command_buf, (size_t)(commands - command_buf), ix, storage);
    } else {
      FEmitUncompressedMetaBlock(in, block_size, ix, storage);
    }
    in += block_size;
    left -= block_size;
  }
  if (*ix - initial_ix > 31 + (input_size << 3)) {
    storage[initial_ix >> 3] &= (uint8_t)((1u << (initial_ix & 7)) - 1);   /* RewindBitPosition :546-552 */
    *ix = initial_ix;
    FEmitUncompressedMetaBlock(input, input_size, ix, storage);
  }
  if (is_last) {
    WriteBits(1, 1, ix, storage);
    WriteBits(1, 1, ix, storage);
    *ix = (*ix + 7u) & ~(size_t)7u;
  }
}

/* One encoder instance at quality 1 driven through BrotliEncoderCompressStream
   (encode.c:1425-1547): call k feeds call_sizes[k] bytes with operation
   call_ops[k] (0 PROCESS, 1 FLUSH, 2 FINISH) and an unbounded output buffer, so
   each call is cut into fragments of min(1 << lgwin, bytes left in the call).
   A one-shot BrotliEncoderCompress is the single call {len, FINISH}. */
size_t oracle_encode_fast(const uint8_t* in, size_t len, int lgwin,
    const uint64_t* call_sizes, const uint8_t* call_ops, size_t ncalls,
    uint8_t* out, size_t out_cap) {
  FArena* s;
  uint32_t* command_buf;
  uint8_t* literal_buf;
  int* table;
  uint8_t* storage;
  uint16_t last_bytes;
  uint8_t last_bytes_bits;
  size_t out_len = 0, k, pos = 0;
  const size_t block_limit = (size_t)1 << lgwin;
  int hl = lgwin < 18 ? 18 : lgwin;            /* encode.c:670-674 */
  int finished = 0, overflow = 0;
  FastStaticInit();
  s = (FArena*)malloc(sizeof(FArena));
  command_buf = (uint32_t*)malloc(F_BLOCK * 4);
  literal_buf = (uint8_t*)malloc(F_BLOCK);
  table = (int*)malloc(sizeof(int) << 17);
  storage = (uint8_t*)malloc(2 * block_limit + 503 + 16);
  last_bytes = (uint16_t)(((hl - 17) << 1) | 1); last_bytes_bits = 4;   /* EncodeWindowBits, lgwin > 17 */
  for (k = 0; k < ncalls && !finished && !overflow; ++k) {
    size_t avail = (size_t)call_sizes[k];
    const int op = call_ops[k];
    if (pos + avail > len) { overflow = 1; break; }
    for (;;) {
      if (avail != 0 || op != 0) {
        size_t block_size = avail < block_limit ? avail : block_limit;
        int is_last = (avail == block_size) && op == 2;
        int force_flush = (avail == block_size) && op == 1;
        size_t ix = last_bytes_bits, table_size = 256, out_bytes;
        if (!(force_flush && block_size == 0)) {
          storage[0] = (uint8_t)last_bytes;
          storage[1] = (uint8_t)(last_bytes >> 8);
          while (table_size < ((size_t)1 << 17) && table_size < block_size) table_size <<= 1;   /* :148-154 */
          memset(table, 0, table_size * sizeof(int));
          FCompressFragmentTwoPass(s, in + pos, block_size, is_last, command_buf, literal_buf,
                                   table, table_size, &ix, storage);
          pos += block_size; avail -= block_size;
          out_bytes = ix >> 3;
          if (out_len + out_bytes > out_cap) { overflow = 1; break; }
          memcpy(out + out_len, storage, out_bytes);
          out_len += out_bytes;
          last_bytes = (uint16_t)storage[ix >> 3];
          last_bytes_bits = (uint8_t)(ix & 7u);
        }
        if (force_flush) {                    /* InjectBytePaddingBlock, :1356-1380 */
          if (last_bytes_bits != 0) {
            uint32_t seal = last_bytes;
            size_t seal_bits = last_bytes_bits, nb, q;
            seal |= 0x6u << seal_bits;
            seal_bits += 6;
            nb = (seal_bits + 7) >> 3;
            if (out_len + nb > out_cap) { overflow = 1; break; }
            for (q = 0; q < nb; ++q) out[out_len++] = (uint8_t)(seal >> (8 * q));
            last_bytes = 0; last_bytes_bits = 0;
          }
          break;
        }
        if (is_last) { finished = 1; break; }
        continue;
      }
      break;
    }
  }
  free(s); free(command_buf); free(literal_buf); free(table); free(storage);
  return overflow ? 0 : out_len;
}
