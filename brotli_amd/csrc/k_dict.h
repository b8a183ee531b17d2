// brotli_amd/csrc/k_dict.h — attached (compound) dictionaries on the device: the lookup the
// reference runs after every FindLongestMatch once BrotliEncoderAttachPreparedDictionary has been
// called (LookupCompoundDictionaryMatch / FindCompoundDictionaryMatch, c/enc/hash.h:526-634,
// 703-717; call sites c/enc/backward_references_inc.h:115-119, 147-152).
//
// Layout (ours, not the reference's PreparedDictionary): a chunk is its raw bytes plus a CSR index
// over the 40-bit hash of the eight bytes at every position — starts[key] .. starts[key + 1] are
// the positions of that key, NEWEST FIRST, at most 32 of them (bucket_limit, compound_dictionary.c:
// 155-173).  The reference keeps 16-bit heads per key inside "slots" and shrinks a slot's limit if
// the 16-bit offset would overflow (:62-90); with its own parameters (bucket_bits - slot_bits = 10,
// 32 per bucket) a slot holds at most 32 768 items, so that never happens and every key keeps
// min(count, 32) positions — which is what the CSR index stores (host side: dict_index.h).
//
// One wave, all lanes active, every decision wave-uniform: lanes 0..31 take the key's items, lanes
// 32..35 the four distance-cache entries that point into the chunk; each lane measures its match
// length against the bytes at the current position (32 bytes in registers, longer matches extended
// by the whole wave), then the candidates are walked in the reference's order with its gate — a
// candidate is only looked at if it agrees with the input on the four bytes ending at best_len —
// decided from the measured lengths (an explicit compare only when a shorter candidate could still
// pass the gate).
#ifndef BROTLI_AMD_CSRC_K_DICT_H_
#define BROTLI_AMD_CSRC_K_DICT_H_

#include "device_common.h"

struct SearchResult {
  uint32_t len, distance, score;
  int32_t len_code_delta;
};

#define DICT_MAX_CHUNKS 15          // SHARED_BROTLI_MAX_COMPOUND_DICTS
#define DICT_BUCKET_LIMIT 32u
#define DICT_SOURCE_SLACK 64u       // readable bytes behind a chunk's source on the device

struct DictChunk {
  const uint8_t* source;      // source_size bytes (+ DICT_SOURCE_SLACK)
  const uint32_t* starts;     // [(1 << bucket_bits) + 1]
  const uint32_t* items;      // [starts[1 << bucket_bits]]
  uint32_t source_size;
  uint32_t bucket_bits;       // 17 .. 22
  uint32_t offset;            // chunk_offsets[d]: bytes of the chunks attached before this one
  uint32_t pad;
};
struct CompoundDict {
  uint32_t num_chunks, total_size;
  DictChunk chunks[DICT_MAX_CHUNKS];
};

// kPreparedDictionaryHashMul64Long, hash_bits = 40 (compound_dictionary.h:31-32, compound_dictionary.c:159)
DEV uint32_t dict_key(uint64_t x, uint32_t bucket_bits) {
  return (uint32_t)(((x & 0xFFFFFFFFFFull) * 0x1FE35A7BD3579BD3ull) >> (64u - bucket_bits));
}

struct DB32 { uint64_t q[4]; };
DEV uint32_t dict_prefix32(const DB32& a, const DB32& b) {
  uint32_t n = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint64_t x = a.q[i] ^ b.q[i];
    if (x != 0) return n + ((uint32_t)dev_ctz64(x) >> 3);
    n += 8;
  }
  return n;
}

// `cur` = the bytes at the current position (max_length of them are input; BROTLI_AMD_INPUT_SLACK is
// readable behind), cur_masked = position & ring mask, dc = the four last distances,
// max_ring_distance = the `dictionary_start` of the call site, max_distance = params->dist.max_distance.
// Only the hashers with a compound variant call this (backward_references.c:194-243).
// (The distance cache comes as four VALUES: indexed by the lane through a pointer it pinned the caller's whole shard
//  state in scratch memory — k_parse4.h, q_dc_entry.)
DEV int32_t dict_dc_pick(int32_t d0, int32_t d1, int32_t d2, int32_t d3, int i) {
  int32_t d = d3;
  d = i == 2 ? d2 : d;
  d = i == 1 ? d1 : d;
  d = i == 0 ? d0 : d;
  return d;
}
DEV void compound_lookup(const CompoundDict* cd, const uint8_t* cur, uint32_t cur_masked, uint32_t ring_mask,
                         int32_t dc0, int32_t dc1, int32_t dc2, int32_t dc3, uint32_t max_length, uint32_t max_ring_distance,
                         uint32_t max_distance, SearchResult& out) {
  const int lane = wave_lane();
  DB32 cur32;
  __builtin_memcpy(&cur32, cur, 32);
  const uint32_t nchunks = cd->num_chunks, total = cd->total_size;
  for (uint32_t d = 0; d < nchunks; ++d) {
    const DictChunk& ch = cd->chunks[d];
    const uint8_t* source = ch.source;
    const uint32_t source_size = ch.source_size;
    // base_offset - chunk_offsets[d], hash.h:709-715
    const uint32_t distance_offset = max_ring_distance + total - ch.offset;
    const uint32_t boundary = distance_offset - source_size;
    const uint32_t key = dict_key(cur32.q[0], ch.bucket_bits);
    const uint32_t s0 = ch.starts[key], n_items = ch.starts[key + 1] - s0;   // <= 32

    // every lane: its candidate's offset into the chunk, then the exact match length
    bool cand = false;
    uint32_t offset = 0;
    if (lane < 32) {
      if ((uint32_t)lane < n_items) { offset = ch.items[s0 + (uint32_t)lane]; cand = true; }
    } else if (lane < 36) {
      const int32_t dcl = dict_dc_pick(dc0, dc1, dc2, dc3, lane - 32);
      const uint32_t distance = (uint32_t)dcl;
      if (dcl > 0 && distance > boundary && distance <= distance_offset) {
        offset = distance_offset - distance;
        cand = offset < source_size;
      }
    }
    uint32_t limit = 0, len = 0;
    bool need_ext = false;
    if (cand) {
      limit = umin(source_size - offset, max_length);
      DB32 src32;
      __builtin_memcpy(&src32, source + offset, 32);
      const uint32_t m = dict_prefix32(cur32, src32);
      len = umin(m, limit);
      need_ext = m == 32u && limit > 32u;
    }
    uint64_t ext = wave_ballot(need_ext);
    while (ext) {               // rare: longer than 32 bytes, 512 bytes per step by the whole wave
      const int j = dev_ctz64(ext);
      ext &= ext - 1;
      const uint32_t oj = wave_bcast(offset, j), lj = wave_bcast(limit, j);
      uint32_t off = 32, L = lj;
      for (;;) {
        const uint32_t o = off + (uint32_t)lane * 8u;
        uint64_t x = 0;
        if (o < lj) x = ld64(cur + o) ^ ld64(source + oj + o);
        const uint64_t mm = wave_ballot(x != 0);
        if (mm) {
          const int f = dev_ctz64(mm);
          const uint64_t xf = wave_bcast64(x, f);
          L = umin(off + (uint32_t)f * 8u + ((uint32_t)dev_ctz64(xf) >> 3), lj);
          break;
        }
        off += 512u;
        if (off >= lj) break;
      }
      if (lane == j) len = L;
    }

    // the reference's walk, wave-uniform
    uint32_t best_score = out.score, best_len = out.len;
    const uint64_t dc_mask = wave_ballot(cand && lane >= 32);
    for (int i = 0; i < 4; ++i) {                                   // hash.h:569-593
      if (!((dc_mask >> (32 + i)) & 1ull)) continue;
      const uint32_t len_i = wave_bcast(len, 32 + i);
      if (len_i < 2) continue;
      uint32_t score = 135u * len_i + 1935u;
      if (!(best_score < score)) continue;
      if (i != 0) score -= 39u + ((0x1CA10u >> (i & 0xE)) & 0xEu);
      if (!(best_score < score)) continue;
      best_score = score;
      if (len_i > best_len) best_len = len_i;
      out.len = len_i;
      out.len_code_delta = 0;
      out.distance = (uint32_t)dict_dc_pick(dc0, dc1, dc2, dc3, i);
      out.score = best_score;
    }
    if (best_len < 3) best_len = 3;
    // (only the items that can change anything are visited, in order: four bytes long, inside the distance limit, with
    //  a score above the one the walk starts with — the best score never falls, the other conditions have no side effect)
    uint64_t worth = wave_ballot(lane < 32 && cand && len >= 4u && distance_offset - offset <= max_distance &&
                                 1920u + 135u * len - 30u * log2floor(distance_offset - offset) > best_score);
    while (worth != 0ull) {                                         // hash.h:599-633
      const int t = dev_ctz64(worth);
      worth &= worth - 1ull;
      const uint32_t off_t = wave_bcast(offset, t);
      const uint32_t len_t = wave_bcast(len, t);
      const uint32_t limit_t = wave_bcast(limit, t);
      const uint32_t distance = distance_offset - off_t;
      if (cur_masked + best_len > ring_mask || best_len >= limit_t) continue;
      if (len_t <= best_len) {
        // the gate compares the four bytes ending at best_len: a candidate that differs at or
        // after best_len - 3 fails it; one that differs earlier is compared for real
        if (len_t + 3u >= best_len) continue;
        if (ld32(cur + best_len - 3u) != ld32(source + off_t + best_len - 3u)) continue;
      }
      if (len_t < 4) continue;
      const uint32_t score = 1920u + 135u * len_t - 30u * log2floor(distance);
      if (!(best_score < score)) continue;
      best_score = score;
      best_len = len_t;
      out.len = len_t;
      out.len_code_delta = 0;
      out.distance = distance;
      out.score = best_score;
    }
  }
}

// ---- the same for 16-lane groups (k_parse4.h: up to four shards per wave, each group at a position of its own) ----
// `want`: this group has a search to improve.  Lane t of a group takes distance-cache entry t (t < 4), then item t and
// item 16 + t of the key; every cross-lane step is taken by all four groups together, what a group decides from it is
// replicated in its lanes.  The walk only visits the candidates that can change anything — at least four bytes long,
// inside the distance limit: the reference's other conditions have no side effect (hash.h:599-633).
DEV uint32_t dict_extend_from32(const uint8_t* a, const uint8_t* b, uint32_t limit) {
  uint32_t off = 32;
  while (off + 8 <= limit) {
    const uint64_t x = ld64(a + off) ^ ld64(b + off);
    if (x) return off + ((uint32_t)dev_ctz64(x) >> 3);
    off += 8;
  }
  while (off < limit && a[off] == b[off]) ++off;
  return off;
}
DEV uint32_t dict_group_from(uint32_t v, int src_t) {      // lane (group base | src_t)'s value, src_t per group
  return wave_shfl(v, (wave_lane() & 48) | src_t);
}
// (DictAhead: the first chunk's key range and this lane's two items, requested by the caller while its own search was
//  under way — the bytes at the position are all they depend on —, so that of the lookup's three dependent round
//  trips, key range -> items -> dictionary bytes, only the last one is left behind the search.)
struct DictAhead { uint32_t s0, n, off0, off1; };
DEV DictAhead dict_ahead16(const CompoundDict* cd, bool want, uint64_t first8) {
  DictAhead a;
  a.s0 = a.n = a.off0 = a.off1 = 0;
  if (want) {
    const DictChunk& ch = cd->chunks[0];
    const uint32_t key = dict_key(first8, ch.bucket_bits);
    a.s0 = ch.starts[key];
    a.n = ch.starts[key + 1] - a.s0;
    const uint32_t t = (uint32_t)(wave_lane() & 15);
    if (t < a.n) a.off0 = ch.items[a.s0 + t];
    if (16u + t < a.n) a.off1 = ch.items[a.s0 + 16u + t];
  }
  return a;
}
DEV void compound_lookup16(const CompoundDict* cd, bool want, const uint8_t* cur, uint32_t cur_masked, uint32_t ring_mask,
                           int32_t dc0, int32_t dc1, int32_t dc2, int32_t dc3, uint32_t max_length,
                           uint32_t max_ring_distance, uint32_t max_distance, SearchResult& out, const DictAhead& ahead) {
  const int t = wave_lane() & 15;
  const int gshift = wave_lane() & 48;
  DB32 cur32;
  __builtin_memcpy(&cur32, cur, 32);
  const uint32_t nchunks = cd->num_chunks, total = cd->total_size;
  for (uint32_t d = 0; d < nchunks; ++d) {
    const DictChunk& ch = cd->chunks[d];
    const uint8_t* source = ch.source;
    const uint32_t source_size = ch.source_size;
    const uint32_t distance_offset = max_ring_distance + total - ch.offset;
    const uint32_t boundary = distance_offset - source_size;
    const uint32_t key = dict_key(cur32.q[0], ch.bucket_bits);
    uint32_t s0 = 0, n_items = 0;
    if (d == 0u) { s0 = ahead.s0; n_items = want ? ahead.n : 0u; }
    else if (want) { s0 = ch.starts[key]; n_items = ch.starts[key + 1] - s0; }   // <= 32
    uint32_t best_score = out.score, best_len = out.len;
    // ---- the four last distances that point into the chunk (hash.h:569-593) ----
    {
      bool cand = false;
      uint32_t offset = 0, len = 0;
      if (want && t < 4) {
        const int32_t dcl = dict_dc_pick(dc0, dc1, dc2, dc3, t);
        const uint32_t distance = (uint32_t)dcl;
        if (dcl > 0 && distance > boundary && distance <= distance_offset) {
          offset = distance_offset - distance;
          cand = offset < source_size;
        }
      }
      if (cand) {
        const uint32_t limit = umin(source_size - offset, max_length);
        DB32 src32;
        __builtin_memcpy(&src32, source + offset, 32);
        const uint32_t m = dict_prefix32(cur32, src32);
        len = umin(m, limit);
        if (m == 32u && limit > 32u) len = dict_extend_from32(cur, source + offset, limit);
      }
      const uint64_t any4 = wave_ballot(cand);
      const uint32_t mask4 = (uint32_t)(any4 >> gshift) & 0xFu;
      uint32_t l4[4] = {0, 0, 0, 0};
      if (any4 != 0ull) {                      // (rare: a last distance points into a dictionary only behind a match there)
#pragma unroll
        for (int i = 0; i < 4; ++i) l4[i] = dict_group_from(len, i);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (!((mask4 >> i) & 1u)) continue;
        const uint32_t len_i = l4[i];
        if (len_i < 2) continue;
        uint32_t score = 135u * len_i + 1935u;
        if (!(best_score < score)) continue;
        if (i != 0) score -= 39u + ((0x1CA10u >> (i & 0xE)) & 0xEu);
        if (!(best_score < score)) continue;
        best_score = score;
        if (len_i > best_len) best_len = len_i;
        out.len = len_i;
        out.len_code_delta = 0;
        out.distance = (uint32_t)dict_dc_pick(dc0, dc1, dc2, dc3, i);
        out.score = best_score;
      }
    }
    if (best_len < 3) best_len = 3;
    // ---- the key's items, newest first, sixteen at a time ----
    for (uint32_t h = 0; h < 2u; ++h) {
      if (!(wave_ballot(want && n_items > 16u * h) != 0ull)) break;
      const uint32_t it = 16u * h + (uint32_t)t;
      const bool cand = want && it < n_items;
      uint32_t offset = 0, limit = 0, len = 0;
      if (cand) {
        offset = d == 0u ? (h == 0u ? ahead.off0 : ahead.off1) : ch.items[s0 + it];
        limit = umin(source_size - offset, max_length);
        DB32 src32;
        __builtin_memcpy(&src32, source + offset, 32);
        const uint32_t m = dict_prefix32(cur32, src32);
        len = umin(m, limit);
        if (m == 32u && limit > 32u) len = dict_extend_from32(cur, source + offset, limit);
      }
      // (an item is taken only with a score above the best so far, which never falls: what does not beat the score
      //  the walk starts with is left out as well — most of a key's items, once the search itself found something)
      const bool worth = cand && len >= 4u && distance_offset - offset <= max_distance &&
                         1920u + 135u * len - 30u * log2floor(distance_offset - offset) > best_score;
      uint32_t m16 = (uint32_t)(wave_ballot(worth) >> gshift) & 0xFFFFu;
      while (wave_ballot(m16 != 0u) != 0ull) {
        const bool act = m16 != 0u;
        const int tt = act ? dev_ctz32(m16) : 0;
        m16 &= m16 - 1u;
        const uint32_t off_t = dict_group_from(offset, tt), len_t = dict_group_from(len, tt), limit_t = dict_group_from(limit, tt);
        if (!act) continue;
        const uint32_t distance = distance_offset - off_t;
        if (cur_masked + best_len > ring_mask || best_len >= limit_t) continue;
        if (len_t <= best_len) {
          if (len_t + 3u >= best_len) continue;
          if (ld32(cur + best_len - 3u) != ld32(source + off_t + best_len - 3u)) continue;
        }
        const uint32_t score = 1920u + 135u * len_t - 30u * log2floor(distance);
        if (!(best_score < score)) continue;
        best_score = score;
        best_len = len_t;
        out.len = len_t;
        out.len_code_delta = 0;
        out.distance = distance;
        out.score = best_score;
      }
    }
  }
}

// compound_extend for a 16-lane group: 16 bytes a step.  `run`: this group extends.
DEV uint32_t compound_extend16(const CompoundDict* cd, bool run, const uint8_t* data_at_pos, uint32_t bytes,
                               uint32_t cmd_dist, uint32_t max_distance, uint32_t last_copy_len) {
  const int t = wave_lane() & 15;
  const int gshift = wave_lane() & 48;
  const uint32_t total = cd->total_size;
  run = run && (cmd_dist - max_distance - 1u) < total && last_copy_len < cmd_dist - max_distance;
  uint32_t address = run ? total - (cmd_dist - max_distance) + last_copy_len : 0u;
  uint32_t gained = 0;
  run = run && bytes != 0 && address < total;
  while (wave_ballot(run) != 0ull) {
    uint32_t room = 0;
    bool ok = false;
    if (run) {
      uint32_t k = 0;
      while (address >= cd->chunks[k].offset + cd->chunks[k].source_size) ++k;
      const DictChunk& ch = cd->chunks[k];
      const uint32_t in_chunk = address - ch.offset;
      room = umin(umin(ch.source_size - in_chunk, bytes), 16u);
      ok = (uint32_t)t < room && data_at_pos[gained + (uint32_t)t] == ch.source[in_chunk + (uint32_t)t];
    }
    const uint32_t m = (uint32_t)(wave_ballot(ok) >> gshift) & 0xFFFFu;
    if (run) {
      const uint32_t n = (m == 0xFFFFu) ? 16u : (uint32_t)dev_ctz32(~m);
      gained += n;
      bytes -= n;
      address += n;
      if (n < room || bytes == 0 || address >= total) run = false;
    }
  }
  return gained;
}

// The part of ExtendLastCommand that continues a copy inside the attached dictionary
// (c/enc/encode.c:930-961): the last command's distance points `cmd_dist - max_distance` bytes
// before the end of the compound dictionary; the copy is extended while input and dictionary agree,
// across chunk borders, until the dictionary ends.  Returns the bytes gained.
DEV uint32_t compound_extend(const CompoundDict* cd, const uint8_t* data_at_pos, uint32_t bytes,
                             uint32_t cmd_dist, uint32_t max_distance, uint32_t last_copy_len) {
  const int lane = wave_lane();
  const uint32_t total = cd->total_size;
  if (!((cmd_dist - max_distance - 1u) < total && last_copy_len < cmd_dist - max_distance)) return 0;
  uint32_t address = total - (cmd_dist - max_distance) + last_copy_len;
  uint32_t gained = 0;
  while (bytes != 0 && address < total) {
    uint32_t k = 0;
    while (address >= cd->chunks[k].offset + cd->chunks[k].source_size) ++k;
    const DictChunk& ch = cd->chunks[k];
    const uint32_t in_chunk = address - ch.offset;
    const uint32_t room = umin(umin(ch.source_size - in_chunk, bytes), 64u);
    const bool ok = (uint32_t)lane < room && data_at_pos[gained + (uint32_t)lane] == ch.source[in_chunk + (uint32_t)lane];
    const uint64_t m = wave_ballot(ok);
    const uint32_t run = (m == ~0ull) ? 64u : (uint32_t)dev_ctz64(~m);
    gained += run;
    bytes -= run;
    address += run;
    if (run < room) break;
  }
  return gained;
}

#endif  // BROTLI_AMD_CSRC_K_DICT_H_
