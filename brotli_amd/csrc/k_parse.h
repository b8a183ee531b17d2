// brotli_amd/csrc/k_parse.h — K1: LZ77 parse of one encoder shard by one
// 64-lane wavefront (match search over the sliding window).
//
// Semantics: CreateBackwardReferences (c/enc/backward_references_inc.h:10-242)
// with the tag-filtered hashers H68 / H58
// (c/enc/hash_longest_match64_simd_inc.h:114-302,
//  c/enc/hash_longest_match_simd_inc.h), the static-dictionary probe
// (c/enc/hash.h:140-202) and the per-block glue of EncodeData
// (c/enc/encode.c:985-1173: stitch :1103, ExtendLastCommand :905-971,
// meta-block cut rule :1141-1173).  Output is bit-exact with the reference.
//
// Design (not a translation):
//  * One wave walks one shard; every decision is wave-uniform (SGPR), lanes
//    are used inside a search step:
//      lanes  0..15  the 16 slots of the bucket row of position p
//      lanes 16..19+ the distance-cache candidates of p
//      lanes 32..47 / 48.. the same for position p+1, evaluated speculatively
//    The reference searches p+1 next in both of its paths (lazy matching after
//    a hit, plain advance after a miss), with parameters that do not depend on
//    the outcome at p.  So (p, p+1) are evaluated in one memory round trip and
//    the result for p+1 is consumed only where the reference would have
//    computed it; the one true dependency (p's own insertion into the bucket
//    of p+1, when both hash to one key) is detected and recomputed.
//  * A bucket is ONE 128-byte record (slots, tags, tag2 fingerprints, count),
//    fetched by one coalesced 128 B load; a store is a single-line update.
//  * Candidates are scored from their exact match length (32 bytes compared
//    in registers, wave-cooperative extension beyond).  The reference's
//    order-dependent "gate" (compare at the evolving best_len, :258-292) is
//    reproduced by a short uniform resolve over the surviving candidates in
//    bucket order; the explicit byte compare is only needed in the rare case
//    len <= best_len with a winning score.
#ifndef BROTLI_AMD_CSRC_K_PARSE_H_
#define BROTLI_AMD_CSRC_K_PARSE_H_

#include "device_common.h"
#include "k_dict.h"

#define K_MIN_SCORE (1920u + 100u)   // BROTLI_SCORE_BASE + 100, hash.h:102-105
#define K_DIST_MAX_DISTANCE 0x3FFFFFCu   // params->dist.max_distance at NPOSTFIX = NDIRECT = 0

struct ParseCtx {
  const uint8_t* data;      // shard byte 0
  uint8_t* table;           // hash records
  const DeviceTables* T;
  int hasher_type, bucket_bits, ndist, htl;
  uint32_t ring_mask, ring_size;
  uint32_t max_backward_limit;
  uint32_t stream_offset;
  uint32_t pos_end;         // end of the block being parsed == bytes "in the ring"
  uint32_t dict_lookups, dict_matches;
  int32_t dc[4];
  bool pair_enabled;
  const CompoundDict* cd;   // attached dictionaries (k_dict.h), nullptr = none
  uint32_t gap;             // their total size: shifts the static dictionary and the distance-code
                            // limit (backward_references_inc.h:31, 114, 173)
};

// Byte the reference would read at ring index (x & mask) for x <= pos_end:
// real data below pos_end; at pos_end the 7 zero bytes written after the
// block on the first lap (encode.c:879-893) or the stale byte of the previous
// lap (c/enc/ringbuffer.h:103-159).
DEV uint32_t ring_byte(const ParseCtx& c, uint32_t x) {
  if (x < c.pos_end) return c.data[x];
  if (c.pos_end <= c.ring_mask) return 0;
  return c.data[x - c.ring_size];
}

struct B32 { uint64_t q[4]; };
DEV B32 load_b32(const uint8_t* p) {
  B32 r;
  __builtin_memcpy(&r, p, 32);
  return r;
}
// Length of the common prefix of two 32-byte strings (0..32).
DEV uint32_t common_prefix32(const B32& a, const B32& b) {
  uint32_t n = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint64_t x = a.q[i] ^ b.q[i];
    if (x != 0) return n + ((uint32_t)dev_ctz64(x) >> 3);
    n += 8;
  }
  return n;
}

// hash.h:80-100: the derived entries of the distance cache.
DEV int32_t dist_cache_entry(const ParseCtx& c, int i) {
  // (values, not c.dc[i]: an index through the member's address keeps the whole ParseCtx in scratch memory — k_parse4.h, q_dc_entry)
  const int32_t d0 = c.dc[0], d1 = c.dc[1], d2 = c.dc[2], d3 = c.dc[3];
  if (i < 4) { int32_t d = d3; d = i == 2 ? d2 : d; d = i == 1 ? d1 : d; d = i == 0 ? d0 : d; return d; }
  const int base = i < 10 ? d0 : d1;
  const int k = (i < 10 ? i - 4 : i - 10);
  const int mag = (k >> 1) + 1;
  return (k & 1) ? base + mag : base - mag;
}

// ---- ordered insertion of up to 64 positions ------------------------------
// Store / StoreRange (..64_simd_inc.h:114-137): positions first + i*stride,
// i < count.  Lanes holding the same key are ranked in position order so the
// bucket ends up exactly as after the reference's serial loop.
DEV void store_positions(const ParseCtx& c, uint32_t first, uint32_t count, uint32_t stride) {
  const int lane = wave_lane();
  const bool act = (uint32_t)lane < count;
  const uint32_t pos = first + (uint32_t)lane * stride;
  KeyTag kt;
  kt.key = 0xFFFFFFFFu - (uint32_t)lane;  // inactive lanes never collide
  kt.tag = kt.tag2 = 0;
  if (act) kt = hash_pos(ld64(c.data + pos), c.hasher_type, c.bucket_bits);
  // match-any on the key: lanes_same = lanes with the same key as me.
  uint64_t same = ~0ull;
  for (int b = 0; b < c.bucket_bits; ++b) {
    const bool bit = (kt.key >> b) & 1;
    const uint64_t m = wave_ballot(act && bit);
    same &= bit ? m : ~m;
  }
  const uint64_t actmask = wave_ballot(act);
  same &= actmask;
  uint8_t* rec = c.table + (size_t)kt.key * REC_BYTES;
  uint16_t num = 0;
  if (act) __builtin_memcpy(&num, rec + REC_NUM_DW * 4, 2);
  wave_sync();  // every lane has its bucket count before any lane updates one
  if (act) {
    const uint32_t below = (uint32_t)dev_popc64(same & ((1ull << lane) - 1ull));
    const uint32_t total = (uint32_t)dev_popc64(same);
    const uint32_t s = ((uint32_t)num - below) & 15u;
    // Only the newest 16 of a same-key group survive in a 16-slot ring; older
    // ones would be overwritten by the serial loop, so they do not write.
    if (total - below <= 16u) {
      st32(rec + REC_SLOT_DW * 4 + s * 4, pos);
      st16(rec + REC_TAG2_DW * 4 + s * 2, (uint16_t)kt.tag2);
      rec[REC_TAG_DW * 4 + s] = (uint8_t)kt.tag;
    }
    if (below + 1 == total) st16(rec + REC_NUM_DW * 4, (uint16_t)(num - total));
  }
  wave_sync();
}

// ---- static dictionary probe (hash.h:140-202) -------------------------------
DEV void dict_search(ParseCtx& c, uint32_t P, uint32_t max_length,
                     uint32_t dictionary_start, SearchResult& out) {
  if (c.dict_matches < (c.dict_lookups >> 7)) return;
  const int lane = wave_lane();
  const DeviceTables* T = c.T;
  const uint32_t key0 = ((ld32(c.data + P) * 0x1E35A7BDu) >> (32 - 14)) << 1;
  // lanes 0..23: bytes of probe 0; lanes 32..55: bytes of probe 1.
  const int probe = lane >> 5, b = lane & 31;
  const uint32_t key = key0 + (uint32_t)probe;
  const uint32_t wlen = T->dict_hash_lengths[key];
  const uint32_t widx = T->dict_hash_words[key];
  const uint32_t offset = T->dict_offsets_by_length[wlen & 31] + wlen * widx;
  bool eq = false;
  if ((uint32_t)b < wlen && wlen <= max_length) eq = c.data[P + b] == T->dict[offset + b];
  const uint64_t eqm = wave_ballot(eq);
  for (int i = 0; i < 2; ++i) {
    const uint32_t len = wave_bcast(wlen, i * 32);
    const uint32_t word_idx = wave_bcast(widx, i * 32);
    c.dict_lookups++;
    if (len == 0) continue;
    if (len > max_length) continue;
    const uint32_t m32 = (uint32_t)(eqm >> (i * 32));
    const uint32_t lenmask = len >= 32 ? 0xFFFFFFFFu : ((1u << len) - 1u);
    const uint32_t neq = ~m32 & lenmask;
    const uint32_t matchlen = neq ? (uint32_t)dev_ctz32(neq) : len;
    if (matchlen + 10 <= len || matchlen == 0) continue;
    const uint32_t cut = len - matchlen;
    const uint32_t transform_id = (cut << 2) + (uint32_t)((0x071B520ADA2D3200ull >> (cut * 6)) & 0x3F);
    const uint32_t backward = dictionary_start + 1 + word_idx +
        (transform_id << T->dict_size_bits_by_length[len]);
    if (backward > 0x3FFFFFCu) continue;  // params->dist.max_distance
    const uint32_t score = 1920u + 135u * matchlen - 30u * log2floor(backward);
    if (score < out.score) continue;
    out.len = matchlen;
    out.len_code_delta = (int32_t)len - (int32_t)matchlen;
    out.distance = backward;
    out.score = score;
    c.dict_matches++;
  }
}

// ---- one search step over (posA, posA+1) ------------------------------------
struct PendingB {
  bool valid;          // window result for posA+1 is usable
  uint32_t pos;
  SearchResult sr;     // window candidates only (dictionary probe pending)
  uint32_t key, tag, tag2, num;
};

// Insert `pos` into its bucket after a search (..64_simd_inc.h:293-295).
DEV void insert_searched(const ParseCtx& c, uint32_t pos, uint32_t key, uint32_t tag,
                         uint32_t tag2, uint32_t num) {
  if (wave_lane() == 0) {
    uint8_t* rec = c.table + (size_t)key * REC_BYTES;
    const uint32_t s = num & 15u;
    st32(rec + REC_SLOT_DW * 4 + s * 4, pos);
    st16(rec + REC_TAG2_DW * 4 + s * 2, (uint16_t)tag2);
    rec[REC_TAG_DW * 4 + s] = (uint8_t)tag;
    st16(rec + REC_NUM_DW * 4, (uint16_t)(num - 1u));
  }
  wave_sync();
}

// Uniform, order-exact selection among the candidates of one half.
DEV SearchResult resolve_half(const ParseCtx& c, int half, uint32_t P,
                              uint32_t max_length, uint64_t dc_ok, uint64_t bk_ok,
                              uint32_t v_len, uint32_t v_prev, uint32_t v_score) {
  SearchResult r;
  r.len = 0; r.distance = 0; r.score = K_MIN_SCORE; r.len_code_delta = 0;
  uint32_t best_len = 0;
  const uint32_t cur_masked = P & c.ring_mask;
  uint32_t m = (uint32_t)(dc_ok >> (half * 32 + 16)) & 0xFFFFu;
  while (m) {
    const int i = dev_ctz32(m);
    m &= m - 1;
    const int L = half * 32 + 16 + i;
    const uint32_t len_i = wave_bcast(v_len, L);
    const uint32_t prev_i = wave_bcast(v_prev, L);
    const uint32_t score_i = wave_bcast(v_score, L);
    if (cur_masked + best_len > c.ring_mask) break;
    if ((prev_i & c.ring_mask) + best_len > c.ring_mask) continue;
    if (!(len_i >= 3 || (len_i == 2 && i < 2))) continue;
    if (!(r.score < score_i)) continue;
    bool pass = len_i > best_len;
    if (!pass && !(len_i == best_len && len_i < max_length)) {
      pass = ring_byte(c, P + best_len) == ring_byte(c, prev_i + best_len);
    }
    if (!pass) continue;
    best_len = len_i;
    r.len = len_i; r.distance = P - prev_i; r.score = score_i;
  }
  if (best_len < 3) best_len = 3;
  m = (uint32_t)(bk_ok >> (half * 32)) & 0xFFFFu;
  while (m) {
    const int t = dev_ctz32(m);
    m &= m - 1;
    const int L = half * 32 + t;
    const uint32_t len_j = wave_bcast(v_len, L);
    const uint32_t prev_j = wave_bcast(v_prev, L);
    const uint32_t score_j = wave_bcast(v_score, L);
    if (cur_masked + best_len > c.ring_mask) break;
    if ((prev_j & c.ring_mask) + best_len > c.ring_mask) continue;
    if (!(r.score < score_j)) continue;
    bool pass = len_j > best_len;
    if (!pass) {
      pass = true;
      for (uint32_t k = best_len - 3; k <= best_len; ++k) {
        if (ring_byte(c, P + k) != ring_byte(c, prev_j + k)) { pass = false; break; }
      }
    }
    if (!pass) continue;
    best_len = len_j;
    r.len = len_j; r.distance = P - prev_j; r.score = score_j;
  }
  return r;
}

// Searches posA (result returned, posA inserted, dictionary probed) and, when
// allowed, posA+1 speculatively (window candidates only; see PendingB).
DEV SearchResult search_pair(ParseCtx& c, uint32_t posA, PendingB& B) {
  const int lane = wave_lane();
  const int half = lane >> 5, sub = lane & 31;
  const bool doB = c.pair_enabled && (posA + 1u + (uint32_t)c.htl <= c.pos_end);
  const bool active = half == 0 || doB;
  const uint32_t P = posA + (uint32_t)half;
  const uint8_t* cur = c.data + (active ? P : posA);
  const uint32_t max_length = c.pos_end - P;
  const uint32_t max_backward = umin(P, c.max_backward_limit);

  // Round trip 1: the bucket record (one 128-byte line per half), the bytes
  // at the position itself.
  const B32 cur32 = load_b32(cur);
  const KeyTag kt = hash_pos(cur32.q[0], c.hasher_type, c.bucket_bits);
  const uint8_t* rec = c.table + (size_t)kt.key * REC_BYTES;
  const uint32_t dw = ld32(rec + sub * 4);

  const uint32_t num = wave_shfl(dw, half * 32 + REC_NUM_DW) & 0xFFFFu;
  const uint32_t head = (num + 1u) & 15u;
  const uint32_t n = (65535u - num) & 0xFFFFu;
  const uint32_t s = (head + (uint32_t)sub) & 15u;
  const uint32_t slot = wave_shfl(dw, half * 32 + REC_SLOT_DW + (int)s);
  const uint32_t tgw = wave_shfl(dw, half * 32 + REC_TAG_DW + (int)(s >> 2));
  const uint32_t t2w = wave_shfl(dw, half * 32 + REC_TAG2_DW + (int)(s >> 1));
  const uint32_t tg = (tgw >> ((s & 3u) * 8u)) & 0xFFu;
  const uint32_t t2 = (t2w >> ((s & 1u) * 16u)) & 0xFFFFu;

  const bool is_bk = sub < 16;
  const bool is_dc = sub >= 16 && sub < 16 + c.ndist;
  bool cand = false;
  uint32_t prev_ix = 0;
  // bucket slots, newest first (:246-262); slots beyond the fill count masked
  // (:250-257); the scan stops at the first tag match out of range (:264-266).
  const bool tagmatch = active && is_bk && (n >= 16u || (uint32_t)sub < n) && tg == kt.tag;
  const bool far = tagmatch && (P - slot) > max_backward;
  const uint64_t farmask = wave_ballot(far);
  if (is_bk) {
    const uint32_t fm = (uint32_t)(farmask >> (half * 32)) & 0xFFFFu;
    const uint32_t first_far = fm ? (uint32_t)dev_ctz32(fm) : 32u;
    cand = tagmatch && (uint32_t)sub < first_far && t2 == kt.tag2;
    prev_ix = slot;
  } else if (is_dc) {
    const int32_t backward = dist_cache_entry(c, sub - 16);
    cand = active && backward > 0 && (uint32_t)backward <= max_backward;
    prev_ix = P - (uint32_t)backward;
  }

  // Round trip 2: 32 bytes at every surviving candidate.
  uint32_t len = 0;
  bool need_ext = false;
  if (cand) {
    const B32 prev32 = load_b32(c.data + prev_ix);
    const uint32_t m = common_prefix32(cur32, prev32);
    len = umin(m, max_length);
    need_ext = m == 32u && max_length > 32u;
  }
  // Rare: matches longer than 32 bytes are extended by the whole wave,
  // 512 bytes per step.
  uint64_t ext = wave_ballot(need_ext);
  while (ext) {
    const int j = dev_ctz64(ext);
    ext &= ext - 1;
    const uint32_t pj = wave_bcast(prev_ix, j);
    const uint32_t Pj = posA + (uint32_t)(j >> 5);
    const uint32_t ml = c.pos_end - Pj;
    uint32_t off = 32, L = ml;
    for (;;) {
      const uint32_t o = off + (uint32_t)lane * 8u;
      uint64_t x = 0;
      if (o < ml) x = ld64(c.data + Pj + o) ^ ld64(c.data + pj + o);
      const uint64_t mm = wave_ballot(x != 0);
      if (mm) {
        const int f = dev_ctz64(mm);
        const uint64_t xf = wave_bcast64(x, f);
        L = umin(off + (uint32_t)f * 8u + ((uint32_t)dev_ctz64(xf) >> 3), ml);
        break;
      }
      off += 512u;
      if (off >= ml) break;
    }
    if (lane == j) len = L;
  }

  // Scores (hash.h:123-138).
  uint32_t score = 0;
  if (is_bk) {
    score = 1920u + 135u * len - 30u * log2floor((P - prev_ix) | 1u);
  } else if (is_dc) {
    const uint32_t i = (uint32_t)(sub - 16);
    score = 135u * len + 1935u;
    if (i != 0) score -= 39u + ((0x1CA10u >> (i & 0xEu)) & 0xEu);
  }
  const uint64_t dc_ok = wave_ballot(cand && is_dc);
  const uint64_t bk_ok = wave_ballot(cand && is_bk && len >= 4u);

  // Half A: resolve, insert, dictionary.
  const uint32_t keyA = wave_bcast(kt.key, 0), tagA = wave_bcast(kt.tag, 0);
  const uint32_t tag2A = wave_bcast(kt.tag2, 0), numA = wave_bcast(num, 0);
  SearchResult ra = resolve_half(c, 0, posA, c.pos_end - posA, dc_ok, bk_ok, len, prev_ix, score);
  B.valid = false;
  if (doB) {
    const uint32_t keyB = wave_bcast(kt.key, 32);
    if (keyB != keyA) {
      B.valid = true;
      B.pos = posA + 1u;
      B.key = keyB;
      B.tag = wave_bcast(kt.tag, 32);
      B.tag2 = wave_bcast(kt.tag2, 32);
      B.num = wave_bcast(num, 32);
      B.sr = resolve_half(c, 1, posA + 1u, c.pos_end - posA - 1u, dc_ok, bk_ok, len, prev_ix, score);
    }
  }
  insert_searched(c, posA, keyA, tagA, tag2A, numA);
  const uint32_t dictionary_start = umin(posA + c.stream_offset, c.max_backward_limit);
  if (ra.score == K_MIN_SCORE) dict_search(c, posA, c.pos_end - posA, dictionary_start + c.gap, ra);
  if (c.cd) compound_lookup(c.cd, c.data + posA, posA & c.ring_mask, c.ring_mask, c.dc[0], c.dc[1], c.dc[2], c.dc[3], c.pos_end - posA,
                            dictionary_start, K_DIST_MAX_DISTANCE, ra);
  return ra;
}

// Completes the speculative half: dictionary probe + insertion.
DEV SearchResult finalize_b(ParseCtx& c, PendingB& B) {
  SearchResult r = B.sr;
  insert_searched(c, B.pos, B.key, B.tag, B.tag2, B.num);
  const uint32_t dictionary_start = umin(B.pos + c.stream_offset, c.max_backward_limit);
  if (r.score == K_MIN_SCORE) dict_search(c, B.pos, c.pos_end - B.pos, dictionary_start + c.gap, r);
  if (c.cd) compound_lookup(c.cd, c.data + B.pos, B.pos & c.ring_mask, c.ring_mask, c.dc[0], c.dc[1], c.dc[2], c.dc[3], c.pos_end - B.pos,
                            dictionary_start, K_DIST_MAX_DISTANCE, r);
  B.valid = false;
  return r;
}

// backward_references.c:87-109
DEV uint32_t compute_distance_code(uint32_t distance, uint32_t max_distance, const int32_t* dc) {
  if (distance <= max_distance) {
    const uint32_t dp3 = distance + 3u;
    const uint32_t o0 = dp3 - (uint32_t)dc[0];
    const uint32_t o1 = dp3 - (uint32_t)dc[1];
    if (distance == (uint32_t)dc[0]) return 0;
    if (distance == (uint32_t)dc[1]) return 1;
    if (o0 < 7) return (0x9750468u >> (4u * o0)) & 0xFu;
    if (o1 < 7) return (0xFDB1ACEu >> (4u * o1)) & 0xFu;
    if (distance == (uint32_t)dc[2]) return 2;
    if (distance == (uint32_t)dc[3]) return 3;
  }
  return distance + 16u - 1u;
}

struct BlockStats { uint64_t searches, pairs, b_used; };

// CreateBackwardReferences for one input block [position, position+num_bytes).
DEV void parse_block(ParseCtx& c, uint32_t position, uint32_t num_bytes,
                     uint32_t& last_insert_len, Command* commands, uint32_t& ncmds,
                     uint32_t& nlits, uint32_t spree_window, BlockStats& st) {
  const int lane = wave_lane();
  const uint32_t htl = (uint32_t)c.htl;
  const uint32_t pos_end = position + num_bytes;
  const uint32_t store_end = num_bytes >= htl ? pos_end - htl + 1u : position;
  uint32_t insert_length = last_insert_len;
  uint32_t apply_random_heuristics = position + spree_window;
  c.pos_end = pos_end;

  PendingB B;
  B.valid = false;
  B.pos = B.key = B.tag = B.tag2 = B.num = 0;
  B.sr.len = B.sr.distance = 0; B.sr.score = K_MIN_SCORE; B.sr.len_code_delta = 0;
  bool lazy = false;
  int delayed = 0;
  SearchResult sr;
  sr.len = 0; sr.distance = 0; sr.score = K_MIN_SCORE; sr.len_code_delta = 0;

  while (lazy || position + htl < pos_end) {
    const uint32_t need = lazy ? position + 1u : position;
    SearchResult cur;
    if (B.valid && B.pos == need) {
      cur = finalize_b(c, B);
      st.b_used++;
    } else {
      cur = search_pair(c, need, B);
      st.pairs++;
    }
    st.searches++;
    if (!lazy) {
      if (cur.score > K_MIN_SCORE) {
        sr = cur;
        delayed = 0;
        lazy = true;   // look one byte ahead before committing (:122-164)
        continue;
      }
      ++insert_length;
      ++position;
      if (position > apply_random_heuristics) {
        // Literal spree (:208-236): store sparsely, skip searches.
        B.valid = false;
        uint32_t step, span, margin;
        if (position > apply_random_heuristics + 4u * spree_window) {
          step = 4; span = 16; margin = umax(htl - 1u, 4u);
        } else {
          step = 2; span = 8; margin = umax(htl - 1u, 2u);
        }
        const uint32_t pos_jump = umin(position + span, pos_end - margin);
        if (position < pos_jump) {
          const uint32_t cnt = (pos_jump - position + step - 1u) / step;
          store_positions(c, position, cnt, step);
          position += cnt * step;
          insert_length += cnt * step;
        }
      }
      continue;
    }
    // lazy: `cur` is the search at position + 1
    if (cur.score >= sr.score + 175u) {
      ++position;
      ++insert_length;
      sr = cur;
      if (++delayed < 4 && position + htl < pos_end) continue;
    }
    lazy = false;
    B.valid = false;
    apply_random_heuristics = position + 2u * sr.len + spree_window;
    {
      const uint32_t dictionary_start = umin(position + c.stream_offset, c.max_backward_limit) + c.gap;
      const uint32_t distance_code = compute_distance_code(sr.distance, dictionary_start, c.dc);
      if (sr.distance <= dictionary_start && distance_code > 0) {
        c.dc[3] = c.dc[2]; c.dc[2] = c.dc[1]; c.dc[1] = c.dc[0]; c.dc[0] = (int32_t)sr.distance;
      }
      if (lane == 0) {
        commands[ncmds] = make_command(insert_length, sr.len, sr.len_code_delta, distance_code);
      }
      ++ncmds;
    }
    nlits += insert_length;
    insert_length = 0;
    {
      uint32_t range_start = position + 2u;
      const uint32_t range_end = umin(position + sr.len, store_end);
      if (sr.distance < (sr.len >> 2)) {
        range_start = umin(range_end, umax(range_start, position + sr.len - (sr.distance << 2)));
      }
      while (range_start < range_end) {
        const uint32_t cnt = umin(range_end - range_start, 64u);
        store_positions(c, range_start, cnt, 1);
        range_start += cnt;
      }
    }
    position += sr.len;
  }
  insert_length += pos_end - position;
  last_insert_len = insert_length;
}

// ExtendLastCommand, encode.c:905-971.
DEV void extend_last_command(const ParseCtx& c, Command* cmds, uint32_t ncmds,
                             uint32_t last_processed_pos, int lgwin, int32_t dc0,
                             uint32_t& bytes, uint32_t& pos) {
  const int lane = wave_lane();
  Command last = cmds[ncmds - 1];
  const uint32_t max_backward_distance = (1u << lgwin) - 16u;
  const uint32_t last_copy_len = last.copy_len & 0x1FFFFFFu;
  const uint32_t lpp = last_processed_pos - last_copy_len;
  const uint32_t max_distance = umin(lpp, max_backward_distance);
  const uint32_t cmd_dist = (uint32_t)dc0;
  uint32_t distance_code;
  {
    const uint32_t dcode = last.dist_prefix & 0x3FFu;
    if (dcode < 16) {
      distance_code = dcode;
    } else {
      const uint32_t nbits = last.dist_prefix >> 10;
      const uint32_t hcode = dcode - 16u;
      const uint32_t offset = ((2u + (hcode & 1u)) << nbits) - 4u;
      distance_code = offset + last.dist_extra + 16u;
    }
  }
  if (!(distance_code < 16u || distance_code - 15u == cmd_dist)) return;
  if (dc0 > 0 && cmd_dist <= max_distance) {
    // while (bytes && data[pos] == data[pos - dist]) extend; 64 bytes a step.
    for (;;) {
      const bool ok = (uint32_t)lane < bytes &&
          c.data[pos + (uint32_t)lane] == c.data[pos + (uint32_t)lane - cmd_dist];
      const uint64_t m = wave_ballot(ok);
      const uint32_t run = (m == ~0ull) ? 64u : (uint32_t)dev_ctz64(~m);
      last.copy_len += run;
      bytes -= run;
      pos += run;
      if (run < 64u || bytes == 0) break;
    }
  } else if (c.cd && cmd_dist > max_distance) {
    const uint32_t gained = compound_extend(c.cd, c.data + pos, bytes, cmd_dist, max_distance, last_copy_len);
    last.copy_len += gained;
    bytes -= gained;
    pos += gained;
  }
  last.cmd_prefix = (uint16_t)combine_length_codes(insert_length_code(last.insert_len),
      copy_length_code((uint32_t)((int)(last.copy_len & 0x1FFFFFFu) + (int)(last.copy_len >> 25))),
      (last.dist_prefix & 0x3FF) == 0);
  if (lane == 0) cmds[ncmds - 1] = last;
  wave_sync();
}

#endif  // BROTLI_AMD_CSRC_K_PARSE_H_
