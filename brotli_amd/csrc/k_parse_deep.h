// brotli_amd/csrc/k_parse_deep.h — K1 for the deep-bucket hashers: one encoder
// shard per wavefront, 32 ... 256 slots per bucket.
//
// Semantics: CreateBackwardReferences (c/enc/backward_references_inc.h:10-242)
// at qualities 6 ... 9: H68 / H58 with 32-slot buckets (quality 6,
// c/enc/hash_longest_match64_simd_inc.h, hash_longest_match_simd_inc.h), H6 / H5
// with 64 / 128 / 256 slots and 10 / 16 distance-cache probes (qualities 7 - 9,
// c/enc/hash_longest_match64_inc.h:105-277, hash_longest_match_inc.h;
// parameters c/enc/quality.h:172-223), plus the shared pieces of k_parse4.h
// (static dictionary, command construction, EncodeData glue).
//
// Design: same per-lane encoder state and driver as k_parse4.h, with the whole
// wave as the group.  Lane t owns slots t, t + 64, ... (E = slots / 64 entries
// per lane, one dwordx2 / dwordx4 / 2 x dwordx4 probe per lane); lanes
// 0 .. ndist-1 also probe the distance cache.  The reference's H5 / H6 have no
// tag array: every visible slot is compared (up to 256 window reads per
// search).  Here every entry carries the 16-bit fingerprint of its first four
// bytes, so only slots that can pass the reference's first-4-bytes test are
// read.  Entries are {u32 position, u16 fingerprint, u8 tag, u8 -}; the bucket
// counters live in a separate u16 array (entries never need clearing).
//
// Candidate resolve: the distance-cache phase of the reference accepts exactly
// the strict prefix maxima of the match length (the byte gate rejects anything
// not longer than the current best; a longer match always scores higher), so
// its result is the earliest candidate of maximum length; the bucket phase is
// the arg-max of the score in scan order (distances grow along the scan).  Both
// are evaluated with wave reductions; the configurations where a byte gate
// could decide (a bucket candidate that beats the cache winner without being
// longer; equal-length cache candidates that reach the block end) are detected
// and handled by the step-by-step emulation d_resolve_slow.
#ifndef BROTLI_AMD_CSRC_K_PARSE_DEEP_H_
#define BROTLI_AMD_CSRC_K_PARSE_DEEP_H_

#include "k_parse4.h"

#define D_DUP_SLOTS 4096u

// Maximum over the 64 lanes, in every lane: DPP row reductions + four readlanes (wave.h).  (Until round 6 a 16-lane
// reduction + two ds_bpermute round trips: seven of them were 15 % of a search — profiles/r06_c_q9_phases.txt.)
DEV uint32_t d_max(uint32_t v) { return wave_max_u32(v); }
// Value of lane `src` (wave-uniform) in every lane: one readlane.
DEV uint32_t d_from(uint32_t v, int src) { return wave_bcast(v, src); }

// hash.h:80-100: the sixteen entries of the prepared distance cache.
DEV uint32_t d_dc_entry(const QShard& g, int i) {
  if (i < 4) return q_dc_entry(g, i);
  const int32_t base = i < 10 ? g.dc[0] : g.dc[1];
  const int k = i < 10 ? i - 4 : i - 10;
  const int32_t mag = (k >> 1) + 1;
  return (uint32_t)((k & 1) ? base + mag : base - mag);
}

struct DeepGeom {
  uint32_t slots;       // 1 << block_bits
  uint32_t mask;        // slots - 1
  uint32_t rec_bytes;   // 8 * slots
  bool tagged;          // H68 / H58 (counter counts down from 0xFFFF) vs H5 / H6 (up from 0)
};
DEV DeepGeom deep_geom(const JobParams& J) {
  DeepGeom d;
  d.slots = 1u << J.block_bits;
  d.mask = d.slots - 1u;
  d.rec_bytes = 8u * d.slots;
  d.tagged = J.hasher_type >= 58;
  return d;
}
// byte offset of slot s inside a record: a lane's E entries are contiguous
template <int E>
DEV uint32_t deep_slot_offset(uint32_t s) { return ((s & 63u) * (uint32_t)E + (s >> 6)) * 8u; }
// (Round 6 tried two other shapes for qualities 7 - 9, measured and dropped — profiles/r06_i_*: positions and
//  fingerprints in arrays of their own, 749 against 663 ms per GiB at 512 KiB shards (eight narrow loads per lane instead
//  of two wide ones); and on top of that four consecutive positions searched at once by 16-lane groups, 840 ms: a bucket
//  of a common 5-gram holds up to 256 candidates that all pass the fingerprint, and 16 lanes walk them in 16 rounds
//  where the whole wave compares them in one.)
DEV uint8_t* d_rec(const DeepGeom& G, const QShard& g, uint32_t key) { return g.table + (size_t)key * G.rec_bytes; }
template <int E>
DEV void d_put_entry(const DeepGeom& G, const QShard& g, uint32_t key, uint32_t s, uint32_t pos, uint32_t tag2, uint32_t tag) {
  const uint64_t e = q_entry(pos, tag2, tag, 0);
  __builtin_memcpy(d_rec(G, g, key) + deep_slot_offset<E>(s), &e, 8);
}

// ---- ordered insertion of up to 64 positions ------------------------------------------
template <int E>
DEV void d_store64(const JobParams& J, const DeepGeom& G, const QShard& g, bool act, uint32_t pos,
                   uint8_t* lds_dup) {
  const int lane = wave_lane();
  KeyTag kt;
  kt.key = 0; kt.tag = 0; kt.tag2 = 0;
  if (act) kt = hash_pos(ld64(g.data + pos), J.hasher_type, J.bucket_bits);
  uint8_t* sb = lds_dup + (kt.key & (D_DUP_SLOTS - 1u));
  if (act) *sb = (uint8_t)lane;
  wave_sync();
  const bool dup = act && *sb != (uint8_t)lane;
  const bool any_dup = wave_any(dup);
  uint32_t num = 0;
  if (act) num = g.nums[kt.key];
  wave_sync();
  uint32_t below = 0, total = 1;
  if (any_dup) {
    uint64_t same = ~0ull;
    for (int b = 0; b < J.bucket_bits; ++b) {
      const bool bit = (kt.key >> b) & 1;
      const uint64_t m = wave_ballot(act && bit);
      same &= bit ? m : ~m;
    }
    same &= wave_ballot(act);
    below = (uint32_t)dev_popc64(same & ((1ull << lane) - 1ull));
    total = (uint32_t)dev_popc64(same);
  }
  if (act) {
    // the reference's serial loop leaves only the newest `slots` of a same-key run
    const uint32_t s = (G.tagged ? num - below : num + below) & G.mask;
    if (total - below <= G.slots) d_put_entry<E>(G, g, kt.key, s, pos, kt.tag2, kt.tag);
    if (below + 1 == total) g.nums[kt.key] = (uint16_t)(G.tagged ? num - total : num + total);
  }
  wave_sync();
}

template <int E>
DEV void d_drain_stores(const JobParams& J, const DeepGeom& G, QShard& g, uint8_t* lds_dup) {
  while (g.st_count != 0) {     // wave-uniform: one shard per wave
    const uint32_t n = umin(g.st_count, 64u);
    const bool act = (uint32_t)wave_lane() < n;
    d_store64<E>(J, G, g, act, g.st_first + (uint32_t)wave_lane() * g.st_stride, lds_dup);
    g.st_first += n * g.st_stride;
    g.st_count -= n;
  }
}

// Byte the reference reads at ring index x & mask for x <= pos_end: the data below pos_end; at
// pos_end the zero bytes behind the block on the first lap (encode.c:879-893) or, once the ring
// has been lapped, the byte of the lap before (c/enc/ringbuffer.h:103-159).
DEV uint32_t d_ring_byte(const JobParams& J, const QShard& g, uint32_t x) {
  if (x < g.pos_end) return g.data[x];
  if (g.pos_end <= J.ring_mask) return 0u;
  return g.data[x - (J.ring_mask + 1u)];
}

// ---- FindLongestMatch -----------------------------------------------------------------
// Step-by-step emulation (..64_inc.h:157-277 / ..64_simd_inc.h:170-302).
template <int E>
DEV QResult d_resolve_slow(const JobParams& J, const DeepGeom& G, const QShard& g, uint32_t P,
                           uint32_t max_length, uint32_t num, uint32_t visible, bool d_cand,
                           uint32_t d_len, uint32_t d_prev, uint32_t d_score, const bool* b_cand,
                           const uint32_t* b_len, const uint32_t* b_prev, const uint32_t* b_score) {
  QResult r;
  r.len = 0; r.distance = 0; r.score = K_MIN_SCORE; r.delta = 0;
  uint32_t best_len = 0;
  for (int i = 0; i < J.ndist; ++i) {
    const bool ok = d_from(d_cand ? 1u : 0u, i) != 0;
    const uint32_t len_i = d_from(d_len, i), prev_i = d_from(d_prev, i), score_i = d_from(d_score, i);
    if (!ok) continue;
    // candidates are not followed across the physical end of the ring (..64_inc.h:187-195)
    if ((P & J.ring_mask) + best_len > J.ring_mask) break;
    if ((prev_i & J.ring_mask) + best_len > J.ring_mask) continue;
    if (d_ring_byte(J, g, P + best_len) != d_ring_byte(J, g, prev_i + best_len)) continue;
    if (!(len_i >= 3 || (len_i == 2 && i < 2))) continue;
    if (!(r.score < score_i)) continue;
    best_len = len_i;
    r.len = len_i; r.distance = P - prev_i; r.score = score_i;
  }
  if (best_len < 3) best_len = 3;
  for (uint32_t j = 0; j < visible; ++j) {
    // j-th slot of the scan, newest first
    const uint32_t s = (G.tagged ? (num + 1u + j) : (num - 1u - j)) & G.mask;
    const int src = (int)(s & 63u);
    const uint32_t e = s >> 6;
    uint32_t ok = 0, len_j = 0, prev_j = 0, score_j = 0;
#pragma unroll
    for (int k = 0; k < E; ++k) {
      const uint32_t o = d_from(b_cand[k] ? 1u : 0u, src), l = d_from(b_len[k], src);
      const uint32_t p = d_from(b_prev[k], src), sc = d_from(b_score[k], src);
      if ((uint32_t)k == e) { ok = o; len_j = l; prev_j = p; score_j = sc; }
    }
    if (!ok) continue;
    if ((P & J.ring_mask) + best_len > J.ring_mask) break;                    // (:243-249)
    if ((prev_j & J.ring_mask) + best_len > J.ring_mask) continue;
    bool pass = true;
    for (uint32_t k = best_len - 3; k <= best_len; ++k) {
      if (d_ring_byte(J, g, P + k) != d_ring_byte(J, g, prev_j + k)) { pass = false; break; }
    }
    if (!pass) continue;
    if (len_j < 4) continue;
    if (!(r.score < score_j)) continue;
    best_len = len_j;
    r.len = len_j; r.distance = P - prev_j; r.score = score_j;
  }
  (void)max_length;
  return r;
}

// The next search's bytes (and, -DDEEP_PREFETCH=2, its bucket counter and record), requested at the END of this search —
// behind this search's own loads, which come back in order, and behind the insertion of P, so that the record is
// current — and used when the next search is at P + 1: four of five are (no match: the next literal; a match: the lazy
// probe, backward_references_inc.h:122-160).  A search is three dependent round trips — bytes at P -> key -> counter +
// record -> candidate strings.  A StoreRange / the spree's stores in between drop the prefetch.
// (Round 6 first requested everything at the START of the search: no gain — the wait moved from the record to the
//  candidate strings, profiles/r06_d_q9_phases_prefetch.txt.)
#ifndef DEEP_PREFETCH
#define DEEP_PREFETCH 1
#endif

#define D_PF_NONE 0xFFFFFFFFu
template <int E>
struct DeepPf { uint32_t pos, key, num; uint64_t ent[E]; B32 cur; };

template <int E>
DEV void d_fetch_record(const DeepGeom& G, const QShard& g, uint32_t key, uint32_t& num, uint64_t (&ent)[E]) {
  const int lane = wave_lane();
  const uint8_t* mine = d_rec(G, g, key) + (uint32_t)lane * (8u * (uint32_t)E);
  const bool in_rec = (uint32_t)lane < G.slots;        // quality 6: 32 slots, upper lanes idle
  num = g.nums[key];
#pragma unroll
  for (int k = 0; k < E; ++k) ent[k] = in_rec ? ld64(mine + 8 * k) : 0ull;
}

template <int E>
DEV QResult d_search(const JobParams& J, const DeepGeom& G, const DeviceTables* T, QShard& g, uint32_t P, DeepPf<E>& pf) {
  const int lane = wave_lane();
  const int ndist = J.ndist;
  uint64_t qt = QP_NOW();
  const uint32_t max_length = g.pos_end - P;
  const uint32_t max_backward = umin(P, J.max_backward_limit);
  const bool ahead = DEEP_PREFETCH >= 1 && pf.pos == P;  // (wave-uniform)
  B32 cur32;
  if (ahead) cur32 = pf.cur; else cur32 = load_b32(g.data + P);
  // distance-cache probes: independent of the hash table, requested first
  const uint32_t backward = d_dc_entry(g, lane & 15);
  const bool d_cand = lane < ndist && (int32_t)backward > 0 && backward <= max_backward;
  const uint32_t d_prev = P - backward;
  const B32 pd = load_b32(g.data + (d_cand ? d_prev : P));
  const KeyTag kt = hash_pos(cur32.q[0], J.hasher_type, J.bucket_bits);
  uint32_t num;
  uint64_t ent[E];
  if (DEEP_PREFETCH >= 2 && ahead) {
    num = pf.num;
#pragma unroll
    for (int k = 0; k < E; ++k) ent[k] = pf.ent[k];
  } else {
    d_fetch_record<E>(G, g, kt.key, num, ent);
  }
#if defined(Q_PROFILE)
  if (ent[0] == 0x123456789ull) g.status |= 0x40000000u;     // (profiling: waits for the record before the lap)
#endif
  QP_ADD(g, 0, qt);                                          // bytes at P -> key -> counter + record
  // visible slots and scan order (newest first)
  uint32_t visible;
  if (G.tagged) {
    const uint32_t n = (65535u - num) & 0xFFFFu;
    visible = n < G.slots ? n : G.slots;
  } else {
    visible = num < G.slots ? num : G.slots;          // num > slots ? slots : num (..64_inc.h:230-236)
  }
  bool b_cand[E];
  uint32_t b_prev[E], b_len[E], b_score[E], b_logical[E];
  B32 pb[E];
#pragma unroll
  for (int k = 0; k < E; ++k) {
    const uint32_t s = (uint32_t)lane + 64u * (uint32_t)k;
    const uint32_t slot = (uint32_t)ent[k];
    const uint32_t tag2 = (uint32_t)(ent[k] >> 32) & 0xFFFFu;
    const uint32_t tag = (uint32_t)(ent[k] >> 48) & 0xFFu;
    b_logical[k] = (G.tagged ? (s - (num + 1u)) : (num - 1u - s)) & G.mask;
    b_prev[k] = slot;
    b_cand[k] = s < G.slots && b_logical[k] < visible && tag2 == kt.tag2 && (!G.tagged || tag == kt.tag) &&
                (P - slot) <= max_backward;
    pb[k].q[0] = pb[k].q[1] = pb[k].q[2] = pb[k].q[3] = 0;
    if (b_cand[k]) pb[k] = load_b32(g.data + slot);
  }
  uint32_t d_len = 0;
  bool ext_any = false;
  {
    const uint32_t md = common_prefix32(cur32, pd);
    if (d_cand) { d_len = umin(md, max_length); if (md == 32u && max_length > 32u) ext_any = true; }
  }
#pragma unroll
  for (int k = 0; k < E; ++k) {
    const uint32_t mb = common_prefix32(cur32, pb[k]);
    b_len[k] = 0;
    if (b_cand[k]) { b_len[k] = umin(mb, max_length); if (mb == 32u && max_length > 32u) ext_any = true; }
  }
  if (wave_any(ext_any)) {
    if (d_cand && d_len == 32u && max_length > 32u) d_len = q_extend(g.data, P, d_prev, max_length);
#pragma unroll
    for (int k = 0; k < E; ++k)
      if (b_cand[k] && b_len[k] == 32u && max_length > 32u) b_len[k] = q_extend(g.data, P, b_prev[k], max_length);
  }
  QP_ADD(g, 1, qt);                                          // candidate strings, lengths, extensions
  // scores (hash.h:123-138)
  uint32_t d_score = 135u * d_len + 1935u;
  if (lane != 0) d_score -= 39u + ((0x1CA10u >> ((uint32_t)lane & 0xEu)) & 0xEu);
  const bool d_ok = d_cand && (d_len >= 3u || (d_len == 2u && lane < 2));
  // distance cache: earliest candidate of maximum length
  const uint32_t d_key = d_ok ? (d_len << 5) | (31u - (uint32_t)lane) : 0u;
  const uint32_t d_best = d_max(d_key);
  const bool d_win = d_key != 0 && d_key == d_best;
  const uint32_t dc_len = d_best >> 5;
  // (the winner's score from its length and lane: no second reduction)
  const uint32_t d_wl = 31u - (d_best & 31u);
  uint32_t dc_score = K_MIN_SCORE;
  if (d_best) { dc_score = 135u * dc_len + 1935u; if (d_wl != 0) dc_score -= 39u + ((0x1CA10u >> (d_wl & 0xEu)) & 0xEu); }
  const uint32_t dc_len3 = dc_len < 3u ? 3u : dc_len;
  // block-end edge: another cache candidate as long as the winner and reaching
  // the end of the block can be let through the gate by the byte past the block
  bool unsure = d_ok && !d_win && d_len == dc_len && d_len == max_length;
  // buckets: arg-max of the score, scan order as the tie break
  uint32_t my_key = 0, my_len = 0, my_dist = 0;
#pragma unroll
  for (int k = 0; k < E; ++k) {
    b_score[k] = 1920u + 135u * b_len[k] - 30u * log2floor((P - b_prev[k]) | 1u);
    const bool ok = b_cand[k] && b_len[k] >= 4u;
    if (ok && b_score[k] > dc_score && b_len[k] <= dc_len3) unsure = true;
    const uint32_t key = ok ? (b_score[k] << 9) | (511u - b_logical[k]) : 0u;
    if (key > my_key) { my_key = key; my_len = b_len[k]; my_dist = P - b_prev[k]; }
  }
  if (g.pos_end > J.ring_mask) {      // (a stream as long as its ring buffer: nothing below can trigger before that)
    // a candidate that sits within a match length of the physical end of the ring may be skipped
    // by the reference (it does not follow matches across that end): rare, resolved step by step
    uint32_t longest = d_cand ? d_len : 0u;
#pragma unroll
    for (int k = 0; k < E; ++k) if (b_cand[k]) longest = umax(longest, b_len[k]);
    longest = umax(d_max(longest), 3u);
    // (after a flush the input blocks are no longer aligned to the ring: P itself can sit within a
    // match length of its physical end, where the reference stops looking, :187, 243)
    if ((P & J.ring_mask) + longest > J.ring_mask) unsure = true;
    if (d_cand && (d_prev & J.ring_mask) + longest > J.ring_mask) unsure = true;
#pragma unroll
    for (int k = 0; k < E; ++k) if (b_cand[k] && (b_prev[k] & J.ring_mask) + longest > J.ring_mask) unsure = true;
  }
  const bool force_slow = (J.flags & JOB_FLAG_FORCE_SLOW) != 0;
  const bool slow = wave_any(unsure) || force_slow;
  const uint32_t best = d_max(my_key);
  const uint32_t b_best_score = best >> 9;
  QResult r;
  if (best != 0 && b_best_score > dc_score && b_best_score > K_MIN_SCORE) {
    // the winner's lane from its place in the scan (the key's low bits): two readlanes instead of two reductions
    const uint32_t logical = 511u - (best & 511u);
    const int wl = (int)(((G.tagged ? (num + 1u + logical) : (num - 1u - logical)) & G.mask) & 63u);
    r.len = d_from(my_len, wl);
    r.distance = d_from(my_dist, wl);
    r.score = b_best_score;
  } else if (d_best != 0) {
    r.len = dc_len;
    r.distance = d_from(backward, (int)d_wl);
    r.score = dc_score;
  } else {
    r.len = 0; r.distance = 0; r.score = K_MIN_SCORE;
  }
  r.delta = 0;
  QP_ADD(g, 2, qt);                                          // scores, reductions
  if (slow) {
    r = d_resolve_slow<E>(J, G, g, P, max_length, num, visible, d_cand, d_len, d_prev, d_score,
                          b_cand, b_len, b_prev, b_score);
  }
  QP_ADD(g, 3, qt);                                          // step-by-step resolve
  // insert P
  {
    const uint32_t ts = num & G.mask;
    if ((uint32_t)lane == (ts & 63u)) {
      d_put_entry<E>(G, g, kt.key, ts, P, kt.tag2, kt.tag);
      g.nums[kt.key] = (uint16_t)(G.tagged ? num - 1u : num + 1u);
    }
  }
  // the search after this one, most likely: P + 1
  pf.pos = D_PF_NONE;
  if (DEEP_PREFETCH >= 1 && P + 1u < g.pos_end) {
    pf.pos = P + 1u;
    pf.cur = load_b32(g.data + P + 1u);
    if (DEEP_PREFETCH >= 2) {
      pf.key = hash_pos((cur32.q[0] >> 8) | (cur32.q[1] << 56), J.hasher_type, J.bucket_bits).key;
      d_fetch_record<E>(G, g, pf.key, pf.num, pf.ent);
    }
  }
  wave_sync();
  QP_ADD(g, 4, qt);                                          // insert
  q_dict_search(J, T, g, r.score == K_MIN_SCORE, P, max_length, r);
  q_compound_lookup(J, g, P, max_length, r);
  QP_ADD(g, 5, qt);                                          // dictionaries
  return r;
}

// ExtendLastCommand + the CreateBackwardReferences prologue (one shard per wave).
template <int E>
DEV void d_setup_block(const JobParams& J, const DeepGeom& G, QShard& g, uint8_t* lds_dup) {
  const int lane = wave_lane();
  const uint32_t htl = hasher_htl(J.hasher_type);
  if (g.blk_flags & QBLK_STITCH) {
    g.st_first = g.blk_pos - 3u;
    g.st_count = 3;
    g.st_stride = 1;
  }
  d_drain_stores<E>(J, G, g, lds_dup);
  uint32_t bytes = g.blk_bytes, pos = g.blk_pos;
  if (g.blk_flags & QBLK_EXTEND) {
    Command last = g.cmds[g.r.ncmds - 1];
    const uint32_t last_copy_len = last.copy_len & 0x1FFFFFFu;
    const uint32_t lpp = g.r.last_processed_pos - last_copy_len;
    const uint32_t max_distance = umin(lpp, J.max_backward_limit);
    const uint32_t cmd_dist = (uint32_t)g.dc[0];
    uint32_t distance_code;
    const uint32_t dcode = last.dist_prefix & 0x3FFu;
    if (dcode < 16) {
      distance_code = dcode;
    } else {
      const uint32_t nbits = last.dist_prefix >> 10;
      const uint32_t hcode = dcode - 16u;
      const uint32_t offset = ((2u + (hcode & 1u)) << nbits) - 4u;
      distance_code = offset + last.dist_extra + 16u;
    }
    if (distance_code < 16u || distance_code - 15u == cmd_dist) {
      if (g.dc[0] > 0 && cmd_dist <= max_distance) {
        for (;;) {
          const bool ok = (uint32_t)lane < bytes &&
              g.data[pos + (uint32_t)lane] == g.data[pos + (uint32_t)lane - cmd_dist];
          const uint64_t m = wave_ballot(ok);
          const uint32_t run = (m == ~0ull) ? 64u : (uint32_t)dev_ctz64(~m);
          last.copy_len += run;
          bytes -= run;
          pos += run;
          if (run < 64u || bytes == 0) break;
        }
      } else {
        q_compound_extend(g, last, cmd_dist, max_distance, bytes, pos);
      }
      last.cmd_prefix = (uint16_t)combine_length_codes(insert_length_code(last.insert_len),
          copy_length_code((uint32_t)((int)(last.copy_len & 0x1FFFFFFu) + (int)(last.copy_len >> 25))),
          (last.dist_prefix & 0x3FF) == 0);
      if (lane == 0) g.cmds[g.r.ncmds - 1] = last;
    }
  }
  wave_sync();
  g.position = pos;
  g.pos_end = pos + bytes;
  g.store_end = bytes >= htl ? g.pos_end - htl + 1u : pos;
  g.insert_length = g.r.last_insert_len;
  g.apply_random_heuristics = pos + J.spree_window;
  g.state = Q_SEARCH;
}

// ---- the kernel body: one shard per wave ------------------------------------------------
template <int E>
DEV void parse_deep_round(const JobParams& J, const ShardDesc& D, ShardState* S, const DeviceTables* T,
                          const uint8_t* input, uint8_t* ws, uint8_t* lds_dup, const CompoundDict* cd = nullptr) {
  const int lane = wave_lane();
  const bool writer = lane == 0;
  const uint32_t htl = hasher_htl(J.hasher_type);
  const DeepGeom G = deep_geom(J);
  if (S->done || S->mb_valid || S->error) return;

  QShard g;
  g.data = input + D.in_off;
  g.table = ws + D.table_off;
  g.nums = (uint16_t*)(ws + D.num_off);
  g.cmds = (Command*)(ws + D.cmds_off);
  g.descs = &D;
  g.wsb = ws;
  g.shard = 0;
  g.stream_offset = D.stream_offset;
  g.cd = cd;
  g.gap = cd ? cd->total_size : 0u;
  regs_load(g.r, S);
  for (int i = 0; i < 4; ++i) g.dc[i] = S->dist_cache[i];
  g.dict_lookups = S->dict_lookups;
  g.dict_matches = S->dict_matches;
  g.blk_flags = g.blk_bytes = g.blk_pos = 0;
  g.position = g.pos_end = g.store_end = g.insert_length = g.apply_random_heuristics = 0;
  g.sr_len = g.sr_dist = 0; g.sr_score = K_MIN_SCORE; g.sr_delta = 0; g.delayed = 0;
  g.st_first = g.st_count = 0; g.st_stride = 1;
  g.st_x = 0; g.st_x_valid = 0;
  g.n32.q[0] = g.n32.q[1] = g.n32.q[2] = g.n32.q[3] = 0;
  g.n32_pos = 0xFFFFFFFFu;
  g.status = 0;
  g.stat_searches = 0;
  g.pf_val = g.pf_acc = 0;
  g.role = 0;
  g.state = Q_PRE;
  for (int i = 0; i < 12; ++i) g.prof[i] = 0;
  uint64_t dt = QP_NOW();
  DeepPf<E> pf;
  pf.pos = D_PF_NONE;

  while (g.state != Q_DONE) {       // all state is wave-uniform here
    QP_ADD(g, 8, dt);                                        // driver
    if (g.state == Q_PRE) q_driver_pre(J, g);
    if (g.state == Q_SETUP) { d_setup_block<E>(J, G, g, lds_dup); pf.pos = D_PF_NONE; }
    if (g.state == Q_SEARCH && !(g.position + htl < g.pos_end)) {
      g.insert_length += g.pos_end - g.position;
      g.r.last_insert_len = g.insert_length;
      g.state = Q_POST;
    }
    if (g.state == Q_SEARCH || g.state == Q_LAZY) {
      const uint32_t P = g.position + (g.state == Q_LAZY ? 1u : 0u);
      dt = QP_NOW();
      const QResult cur = d_search<E>(J, G, T, g, P, pf);
      dt = QP_NOW();
      g.stat_searches++;
      bool commit = false;
      if (g.state == Q_SEARCH) {
        if (cur.score > K_MIN_SCORE) {
          g.sr_len = cur.len; g.sr_dist = cur.distance; g.sr_score = cur.score; g.sr_delta = cur.delta;
          g.delayed = 0;
          g.state = Q_LAZY;
        } else {
          ++g.insert_length;
          ++g.position;
          if (g.position > g.apply_random_heuristics) {
            uint32_t step, span, margin;
            if (g.position > g.apply_random_heuristics + 4u * J.spree_window) {
              step = 4; span = 16; margin = umax(htl - 1u, 4u);
            } else {
              step = 2; span = 8; margin = umax(htl - 1u, 2u);
            }
            const uint32_t pos_jump = umin(g.position + span, g.pos_end - margin);
            if (g.position < pos_jump) {
              const uint32_t cnt = (pos_jump - g.position + step - 1u) / step;
              g.st_first = g.position;
              g.st_count = cnt;
              g.st_stride = step;
              g.position += cnt * step;
              g.insert_length += cnt * step;
            }
          }
        }
      } else {
        commit = true;
        if (cur.score >= g.sr_score + 175u) {
          ++g.position;
          ++g.insert_length;
          g.sr_len = cur.len; g.sr_dist = cur.distance; g.sr_score = cur.score; g.sr_delta = cur.delta;
          if (++g.delayed < 4 && g.position + htl < g.pos_end) commit = false;
        }
      }
      if (commit) {
        g.state = Q_SEARCH;
        uint32_t range_start = g.position + 2u;
        const uint32_t range_end = umin(g.position + g.sr_len, g.store_end);
        if (g.sr_dist < (g.sr_len >> 2)) {
          range_start = umin(range_end, umax(range_start, g.position + g.sr_len - (g.sr_dist << 2)));
        }
        if (range_start < range_end) {
          g.st_first = range_start;
          g.st_count = range_end - range_start;
          g.st_stride = 1;
        }
        g.apply_random_heuristics = g.position + 2u * g.sr_len + J.spree_window;
        const uint32_t dictionary_start = umin(g.position + g.stream_offset, J.max_backward_limit) + g.gap;
        const uint32_t distance_code = compute_distance_code(g.sr_dist, dictionary_start, g.dc);
        if (g.sr_dist <= dictionary_start && distance_code > 0) {
          g.dc[3] = g.dc[2]; g.dc[2] = g.dc[1]; g.dc[1] = g.dc[0]; g.dc[0] = (int32_t)g.sr_dist;
        }
        if (lane == 0) g.cmds[g.r.ncmds] = make_command(g.insert_length, g.sr_len, g.sr_delta, distance_code);
        ++g.r.ncmds;
        g.r.nlits += g.insert_length;
        g.insert_length = 0;
        g.position += g.sr_len;
      }
      QP_ADD(g, 6, dt);                                      // decide + commit
      if (DEEP_PREFETCH >= 2 && g.st_count != 0) pf.pos = D_PF_NONE;   // (a StoreRange / the spree's stores: the prefetched record may be theirs)
      if (DEEP_PREFETCH == 1 && pf.pos != g.position && g.state == Q_SEARCH && g.position < g.pos_end) {
        // the search goes on somewhere else (behind a copy, behind the spree's skip): its bytes travel during the stores
        pf.pos = g.position;
        pf.cur = load_b32(g.data + g.position);
      }
      d_drain_stores<E>(J, G, g, lds_dup);
      QP_ADD(g, 7, dt);                                      // StoreRange
    }
    if (g.state == Q_POST) q_driver_post(J, g, writer);
  }

  wave_sync();
  if (writer) {
    regs_save(g.r, S);
    for (int i = 0; i < 4; ++i) S->dist_cache[i] = g.dc[i];
    S->dict_lookups = g.dict_lookups;
    S->dict_matches = g.dict_matches;
    S->done = (g.status & QST_DONE) ? 1u : 0u;
    S->mb_valid = (g.status & QST_HAVE_MB) ? 1u : 0u;
    if (g.status & QST_ERROR) S->error = 1;
    if (g.status & QST_HAVE_MB) {
      S->mb_start = g.r.last_flush_pos;
      S->mb_bytes = g.r.input_pos - g.r.last_flush_pos;
      S->mb_is_last = (g.blk_flags & QBLK_LAST) ? 1u : 0u;
      S->mb_force_flush = (g.blk_flags & QBLK_FLUSH) ? ((g.blk_flags & QBLK_NOSEAL) ? 2u : 1u) : 0u;
      S->mb_raw = 0;
    }
    S->stat_searches += g.stat_searches;
    S->stat_pairs += g.stat_searches;
#if defined(Q_PROFILE)
    for (int i = 0; i < 12; ++i) S->prof[i] += g.prof[i];
#endif
  }
  wave_sync();
}

#endif  // BROTLI_AMD_CSRC_K_PARSE_DEEP_H_
