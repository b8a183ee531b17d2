// brotli_amd/csrc/k_index.h — the data-parallel half of the quality-5 LZ77 parse
// (JOB_FLAG_INDEXED): a position index that answers "what would the 16-slot
// tagged bucket of this position hold, and which of its entries is the best
// match" for EVERY position of a shard at once, without a hash table.
//
// What the reference computes (c/enc/hash_longest_match64_simd_inc.h:114-302,
// hash_longest_match_simd_inc.h for the 4-byte hasher): Store() puts positions
// into the 16-slot ring of their bucket key in increasing position order, and
// FindLongestMatch(P) looks at the (up to) 16 most recently stored positions of
// P's key whose 8-bit tag equals P's.  Which positions get stored depends on
// the parse, but only through a few, locally known exceptions (the position
// after a chain of lazy matches, the clipped range of a run-length copy, the
// literal spree, block tails — c/enc/backward_references_inc.h:122-236): on
// ordinary data almost every position is stored.  So:
//
//   * F = the positions that CAN be stored (block structure only): x + HTL <=
//     block end, or one of the three positions StitchToPreviousBlock adds;
//   * the index is built for F: positions sorted by (key, position) — a
//     two-level stable counting sort, 256 .. 1024 buckets by the top key bits
//     (ix_count / ix_scan / ix_scatter: 4 bytes per position through HBM) and the
//     remaining key bits inside a bucket (ix_bucket, which fetches the 16 bytes at
//     each position from the shard's input — L2 resident — once);
//   * ix_bucket then evaluates, for every searchable position P, the window of
//     the <= 16 predecessors of P in its key run exactly as FindLongestMatch's
//     bucket loop would (tag filter, first four bytes, match length, score, order
//     of visit as the tie break) and stores the arg-max in res[P];
//   * the serial chain (k_chain.h) walks CreateBackwardReferences with res[P]
//     plus the distance-cache candidates it computes itself.  It keeps a bitmap
//     of the positions of F it did NOT store and sets a taint bit in res[] of the
//     (at most 16) positions that follow such a position in its key run; a search
//     that is tainted, or whose result the index marked as not decidable in
//     isolation, is redone exactly from the sorted array (`srt`), so the output
//     never depends on how often the shortcut applied.
//
// Everything here is one 64-lane wave per workgroup, no inter-workgroup
// communication inside a kernel; kernels hand over through HBM.
#ifndef BROTLI_AMD_CSRC_K_INDEX_H_
#define BROTLI_AMD_CSRC_K_INDEX_H_

#include "device_common.h"
#include "k_index_layout.h"

// ---- block structure of a completely fed shard ------------------------------------
// (BrotliEncoderCompressStream re-blocks the input to 1 << lgblock, encode.c:1666-1681;
// a shard with a stream offset first emits its two "flint" bytes as a block of their own,
// encode.c:1686-1694.)
// Positions in ix_block_end / ix_storable / ix_searchable are positions of the shard (of the stream, JOB_FLAG_STREAMT).
// An index chunk of a stream (ShardDesc::ix_glen != 0) holds the stream positions [base, base + len) under the local
// positions [0, len): the index kernels work in local positions (24 bits in an entry) and ask these functions about
// local + base; [0, own) is the chunk's look-back (candidates for the searches from `own` on, not searched itself —
// the chunk before this one searches them).
// `older`: the chunk's look-back is shorter than the window (chunk_log2 < lgwin: a stream at lgwin 24, whose chunks are
// half a window so that a chunk with its look-back stays within 24-bit positions) — a search with fewer than 16
// same-key entries before it in the chunk may have candidates below the chunk's base: the chain's to do (k_chain.h,
// c_search_exact goes on in the chunk before).
struct IxGeom { uint32_t n, first, lgblock, htl, len, base, own, ownc, maxdist, ring_mask; bool stream, older; };
DEV IxGeom ix_geom(const JobParams& J, const ShardDesc& D) {
  IxGeom g;
  g.stream = D.ix_glen != 0u;
  g.older = g.stream && J.chunk_log2 < (uint32_t)J.lgwin && D.ix_base != 0u;
  g.n = g.stream ? D.ix_glen : D.len;
  g.len = D.len;
  g.base = D.ix_base;
  g.own = D.ix_own;
  g.ownc = D.ix_ownc;
  g.maxdist = g.stream ? J.max_backward_limit : 0xFFFFFFFFu;
  g.ring_mask = J.ring_mask;
  g.first = D.stream_offset != 0 ? 2u : 0u;
  g.lgblock = (uint32_t)J.lgblock;
  g.htl = hasher_htl(J.hasher_type);
  return g;
}
// The three position bitmaps of a shard (bit x - first) — for a chunk of a stream the chunk's window on the stream's.
struct IxBitmaps { uint8_t* skip; uint8_t* prev; uint8_t* ev; };
DEV IxBitmaps ix_bitmaps(const JobParams& J, const ShardDesc& D, uint8_t* ws, const IxLayout& L) {
  IxBitmaps b;
  if (D.ix_glen != 0u) {
    uint8_t* G = ws + J.sbm_off + D.ix_base / 8u;
    b.skip = G; b.prev = G + J.sbm_stride; b.ev = G + 2u * J.sbm_stride;
  } else {
    uint8_t* base = ws + D.ix_off;
    b.skip = base + L.skip; b.prev = base + L.skip_prev; b.ev = base + L.ev;
  }
  return b;
}
// End of the input block that contains x.
DEV uint32_t ix_block_end(const IxGeom& g, uint32_t x) {
  if (x < g.first) return umin(g.first, g.n);
  const uint32_t e = g.first + ((((x - g.first) >> g.lgblock) + 1u) << g.lgblock);
  return umin(e, g.n);
}
// Can position x ever be stored?  StoreRange stops at pos_end - HTL + 1
// (backward_references_inc.h:41, 190-199), searches need x + HTL < pos_end (:44),
// StitchToPreviousBlock adds the last three positions of a block when the next one
// has >= HTL - 1 bytes and starts at >= 3 (..64_simd_inc.h:139-151).
DEV bool ix_storable(const IxGeom& g, uint32_t x) {
  if (x < g.first || x >= g.n) return false;
  const uint32_t e = ix_block_end(g, x);
  if (x + g.htl <= e) return true;
  return x + 3u >= e && e >= 3u && g.n - e >= g.htl - 1u;
}
// Positions FindLongestMatch can be called for: the loop guard is x + HTL < pos_end (:44), the
// lazy probe of x + 1 inside the loop has no guard of its own (:127-133).
DEV bool ix_searchable(const IxGeom& g, uint32_t x) {
  if (x < g.first || x >= g.n) return false;
  return x + g.htl <= ix_block_end(g, x);
}

// The 16 bytes at the entry's position, its full bucket key (w1), and its tag in place of the low key bits the
// entry travelled with (w0 = position | tag << 24 from here on: what the window search compares and srt[] keeps).
DEV void ix_fetch(const JobParams& J, const uint8_t* data, IxEntry& e) {
  uint64_t b[2];
  __builtin_memcpy(b, data + (e.w0 & 0xFFFFFFu), 16);
  e.d = b[0]; e.d2 = b[1];
  const KeyTag kt = hash_pos(e.d, J.hasher_type, J.bucket_bits);
  e.w1 = kt.key;
  e.w0 = (e.w0 & 0xFFFFFFu) | (kt.tag << 24);
}

// Lanes of the wave whose `v` (nbits wide) equals this lane's, among the lanes with `act`.
DEV uint64_t ix_match_any(bool act, uint32_t v, int nbits) {
  uint64_t same = wave_ballot(act);
  for (int b = 0; b < nbits; ++b) {
    const bool bit = (v >> b) & 1u;
    const uint64_t m = wave_ballot(act && bit);
    same &= bit ? m : ~m;
  }
  return act ? same : 0ull;
}

// ---- level 1: buckets by the top key bits ----------------------------------------------
// Slice `w` of the shard: positions [w * per, (w + 1) * per), per a multiple of 64.
DEV uint32_t ix_slice_len(uint32_t n, uint32_t slices) {
  return (((n + slices - 1u) / slices) + 63u) & ~63u;
}

// grid = nshards * slices.  Counts the storable positions of the slice per bucket;
// cnt[bucket * slices + w].  Also clears the slice's part of the unstored-position bitmap.
DEV void ix_count(const JobParams& J, const ShardDesc& D, const uint8_t* input, uint8_t* ws,
                  uint32_t w, uint32_t* lds_cnt, bool first_wave) {
  const int lane = wave_lane();
  const IxGeom g = ix_geom(J, D);
  IxLayout L;
  ix_layout(g.len, J.ix_slices, J.ix_nb_log2, &L);
  uint8_t* base = ws + D.ix_off;
  const uint8_t* data = input + D.in_off;
  const uint32_t per = ix_slice_len(g.len, J.ix_slices);
  const uint32_t lo = w * per, hi = umin(lo + per, g.len);
  const uint32_t nbk = 1u << J.ix_nb_log2;
  for (uint32_t b = (uint32_t)lane; b < nbk; b += 64u) lds_cnt[b] = 0;
  wave_sync();
  const int shift = J.bucket_bits - (int)J.ix_nb_log2;
  for (uint32_t x0 = lo; x0 < hi; x0 += 256u) {
    uint64_t v[4];
#pragma unroll
    for (uint32_t u = 0; u < 4u; ++u) { const uint32_t x = x0 + u * 64u + (uint32_t)lane; v[u] = x < hi ? ld64(data + x) : 0ull; }
#pragma unroll
    for (uint32_t u = 0; u < 4u; ++u) {
      const uint32_t x = x0 + u * 64u + (uint32_t)lane;
      if (x < hi && ix_storable(g, x + g.base)) {
        const KeyTag kt = hash_pos(v[u], J.hasher_type, J.bucket_bits);
        lds_atomic_add(&lds_cnt[kt.key >> shift], 1u);
      }
    }
  }
  wave_sync();
  uint32_t* cnt = (uint32_t*)(base + L.cnt);
  for (uint32_t b = (uint32_t)lane; b < nbk; b += 64u) cnt[b * J.ix_slices + w] = lds_cnt[b];
  // bitmap bytes of this slice (per is a multiple of 64 positions = 8 bytes)
  const IxBitmaps bm = ix_bitmaps(J, D, ws, L);
  uint32_t* skip = (uint32_t*)bm.skip;
  const uint32_t w_lo = lo / 32u, w_hi = (umin(lo + per, g.len + 128u) + 31u) / 32u;
  for (uint32_t i = w_lo + (uint32_t)lane; i < w_hi; i += 64u) skip[i] = 0;
#if defined(IX_RES4X)
  if (IX_RES4X == 3) { uint32_t* r4 = (uint32_t*)(base + L.res); for (uint32_t i = lo + (uint32_t)lane; i < hi; i += 64u) r4[i] = 0; }
#endif
  if (g.stream) {
    // the chunk's key table (k_tile.h): every slice clears its share
    const uint32_t cj = g.ownc == 0u ? 0u : (g.base >> J.chunk_log2) + 1u;
    uint32_t* kt = (uint32_t*)(ws + J.skt_off + (uint64_t)cj * skt_chunk_bytes((uint32_t)J.bucket_bits));
    const uint32_t words = SKT_WORDS << J.bucket_bits, share = (words + J.ix_slices - 1u) / J.ix_slices;
    for (uint32_t i = w * share + (uint32_t)lane; i < umin((w + 1u) * share, words); i += 64u) kt[i] = 0;
  }
  if (J.flags & JOB_FLAG_TILED) {
    uint32_t* prev = (uint32_t*)bm.prev;
    uint32_t* ev = (uint32_t*)bm.ev;
    for (uint32_t i = w_lo + (uint32_t)lane; i < w_hi; i += 64u) { prev[i] = 0; ev[i] = 0; }
  }
  // the block lists of the buckets too big for LDS (ix_bucket fills them, ix_big empties them): the job's first wave
  if (first_wave && lane < 32) ((uint32_t*)(ws + J.big_off))[lane] = 0;     // ([0, 8) records, [8, 16) cursors, 16 / 17 diagnostics)
  wave_sync();
}

// grid = nshards.  Exclusive scan of cnt[] in (bucket, slice) order, in place;
// cnt[buckets * slices] = number of storable positions of the shard.
DEV void ix_scan(const JobParams& J, const ShardDesc& D, uint8_t* ws) {
  const int lane = wave_lane();
  IxLayout L;
  ix_layout(D.len, J.ix_slices, J.ix_nb_log2, &L);
  uint32_t* cnt = (uint32_t*)(ws + D.ix_off + L.cnt);
  const uint32_t total = J.ix_slices << J.ix_nb_log2;
  const uint32_t per = (total + 63u) / 64u;
  const uint32_t lo = (uint32_t)lane * per, hi = umin(lo + per, total);
  // (eight loads in flight: one at a time, each waited for, was 1.3 ms per GiB for a scan of 32 MiB of counters)
  uint32_t sum = 0;
  for (uint32_t i = lo; i < hi; i += 8u) {
    uint32_t v[8];
#pragma unroll
    for (uint32_t u = 0; u < 8u; ++u) v[u] = i + u < hi ? cnt[i + u] : 0u;
#pragma unroll
    for (uint32_t u = 0; u < 8u; ++u) sum += v[u];
  }
  const uint32_t incl = wave_incl_scan(sum);
  uint32_t run = incl - sum;
  wave_sync();
  for (uint32_t i = lo; i < hi; i += 8u) {
    uint32_t v[8];
#pragma unroll
    for (uint32_t u = 0; u < 8u; ++u) v[u] = i + u < hi ? cnt[i + u] : 0u;
#pragma unroll
    for (uint32_t u = 0; u < 8u; ++u) { if (i + u < hi) cnt[i + u] = run; run += v[u]; }
  }
  if (lane == 63) cnt[total] = incl;
  wave_sync();
}

// grid = nshards * slices.  Stable scatter of the slice's entries to their buckets.
// A row of 64 consecutive positions lands in ~64 different buckets: written straight to HBM that
// is one 4-byte transaction per entry (measured: 1 G transactions bound the kernel at 16 ms per
// GiB).  So the slice goes through LDS in chunks of IX_CHUNK positions: counted, scanned and
// counting-sorted by bucket inside the chunk (entries packed as chunk-relative position | low key bits |
// bucket), then copied out index by index — neighbours in LDS are neighbours in their bucket's
// range, and a bucket's share of a chunk leaves as one contiguous piece.
#ifndef IX_CHUNK
#define IX_CHUNK 2048u
#endif
#define IX_CHUNK_RANKED 48u   // up to this many entries of a chunk in one bucket are ranked by counting
#define IX_SCATTER_LDS_WORDS (IX_CHUNK + 3u * IX_NB_MAX)
DEV void ix_scatter(const JobParams& J, const ShardDesc& D, const uint8_t* input, uint8_t* ws,
                    uint32_t w, uint32_t* lds) {
  const uint32_t lane = (uint32_t)wave_lane();
  const IxGeom g = ix_geom(J, D);
  IxLayout L;
  ix_layout(g.len, J.ix_slices, J.ix_nb_log2, &L);
  uint8_t* base = ws + D.ix_off;
  const uint8_t* data = input + D.in_off;
  const uint32_t* cnt = (const uint32_t*)(base + L.cnt);
  uint32_t* ent = (uint32_t*)(base + L.ent);
  const uint32_t per = ix_slice_len(g.len, J.ix_slices);
  const uint32_t lo = w * per, hi = umin(lo + per, g.len);
  const uint32_t nbk = 1u << J.ix_nb_log2;
  uint32_t* sorted = lds;                       // [IX_CHUNK] packed entries in bucket order
  uint32_t* start = lds + IX_CHUNK;             // [nbk] first LDS index of a bucket in this chunk (counts first)
  uint32_t* cur = start + nbk;                  // [nbk] next free LDS index of a bucket
  uint32_t* glob = cur + nbk;                   // [nbk] next free entry of a bucket in HBM
  for (uint32_t b = lane; b < nbk; b += 64u) { glob[b] = cnt[b * J.ix_slices + w]; start[b] = 0; }
  wave_sync();
  const int shift = J.bucket_bits - (int)J.ix_nb_log2;
  for (uint32_t c0 = lo; c0 < hi; c0 += IX_CHUNK) {
    const uint32_t c1 = umin(c0 + IX_CHUNK, hi);
    uint32_t pk[IX_CHUNK / 64u];
    // (1) hash, count
#pragma unroll
    for (uint32_t r = 0; r < IX_CHUNK / 64u; ++r) {
      const uint32_t x = c0 + r * 64u + lane;
      pk[r] = 0xFFFFFFFFu;
      if (x < c1 && ix_storable(g, x + g.base)) {
        const KeyTag kt = hash_pos(ld64(data + x), J.hasher_type, J.bucket_bits);
        const uint32_t b = kt.key >> shift;
        pk[r] = (x - c0) | ((kt.key & ((1u << shift) - 1u)) << 11) | (b << 19);    // position | low key bits | bucket
        lds_atomic_add(&start[b], 1u);
      }
    }
    wave_sync();
    // (2) counts -> first indices (nbk <= 1024: up to 16 consecutive buckets per lane)
    uint32_t total, biggest = 0;
    {
      const uint32_t per_lane = (nbk + 63u) / 64u;
      uint32_t sum = 0;
      for (uint32_t k = 0; k < per_lane; ++k) {
        const uint32_t b = lane * per_lane + k;
        if (b < nbk) { sum += start[b]; biggest = umax(biggest, start[b]); }
      }
      biggest = wave_max_u32(biggest);
      const uint32_t incl = wave_incl_scan(sum);
      uint32_t run = incl - sum;
      total = wave_bcast(incl, 63);
      wave_sync();
      for (uint32_t k = 0; k < per_lane; ++k) {
        const uint32_t b = lane * per_lane + k;
        if (b >= nbk) break;
        const uint32_t v = start[b];
        start[b] = run; cur[b] = run;
        run += v;
      }
    }
    wave_sync();
    if (biggest <= IX_CHUNK_RANKED) {
      // (3) common case, every bucket holds only a handful of the chunk's entries: slots by LDS
      // atomics (no row waits for the one before), then (4) out — straight if the runs came
      // out in position order, else every entry finds its place by counting the entries of its
      // bucket's run that come before it
#pragma unroll
      for (uint32_t r = 0; r < IX_CHUNK / 64u; ++r) {
        if (pk[r] != 0xFFFFFFFFu) sorted[lds_atomic_add(&cur[pk[r] >> 19], 1u)] = pk[r];
      }
      wave_sync();
      // The LDS unit serves the lanes of one atomic in lane order and a wave's atomics in
      // program order, which already IS position order — but nothing promises the former, so it
      // is checked (one look at the left neighbour) and only a chunk that fails is ranked.
      bool ascending = true;
      for (uint32_t i = lane; i < total; i += 64u) {
        const uint32_t e = sorted[i];
        if (i != start[e >> 19]) ascending = ascending && (sorted[i - 1u] & 2047u) < (e & 2047u);
      }
      if (!wave_ballot(!ascending)) {
        for (uint32_t i = lane; i < total; i += 64u) {
          const uint32_t e = sorted[i], b = e >> 19;
          ent[glob[b] + (i - start[b])] = (c0 + (e & 2047u)) | (((e >> 11) & 255u) << 24);
        }
      } else {
        for (uint32_t i = lane; i < total; i += 64u) {
          const uint32_t e = sorted[i], b = e >> 19;
          const uint32_t s0 = start[b], s1 = cur[b];
          uint32_t before = 0;
          for (uint32_t j = s0; j < s1; ++j) before += (sorted[j] & 2047u) < (e & 2047u) ? 1u : 0u;
          ent[glob[b] + before] = (c0 + (e & 2047u)) | (((e >> 11) & 255u) << 24);
        }
      }
    } else {
      // (3') a crowded bucket (runs, zeros): rows strictly in position order, ranks by match-any
#pragma unroll
      for (uint32_t r = 0; r < IX_CHUNK / 64u; ++r) {
        if (c0 + r * 64u >= c1) break;
        const bool act = pk[r] != 0xFFFFFFFFu;
        const uint32_t b = act ? pk[r] >> 19 : 0u;
        const uint64_t same = ix_match_any(act, b, (int)J.ix_nb_log2);
        const uint32_t rank = (uint32_t)dev_popc64(same & ((1ull << lane) - 1ull));
        const uint32_t group = (uint32_t)dev_popc64(same);
        uint32_t at = 0;
        if (act) at = cur[b];
        wave_sync();
        if (act && rank + 1u == group) cur[b] = at + group;
        if (act) sorted[at + rank] = pk[r];
        wave_sync();
      }
      // (4') out: LDS index i of bucket b goes to glob[b] + (i - start[b])
      for (uint32_t i = lane; i < total; i += 64u) {
        const uint32_t e = sorted[i], b = e >> 19;
        ent[glob[b] + (i - start[b])] = (c0 + (e & 2047u)) | (((e >> 11) & 255u) << 24);
      }
    }
    wave_sync();
    for (uint32_t b = lane; b < nbk; b += 64u) { glob[b] += cur[b] - start[b]; start[b] = 0; }
    wave_sync();
  }
}

// (experiment builds: IX_NT = the entries are read, and srt[] is written, with the streaming hint)
#if defined(IX_NT)
#define IX_LD_STREAM(p) __builtin_nontemporal_load(p)
#define IX_ST_STREAM(p, v) __builtin_nontemporal_store((v), (p))
#else
#define IX_LD_STREAM(p) (*(p))
#define IX_ST_STREAM(p, v) (*(p) = (v))
#endif
// ---- level 2 + window search: one wave per (shard, bucket) -------------------------------
DEV uint32_t ix_score(uint32_t len, uint32_t dist) { return 1920u + dev_mul24(135u, len) - dev_mul24(30u, log2floor(dist)); }
// A candidate as ONE comparable word: score (13 bits: <= 1920 + 135 * IX_CAP) | 16 - j (visiting order: the nearer
// candidate wins a tie, ..64_simd_inc.h:282 `score > best_score` in the order of the walk) | length (6 bits).  The
// arg-max of the bucket loop is a v_max_u32 per candidate; length and distance are read back from the winner.
DEV uint32_t ix_cand_key(uint32_t len, uint32_t dist, uint32_t j) { return (ix_score(len, dist) << 11) | ((16u - j) << 6) | len; }

// The sorted bucket (or a block of it) in LDS, in (key, position) order: w0 = position | tag << 24, bytes 0..7,
// bytes 8..15 of the entry, and a word per entry that first carries what the sort knows of the entry (its place in
// its key run) and then the block's work list (ix_prepare's words, the entries with many candidates first).
struct IxLds { uint32_t* w0; uint64_t* d; uint64_t* d2; uint32_t* aux; };

// ix_prepare's word: LDS slot (9 bits) | successors in the key run, capped at 16 (5) | IX_FULLRUN (1) | IX_DANGER (1)
// | the slots j = 1..16 before the entry that hold a candidate (bit j - 1; 16 bits)
#define IXW_NSUCC_SHIFT 9u
#define IXW_FULLRUN (1u << 14)
#define IXW_DANGER (1u << 15)
#define IXW_NONE 0xFFFFFFFFu

DEV uint64_t ix_res_word(uint32_t kind, uint32_t len, uint32_t dist, uint32_t sidx, uint32_t word) {
  const uint32_t lo = (kind << 30) | (len << 24) | dist;
  const uint32_t hi = sidx | (((word >> IXW_NSUCC_SHIFT) & 31u) << IX_NSUCC_SHIFT) | ((word & IXW_DANGER) ? IX_DANGER : 0u) |
                      ((word & IXW_FULLRUN) ? IX_FULLRUN : 0u);
  return (uint64_t)lo | ((uint64_t)hi << 32);
}

// What the bucket loop of FindLongestMatch (..64_simd_inc.h:246-292) will look at for the entry in LDS slot x:
// of the (up to) 16 entries before it in its key run, the ones with its tag (:263: the tag filter), inside the window
// (a stream, :239-241).  rank = same-key entries before it, nsucc = same-key entries after it, sidx = its index in
// srt[].  Writes srt[].  An entry without a candidate (41 % of the positions of text) gets its res[] here and is
// not listed (IXW_NONE); the others return their word.
// (STREAM: a chunk of a tiled stream — the window limit, the ring's end and the key table exist there only; a plain
//  shard's instantiation carries none of it)
template <bool STREAM>
DEV uint32_t ix_prepare(const IxGeom& g, const IxLds& S, uint32_t x, bool act, uint32_t key, uint32_t rank, uint32_t nsucc,
                        uint32_t sidx, uint32_t* srt, uint64_t* res, uint32_t* kt, uint32_t nkeys, bool* disorder) {
  const uint32_t w = S.w0[x];
  const uint32_t p = w & 0xFFFFFFu;
  const uint32_t P = p + g.base;                                // its position in the shard / stream
  const uint32_t before = S.w0[(int)x - 1] & 0xFFFFFFu;         // (slot -1 of the first entry: never used, rank == 0)
  if (act && rank != 0u && before >= p) *disorder = true;       // a key run must be in position order
  if (STREAM && kt != nullptr && act) {
    // a stream's chunk: the key run's place in srt[], and how much of it lies in the chunk's own part
    if (rank == 0u) { kt[SKT_RS * nkeys + key] = sidx; kt[SKT_RL * nkeys + key] = nsucc + 1u; }
    if (p >= g.ownc && (rank == 0u || before < g.ownc)) kt[SKT_OWN * nkeys + key] = nsucc + 1u;
  }
  if (act) IX_ST_STREAM(&srt[sidx], w);
  const bool search = act && p >= g.own && ix_searchable(g, P);
#if defined(IX_NOWIN)       // (timing experiments only: results are wrong)
  const uint32_t nwin = 0u;
#else
  const uint32_t nwin = search ? umin(rank, 16u) : 0u;
#endif
  const uint32_t maxb = umin(P, g.maxdist);                     // max_backward (backward_references_inc.h:56-57)
  uint32_t mask = 0;
#pragma unroll
  for (uint32_t j = 1; j <= 16u; ++j) {
    const uint32_t qw = S.w0[(int)x - (int)j];                  // (slots below 0 hold other words of the wave's LDS: masked by nwin)
    bool c = ((qw ^ w) >> 24) == 0u;
    if (STREAM) c = c && p - (qw & 0xFFFFFFu) <= maxb;          // beyond the window (:239-241: it and everything older)
    mask |= c ? 1u << (j - 1u) : 0u;
  }
  mask &= (1u << nwin) - 1u;
  // (a stream: what the 16-bit counter hides depends on the stores since the stream's start — k_tile.h finds those
  //  positions once the chunks before this one are parsed)
  const bool danger = !STREAM && rank >= 65520u;
  // (STREAM, g.older: fewer than 16 same-key entries before it here and a window that reaches below the chunk — the
  //  word's IXW_FULLRUN bit, which a stream does not use, carries it to ix_search_block; res[] gets IX_KIND_SLOW)
  const bool older = STREAM && g.older && search && rank < 16u && P - maxb < g.base;
  const uint32_t word = x | (umin(nsucc, 16u) << IXW_NSUCC_SHIFT) | (((!STREAM && rank <= 16u) || older) ? IXW_FULLRUN : 0u) |
                        (danger ? IXW_DANGER : 0u) | (mask << 16);
  if (!act) return IXW_NONE;
  if (mask == 0u) {
    if (older) { res[p] = ix_res_word(IX_KIND_SLOW, 0u, 0u, sidx, word & ~IXW_FULLRUN); return IXW_NONE; }
#if defined(IX_NTRES)     // (experiment: streaming stores)
    __builtin_nontemporal_store(ix_res_word(search && danger ? IX_KIND_SLOW : IX_KIND_NONE, 0u, 0u, sidx, word), &res[p]);
#elif defined(IX_RES4X)    // (timing experiments only: results are wrong)
    { const uint64_t v_ = ix_res_word(search && danger ? IX_KIND_SLOW : IX_KIND_NONE, 0u, 0u, sidx, word);
      if (IX_RES4X != 3 || rank > 16u) ((uint32_t*)res)[p] = (uint32_t)v_;
      if (IX_RES4X >= 6) ((uint16_t*)((uint32_t*)res + g.len))[p] = (uint16_t)(v_ >> 32); }
#elif defined(IX_NORES)    // (timing experiments only: results are wrong)
    { const uint64_t v_ = ix_res_word(search && danger ? IX_KIND_SLOW : IX_KIND_NONE, 0u, 0u, sidx, word); if ((uint32_t)v_ == 0x12345u) res[p] = v_; }
#else
    res[p] = ix_res_word(search && danger ? IX_KIND_SLOW : IX_KIND_NONE, 0u, 0u, sidx, word);
#endif
    return IXW_NONE;
  }
  return word;
}

// The work list of a block: the words of ix_prepare, the entries with many candidates first (three classes: a row of
// 64 lanes then walks about equally many — on text 26 % of the positions have sixteen candidates, 8 % have one).
// Slots by ballots: no LDS atomics, no scan.  words[]: this lane's words, row by row (IXW_NONE: none).  Returns the
// length of the list in S.aux[].
template <int NROWS>
DEV uint32_t ix_worklist(const IxLds& S, const uint32_t (&words)[NROWS], uint32_t nrows) {
  const uint32_t lane = (uint32_t)wave_lane();
  const uint64_t below = (1ull << lane) - 1ull;
  uint32_t n_hi = 0, n_mid = 0, n_lo = 0;
#pragma unroll
  for (uint32_t r = 0; r < (uint32_t)NROWS; ++r) {
    if (r >= nrows) break;
    const bool have = words[r] != IXW_NONE;
    const uint32_t e = (uint32_t)__builtin_popcount(words[r] >> 16);
    n_hi += (uint32_t)dev_popc64(wave_ballot(have && e >= 11u));
    n_mid += (uint32_t)dev_popc64(wave_ballot(have && e >= 4u && e < 11u));
    n_lo += (uint32_t)dev_popc64(wave_ballot(have && e < 4u));
  }
  uint32_t at_hi = 0, at_mid = n_hi, at_lo = n_hi + n_mid;
  wave_sync();
#pragma unroll
  for (uint32_t r = 0; r < (uint32_t)NROWS; ++r) {
    if (r >= nrows) break;
    const bool have = words[r] != IXW_NONE;
    const uint32_t e = (uint32_t)__builtin_popcount(words[r] >> 16);
    const uint64_t b_hi = wave_ballot(have && e >= 11u), b_mid = wave_ballot(have && e >= 4u && e < 11u), b_lo = wave_ballot(have && e < 4u);
    uint32_t slot;
    if (e >= 11u) slot = at_hi + (uint32_t)dev_popc64(b_hi & below);
    else if (e >= 4u) slot = at_mid + (uint32_t)dev_popc64(b_mid & below);
    else slot = at_lo + (uint32_t)dev_popc64(b_lo & below);
    if (have) S.aux[slot] = words[r];
    at_hi += (uint32_t)dev_popc64(b_hi); at_mid += (uint32_t)dev_popc64(b_mid); at_lo += (uint32_t)dev_popc64(b_lo);
  }
  wave_sync();
  return n_hi + n_mid + n_lo;
}

#if defined(IX_PROFILE)
#define IX_T0 , ix_t0, ix_acc
#define IX_LAP(k) do { const uint64_t t_ = (uint64_t)clock64(); ix_acc[k] += t_ - ix_t0; ix_t0 = t_; } while (0)
__device__ uint64_t ix_prof[16];
#else
#define IX_T0
#define IX_LAP(k) do {} while (0)
#endif

// The bucket loop itself for the block's work list (S.aux[0 .. n)), a row of 64 entries at a time: candidates of up
// to 16 bytes are decided from LDS alone — two candidates per lane and round trip —; the ones equal in all 16 bytes
// compare on in the input (bytes 16 .. IX_CAP - 1 in one round trip, four candidates at a time).  Writes res[].
template <bool STREAM>
DEV void ix_search_block(const IxGeom& g, const uint8_t* data, const IxLds& S, uint32_t n, uint32_t sbase, uint64_t* res
#if defined(IX_PROFILE)
                         , uint64_t& ix_t0, uint64_t (&ix_acc)[8]
#endif
                         ) {
  const uint32_t lane = (uint32_t)wave_lane();
  for (uint32_t r0 = 0; r0 < n; r0 += 64u) {
    const bool act = r0 + lane < n;
    const uint32_t word = act ? S.aux[r0 + lane] : 0u;
    const uint32_t x = act ? word & 511u : 16u;
    uint32_t mask = act ? word >> 16 : 0u;
    const uint32_t w = S.w0[x];
    const uint64_t ed = S.d[x], ed2 = S.d2[x];
    const uint32_t p = w & 0xFFFFFFu;
    const uint32_t P = p + g.base;
    const uint32_t max_length = act ? ix_block_end(g, P) - P : 0u;            // (listed: the position is searchable)
    // the ring buffer's physical end (..64_simd_inc.h:243-249: no candidate is looked at once the current position is
    // within best_len of it, one that is within best_len of it is passed over): a candidate whose match does not reach
    // over the end — at either position — loses whenever one of the two rules would have applied to it (best_len is
    // then longer than its match), so only a candidate whose match does reach over it makes the search order-dependent
    // — the chain's to do
    bool ringrisk = false;
    const uint32_t rm = STREAM ? g.ring_mask : 0xFFFFFFFFu;
    uint32_t best = 0, longmask = 0;                                          // ix_cand_key of the best exact candidate
    auto eval = [&](uint32_t j, uint32_t qw, uint64_t qd, uint64_t qd2) {
      const uint32_t q = qw & 0xFFFFFFu;
      const uint64_t xx = qd ^ ed;
      uint32_t l = xx ? ((uint32_t)dev_ctz64(xx) >> 3) : 8u;
      if (l < 4u) return;                                                     // first4 != current4
      if (l == 8u) {
        const uint64_t x2 = qd2 ^ ed2;
        l = x2 ? 8u + ((uint32_t)dev_ctz64(x2) >> 3) : 16u;
#if !defined(IX_NOLONG)     // (timing experiments only: results are wrong)
        if (l == 16u && max_length > 16u) { longmask |= 1u << j; return; }
#endif
      }
      const uint32_t len = umin(l, max_length);
      if (STREAM && (((q + g.base) & rm) + len > rm || (P & rm) + len > rm)) ringrisk = true;
      best = umax(best, ix_cand_key(len, p - q, j));
    };
    while (wave_ballot(mask != 0u) != 0ull) {
      if (mask == 0u) continue;
      const uint32_t j1 = (uint32_t)dev_ctz32(mask) + 1u;
      mask &= mask - 1u;
      const bool two = mask != 0u;
      const uint32_t j2 = two ? (uint32_t)dev_ctz32(mask) + 1u : j1;
      mask &= mask - 1u;
      const int i1 = (int)x - (int)j1, i2 = (int)x - (int)j2;
      const uint32_t qw1 = S.w0[i1], qw2 = S.w0[i2];
      const uint64_t qd1 = S.d[i1], qe1 = S.d2[i1], qd2 = S.d[i2], qe2 = S.d2[i2];
      eval(j1, qw1, qd1, qe1);
      if (two) eval(j2, qw2, qd2, qe2);
    }
    IX_LAP(5);
    uint32_t ncapped = 0, cap_key = 0;
    if (wave_ballot(longmask != 0) != 0) {
      uint64_t mine[3] = {0, 0, 0};
      if (longmask != 0) __builtin_memcpy(mine, data + p + 16u, 24);
      while (wave_ballot(longmask != 0) != 0) {
        // (two capped candidates make the search the chain's — IX_KIND_SLOW — whatever the others are: runs of zeros,
        //  where all 16 candidates are equal for as long as one looks, stop after the first round)
        if (ncapped >= 2u) longmask = 0;
        uint32_t jj[4], qp[4];
        uint64_t c[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          jj[u] = longmask != 0 ? (uint32_t)dev_ctz32(longmask) : 0u;
          longmask &= longmask - 1u;
          qp[u] = S.w0[(int)x - (int)jj[u]] & 0xFFFFFFu;
          c[u][0] = c[u][1] = c[u][2] = 0;
          if (jj[u] != 0) __builtin_memcpy(c[u], data + qp[u] + 16u, 24);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (jj[u] == 0) continue;
          const uint64_t x0 = c[u][0] ^ mine[0], x1 = c[u][1] ^ mine[1], x2 = c[u][2] ^ mine[2];
          const uint32_t ln = x0 ? 16u + ((uint32_t)dev_ctz64(x0) >> 3) : x1 ? 24u + ((uint32_t)dev_ctz64(x1) >> 3) :
                              x2 ? 32u + ((uint32_t)dev_ctz64(x2) >> 3) : 40u;
          const uint32_t len = umin(ln, max_length);
          const bool capped = len == IX_CAP && max_length > IX_CAP;
          {
            const uint32_t reach = capped ? max_length : len;                 // (a capped one: as far as it may go)
            if (STREAM && (((qp[u] + g.base) & rm) + reach > rm || (P & rm) + reach > rm)) ringrisk = true;
          }
          const uint32_t k = ix_cand_key(len, p - qp[u], jj[u]);
          if (capped) { ++ncapped; cap_key = umax(cap_key, k); }
          else best = umax(best, k);
        }
      }
    }
    IX_LAP(6);
    if (act) {
      const bool danger = (word & IXW_DANGER) != 0u;
      uint32_t kind, key = 0;
      const bool older = STREAM && (word & IXW_FULLRUN) != 0u;
      if (danger || ringrisk || ncapped >= 2u || older) kind = IX_KIND_SLOW;
      else if (ncapped == 1u) {
        // the long candidate's score can only grow with its real length
        if (cap_key > best) { kind = IX_KIND_LONG; key = cap_key; }
        else kind = IX_KIND_SLOW;
      } else if (best != 0) { kind = IX_KIND_EXACT; key = best; }
      else kind = IX_KIND_NONE;
      uint32_t len = 0, dist = 0;
      if (key != 0u) {
        len = key & 63u;
        dist = p - (S.w0[(int)x - (int)(16u - ((key >> 6) & 31u))] & 0xFFFFFFu);
      }
#if defined(IX_NTRES)
      __builtin_nontemporal_store(ix_res_word(kind, len, dist, sbase + x, word), &res[p]);
#elif defined(IX_RES4X)       // (timing experiments only: results are wrong)
      { const uint64_t v_ = ix_res_word(kind, len, dist, sbase + x, word);
        if (IX_RES4X != 3 || kind != IX_KIND_NONE || !(word & IXW_FULLRUN)) ((uint32_t*)res)[p] = (uint32_t)v_;
        if (IX_RES4X >= 6) ((uint16_t*)((uint32_t*)res + g.len))[p] = (uint16_t)(v_ >> 32); }
#elif defined(IX_NORES)       // (timing experiments only: results are wrong)
      { const uint64_t v_ = ix_res_word(kind, len, dist, sbase + x, word); if ((uint32_t)v_ == 0x12345u) res[p] = v_; }
#else
      res[p] = ix_res_word(kind, len, dist, sbase + x, STREAM ? word & ~IXW_FULLRUN : word);
#endif
    }
    IX_LAP(7);
  }
}

#define IX_BUCKET_LDS_WORDS (256u + 64u * IX_LROWS * 6u)
#define IX_BROWS (IX_LROWS - 1u)
DEV void ix_lds_put(const IxLds& S, uint32_t i, const IxEntry& e) {
  S.w0[i] = e.w0; S.d[i] = e.d; S.d2[i] = e.d2;
}

// One block of a big bucket (a record of the lists): the sorted entries [k * IX_BIG_BLOCK, ...) of bucket `bucket` of
// unit D, with the 16 entries before them as look-back.  LDS slots 0..15 = the look-back, 16 + i = entry i of the block.
// In three steps, so that a wave has the next block's loads under way while it searches this one.
struct IxBigCtx {
  IxGeom g;
  const uint8_t* data;
  const uint32_t* ent2;
  const uint32_t* keep;
  uint32_t* srt;
  uint64_t* res;
  uint32_t* kt;
  uint32_t start, m, b0, nb, unit, bucket;
};
template <bool STREAM>
DEV void ix_big_ctx(const JobParams& J, const ShardDesc* units, const uint8_t* input, uint8_t* ws, uint64_t rec, IxBigCtx& c) {
  c.unit = (uint32_t)(rec >> 40);
  c.bucket = (uint32_t)(rec >> 28) & 0xFFFu;
  const ShardDesc& D = units[c.unit];
  c.g = ix_geom(J, D);
  IxLayout L;
  ix_layout(c.g.len, J.ix_slices, J.ix_nb_log2, &L);
  uint8_t* base = ws + D.ix_off;
  c.data = input + D.in_off;
  const uint32_t* cnt = (const uint32_t*)(base + L.cnt);
  c.ent2 = (const uint32_t*)(base + L.ent2);
  c.srt = (uint32_t*)(base + L.srt);
  c.res = (uint64_t*)(base + L.res);
  c.kt = nullptr;
  if (STREAM) c.kt = (uint32_t*)(ws + J.skt_off + (uint64_t)(c.g.ownc == 0u ? 0u : (c.g.base >> J.chunk_log2) + 1u) * skt_chunk_bytes((uint32_t)J.bucket_bits));
  c.start = cnt[c.bucket * J.ix_slices];
  c.m = cnt[(c.bucket + 1u) * J.ix_slices] - c.start;
  c.keep = (const uint32_t*)(base + L.ent) + c.start;
  c.b0 = ((uint32_t)rec & 0xFFFFFFFu) * IX_BIG_BLOCK;
  c.nb = umin(IX_BIG_BLOCK, c.m - c.b0);
}
// rows 0 .. 3: the block; "row" 4: the 16 entries before it; binsr: the two bin starts this lane keeps in LDS
struct IxBigRegs { uint32_t wn[IX_BROWS + 1u]; uint64_t bn[IX_BROWS + 1u][2]; uint32_t binsr[2]; };
DEV void ix_big_fetch(const IxBigCtx& c, IxBigRegs& r, bool with_bins = true) {
  const uint32_t lane = (uint32_t)wave_lane();
#pragma unroll
  for (uint32_t q = 0; q < IX_BROWS; ++q) r.wn[q] = c.ent2[c.start + umin(c.b0 + q * 64u + lane, c.m - 1u)];
  r.wn[IX_BROWS] = c.ent2[c.start + (c.b0 >= 16u ? c.b0 - 16u + (lane & 15u) : 0u)];
  if (with_bins) {
    r.binsr[0] = c.keep[2u * lane];
    r.binsr[1] = c.keep[2u * lane + 1u];
  }
#pragma unroll
  for (uint32_t q = 0; q <= IX_BROWS; ++q) __builtin_memcpy(r.bn[q], c.data + (r.wn[q] & 0xFFFFFFu), 16);
}
// The block's entries go from the registers to LDS (keyr: their keys stay with the lanes).
DEV void ix_big_stage(const JobParams& J, const IxBigCtx& c, const IxBigRegs& r, const IxLds& S, uint32_t* bins, uint32_t (&keyr)[IX_BROWS]) {
  const uint32_t lane = (uint32_t)wave_lane();
  wave_sync();
  bins[2u * lane] = r.binsr[0];
  bins[2u * lane + 1u] = r.binsr[1];
#pragma unroll
  for (uint32_t q = 0; q <= IX_BROWS; ++q) {
    IxEntry e;
    e.w0 = r.wn[q]; e.d = r.bn[q][0]; e.d2 = r.bn[q][1];
    const KeyTag kt2 = hash_pos(e.d, J.hasher_type, J.bucket_bits);
    e.w0 = (e.w0 & 0xFFFFFFu) | (kt2.tag << 24);
    if (q < IX_BROWS) {
      keyr[q] = kt2.key;
      if (q * 64u < c.nb) ix_lds_put(S, 16u + q * 64u + lane, e);              // (lanes beyond the block: a copy of the last entry, never used)
    } else if (lane < 16u) ix_lds_put(S, lane, e);                               // (the first block's look-back: never used, rank <= index)
  }
  wave_sync();
}
template <bool STREAM>
DEV bool ix_big_search(const JobParams& J, const IxBigCtx& c, const IxLds& S, const uint32_t* bins, const uint32_t (&keyr)[IX_BROWS], bool checked) {
  const uint32_t lane = (uint32_t)wave_lane();
#if defined(IX_PROFILE)
  uint64_t ix_t0 = (uint64_t)clock64();
  uint64_t ix_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  const int lowbits = J.bucket_bits - (int)J.ix_nb_log2;    // 4 .. 7
  const uint32_t lowmask = (1u << lowbits) - 1u;
  const uint32_t nkeys = STREAM ? 1u << J.bucket_bits : 0u;
  uint32_t words[IX_BROWS];
  bool disorder = false;
#pragma unroll
  for (uint32_t r = 0; r < IX_BROWS; ++r) {
    words[r] = IXW_NONE;
    if (r * 64u >= c.nb) break;
    const uint32_t q = r * 64u + lane, i = c.b0 + q;
    const bool act = q < c.nb;
    const uint32_t kl = keyr[r] & lowmask;
    const uint32_t rank = act ? i - bins[kl] : 0u;
    const uint32_t nsucc = act ? (kl == lowmask ? c.m : bins[kl + 1u]) - i - 1u : 0u;   // entries after this one in its key run
    words[r] = ix_prepare<STREAM>(c.g, S, 16u + (act ? q : 0u), act, keyr[r], rank, nsucc, c.start + i, c.srt, c.res, c.kt, nkeys, &disorder);
  }
  if (checked && wave_ballot(disorder) != 0ull) return true;   // (the caller sorts the bucket again, by ranks)
  const uint32_t nlist = ix_worklist<IX_BROWS>(S, words, (c.nb + 63u) / 64u);
  ix_search_block<STREAM>(c.g, c.data, S, nlist, c.start + c.b0 - 16u, c.res IX_T0);
  wave_sync();
  return false;
}

// lds (words): [0, 128) bin starts, [128, 256) cursors — the 17 counters of ix_worklist once the sort is done —,
// then w0[N], bytes 0..7 [N] (2 words each), bytes 8..15 [N], aux[N]; N = 64 * IX_LROWS entries of the sorted bucket —
// or, for a bigger bucket, 16 entries of look-back and a block of IX_BROWS rows being searched
// (STREAM: the index chunks of a tiled stream — a kernel of its own, k_ix_bucket_s: the plain kernel carries neither
//  the window limit / ring end / key table code nor the registers it pins.)
template <bool STREAM>
DEV void ix_bucket(const JobParams& J, const ShardDesc& D, const uint8_t* input, uint8_t* ws,
                   uint32_t bucket, uint32_t* lds, uint32_t unit) {
  const uint32_t xcd = ((J.flags & JOB_FLAG_IXSPREAD) ? unit + bucket : unit) & 7u;      // (the list its big blocks go to)
  const int lane = wave_lane();
#if defined(IX_PROFILE)
  uint64_t ix_t0 = (uint64_t)clock64();
  uint64_t ix_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  struct IxFlush { uint64_t (&a)[8]; uint32_t b; __device__ ~IxFlush() { if (wave_lane() == 0 && (b & 63u) == 0u) for (int k = 0; k < 8; ++k) atomicAdd((unsigned long long*)&ix_prof[k], (unsigned long long)a[k]); } } ix_flush{ix_acc, bucket};
#endif
  const IxGeom g = ix_geom(J, D);
  IxLayout L;
  ix_layout(g.len, J.ix_slices, J.ix_nb_log2, &L);
  uint8_t* base = ws + D.ix_off;
  const uint8_t* data = input + D.in_off;
  const uint32_t* cnt = (const uint32_t*)(base + L.cnt);
  const uint32_t* ent = (const uint32_t*)(base + L.ent);
  uint32_t* ent2 = (uint32_t*)(base + L.ent2);
  uint32_t* srt = (uint32_t*)(base + L.srt);
  uint64_t* res = (uint64_t*)(base + L.res);
  uint32_t* kt = nullptr;
  if (STREAM) kt = (uint32_t*)(ws + J.skt_off + (uint64_t)(g.ownc == 0u ? 0u : (g.base >> J.chunk_log2) + 1u) * skt_chunk_bytes((uint32_t)J.bucket_bits));
  const uint32_t nkeys = STREAM ? 1u << J.bucket_bits : 0u;
  const uint32_t start = cnt[bucket * J.ix_slices];
  const uint32_t end = cnt[(bucket + 1u) * J.ix_slices];   // (the last one reads the total)
  const uint32_t m = end - start;
  if (m == 0) return;
  const int lowbits = J.bucket_bits - (int)J.ix_nb_log2;    // 4 .. 7
  const uint32_t lowmask = (1u << lowbits) - 1u;
  uint32_t* bins = lds;
  uint32_t* cursor = lds + 128;
  const uint32_t NL = 64u * IX_LROWS;
  IxLds S;
  S.w0 = lds + 256;
  S.d = (uint64_t*)(lds + 256 + NL);
  S.d2 = (uint64_t*)(lds + 256 + 3u * NL);
  S.aux = lds + 256 + 5u * NL;
  wave_sync();
  for (uint32_t b = (uint32_t)lane; b < 128u; b += 64u) bins[b] = 0;
  wave_sync();
  if (m <= NL) {
    // ---- the whole bucket in registers, then sorted into LDS ----
    IxEntry row[IX_LROWS];
#pragma unroll
    for (uint32_t r = 0; r < IX_LROWS; ++r) {
      const uint32_t i = r * 64u + (uint32_t)lane;
      row[r].w0 = row[r].w1 = 0; row[r].d = row[r].d2 = 0;
      if (i < m) row[r].w0 = IX_LD_STREAM(&ent[start + i]);
    }
#pragma unroll
    for (uint32_t r = 0; r < IX_LROWS; ++r) {
      if (r * 64u >= m) break;
      if (r * 64u + (uint32_t)lane < m) {
        ix_fetch(J, data, row[r]);
        lds_atomic_add(&bins[row[r].w1 & lowmask], 1u);
      }
    }
    wave_sync();
    IX_LAP(0);
    {
      const uint32_t a = bins[2 * lane], b = bins[2 * lane + 1];
      const uint32_t incl = wave_incl_scan(a + b);
      wave_sync();
      bins[2 * lane] = incl - a - b;
      bins[2 * lane + 1] = incl - b;
    }
    wave_sync();
    IX_LAP(1);
    uint32_t words[IX_LROWS];
    // The entries arrive in position order.  First the slots are handed out by LDS atomics — the LDS unit serves the
    // lanes of one atomic in lane order and a wave's atomics in program order, which already IS position order —; but
    // nothing promises the former, so ix_prepare checks every key run (one look at the entry before, which it reads
    // anyway) and a bucket that fails is placed once more with its ranks counted by ballots.
    for (uint32_t attempt = 0; attempt < 2u; ++attempt) {
      cursor[2 * lane] = bins[2 * lane];
      cursor[2 * lane + 1] = bins[2 * lane + 1];
      wave_sync();
#pragma unroll
      for (uint32_t r = 0; r < IX_LROWS; ++r) {
        if (r * 64u >= m) break;
        const bool act = r * 64u + (uint32_t)lane < m;
        const uint32_t kl = row[r].w1 & lowmask;
        uint32_t at = 0;
        if (attempt == 0u) {
          if (act) at = lds_atomic_add(&cursor[kl], 1u);
        } else {
          const uint64_t same = ix_match_any(act, kl, lowbits);
          const uint32_t rank = (uint32_t)dev_popc64(same & ((1ull << lane) - 1ull));
          const uint32_t total = (uint32_t)dev_popc64(same);
          if (act) at = cursor[kl];
          wave_sync();
          if (act && rank + 1u == total) cursor[kl] = at + total;
          at += rank;
          wave_sync();
        }
        if (act) {
          ix_lds_put(S, at, row[r]);
          // its place in its key run: same-key entries before it (9 bits), after it (9), the key's low bits
          const uint32_t b0 = bins[kl], b1 = kl == lowmask ? m : bins[kl + 1u];
          S.aux[at] = (at - b0) | ((b1 - at - 1u) << 9) | (kl << 18);
        }
      }
      wave_sync();
      IX_LAP(2);
      bool disorder = false;
#pragma unroll
      for (uint32_t r = 0; r < IX_LROWS; ++r) {
        words[r] = 0xFFFFFFFFu;
        if (r * 64u >= m) break;
        const uint32_t i = r * 64u + (uint32_t)lane;
        const bool act = i < m;
        const uint32_t a = act ? S.aux[i] : 0u;
        words[r] = ix_prepare<STREAM>(g, S, act ? i : 16u, act, (bucket << lowbits) | (a >> 18), a & 511u, (a >> 9) & 511u,
                                      start + i, srt, res, kt, nkeys, &disorder);
      }
      if (wave_ballot(disorder) == 0ull) break;
      if (lane == 0) glb_atomic_add(&((uint32_t*)(ws + J.big_off))[16], 1u);     // (diagnostics: buckets placed a second time)
    }
    IX_LAP(3);
    const uint32_t nlist = ix_worklist<IX_LROWS>(S, words, (m + 63u) / 64u);
    IX_LAP(4);
    ix_search_block<STREAM>(g, data, S, nlist, start, res IX_T0);
    wave_sync();
    return;
  }
  // ---- a bigger bucket: sorted through HBM (ent -> ent2), searched block by block ----
  // (Every bucket of a shard of a MiB, the one bucket a run of zeros fills.  A row's chain — entries, the bytes at
  //  their positions, the search — is a wave's own here, one row after the other, so the loads run ahead of it: four
  //  rows of entries per round trip while counting, the next row's entries under way while this one is ranked, and in
  //  the search the next block's entries and bytes while this one is searched.)
  for (uint32_t r0 = 0; r0 < m; r0 += 256u) {
    uint32_t v[4];
#pragma unroll
    for (uint32_t u = 0; u < 4u; ++u) {
      const uint32_t i = r0 + u * 64u + (uint32_t)lane;
      v[u] = ent[start + umin(i, m - 1u)];
    }
#pragma unroll
    for (uint32_t u = 0; u < 4u; ++u) {
      const uint32_t i = r0 + u * 64u + (uint32_t)lane;
      if (i < m) lds_atomic_add(&bins[v[u] >> 24], 1u);          // (the entry carries its low key bits: no fetch)
    }
  }
  wave_sync();
  {
    const uint32_t a = bins[2 * lane], b = bins[2 * lane + 1];
    const uint32_t incl = wave_incl_scan(a + b);
    wave_sync();
    bins[2 * lane] = incl - a - b;
    bins[2 * lane + 1] = incl - b;
    cursor[2 * lane] = incl - a - b;
    cursor[2 * lane + 1] = incl - b;
  }
  wave_sync();
  // Entries to their places in ent2.  A bucket this wave searches itself (<= ix_giant entries) takes its slots from LDS
  // atomics first, as the small buckets do — the atomics of a wave come out in position order, which the search
  // checks for every key run — and is sorted again with ranks counted by ballots when a check fails; a giant bucket,
  // searched by other waves, is ranked at once.
  for (uint32_t attempt = m > J.ix_giant ? 1u : 0u; attempt < 2u; ++attempt) {
    cursor[2 * lane] = bins[2 * lane];
    cursor[2 * lane + 1] = bins[2 * lane + 1];
    wave_sync();
    {
      uint32_t ahead = ent[start + umin((uint32_t)lane, m - 1u)];
      for (uint32_t r0 = 0; r0 < m; r0 += 64u) {
        const uint32_t i = r0 + (uint32_t)lane;
        const bool act = i < m;
        const uint32_t ew = ahead;
        ahead = ent[start + umin(i + 64u, m - 1u)];
        const uint32_t kl = ew >> 24;
        if (attempt == 0u) {
          if (act) ent2[start + lds_atomic_add(&cursor[kl], 1u)] = ew;
        } else {
          const uint64_t same = ix_match_any(act, kl, lowbits);
          const uint32_t rank = (uint32_t)dev_popc64(same & ((1ull << lane) - 1ull));
          const uint32_t total = (uint32_t)dev_popc64(same);
          uint32_t at = 0;
          if (act) at = cursor[kl];
          wave_sync();
          if (act && rank + 1u == total) cursor[kl] = at + total;
          wave_sync();
          if (act) ent2[start + at + rank] = ew;
        }
      }
    }
    wave_sync();
    if (m > J.ix_giant) {
      // A giant bucket (a run of zeros is ONE bucket of a whole shard): its key runs' starts go where its unsorted
      // entries were (ent[start .. start + 128)), and its blocks of IX_BIG_BLOCK sorted entries onto this XCD's list —
      // k_ix_big searches them, as many waves at a time as there are blocks.
      uint32_t* keep = (uint32_t*)(base + L.ent) + start;
      keep[2 * lane] = bins[2 * lane];
      keep[2 * lane + 1] = bins[2 * lane + 1];
      uint32_t* hdr = (uint32_t*)(ws + J.big_off);
      uint64_t* list = (uint64_t*)(ws + J.big_off + IX_BIG_HEADER_BYTES) + (uint64_t)xcd * J.big_cap;
      const uint32_t nblk = (m + IX_BIG_BLOCK - 1u) / IX_BIG_BLOCK;
      uint32_t at = 0;
      if (lane == 0) at = glb_atomic_add(&hdr[xcd], nblk);
      at = wave_bcast(at, 0);
      for (uint32_t k = (uint32_t)lane; k < nblk; k += 64u)
        if (at + k < J.big_cap) list[at + k] = (uint64_t)k | ((uint64_t)bucket << 28) | ((uint64_t)unit << 40);
      wave_sync();
      return;
    }
    // The others, block by block, by this wave: the next block's entries and bytes under way while this one is searched.
    IxBigCtx c;
    c.g = g; c.data = data; c.ent2 = ent2; c.keep = nullptr; c.srt = srt; c.res = res; c.kt = kt;
    c.start = start; c.m = m; c.unit = unit; c.bucket = bucket;
    IxBigRegs regs;
    regs.binsr[0] = bins[2 * lane];
    regs.binsr[1] = bins[2 * lane + 1];
    c.b0 = 0; c.nb = umin(IX_BIG_BLOCK, m);
    ix_big_fetch(c, regs, false);
    bool again = false;
    for (uint32_t b0 = 0; b0 < m && !again; b0 += IX_BIG_BLOCK) {
      IxBigCtx cur = c;
      cur.b0 = b0; cur.nb = umin(IX_BIG_BLOCK, m - b0);
      uint32_t keyr[IX_BROWS];
      ix_big_stage(J, cur, regs, S, bins, keyr);
      if (b0 + IX_BIG_BLOCK < m) {
        c.b0 = b0 + IX_BIG_BLOCK; c.nb = umin(IX_BIG_BLOCK, m - c.b0);
        ix_big_fetch(c, regs, false);
      }
      again = ix_big_search<STREAM>(J, cur, S, bins, keyr, attempt == 0u);
    }
    wave_sync();
    if (!again) break;                                         // (attempt 1 cannot fail: its order is exact)
    if (lane == 0) glb_atomic_add(&((uint32_t*)(ws + J.big_off))[17], 1u);     // (diagnostics, as word 16)
  }
}

// grid = 8 * waves per XCD; workgroup b works on XCD b % 8's list, IX_BIG_PULL records at a time.
#define IX_BIG_PULL 8u
template <bool STREAM>
DEV void ix_big(const JobParams& J, const ShardDesc* units, const uint8_t* input, uint8_t* ws, uint32_t xcd, uint32_t* lds) {
  uint32_t* hdr = (uint32_t*)(ws + J.big_off);
  const uint64_t* list = (const uint64_t*)(ws + J.big_off + IX_BIG_HEADER_BYTES) + (uint64_t)xcd * J.big_cap;
  const uint32_t total = umin(hdr[xcd], (uint32_t)J.big_cap);
  uint32_t* bins = lds;
  const uint32_t NL = 64u * IX_LROWS;
  IxLds S;
  S.w0 = lds + 256;
  S.d = (uint64_t*)(lds + 256 + NL);
  S.d2 = (uint64_t*)(lds + 256 + 3u * NL);
  S.aux = lds + 256 + 5u * NL;
  for (;;) {
    uint32_t at = 0;
    if (wave_lane() == 0) at = glb_atomic_add(&hdr[8u + xcd], IX_BIG_PULL);
    at = wave_bcast(at, 0);
    if (at >= total) break;
    const uint32_t n = umin(IX_BIG_PULL, total - at);
    IxBigCtx next;
    IxBigRegs regs;
    ix_big_ctx<STREAM>(J, units, input, ws, list[at], next);
    ix_big_fetch(next, regs);
    for (uint32_t u = 0; u < n; ++u) {
      const IxBigCtx cur = next;
      uint32_t keyr[IX_BROWS];
      ix_big_stage(J, cur, regs, S, bins, keyr);
      if (u + 1u < n) {                                        // the next block's loads, under way during this one's search
        ix_big_ctx<STREAM>(J, units, input, ws, list[at + u + 1u], next);
        ix_big_fetch(next, regs);
      }
      (void)ix_big_search<STREAM>(J, cur, S, bins, keyr, false);
    }
  }
}

#endif  // BROTLI_AMD_CSRC_K_INDEX_H_
