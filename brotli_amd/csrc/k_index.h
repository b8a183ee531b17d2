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
struct IxGeom { uint32_t n, first, lgblock, htl, len, base, own, ownc, maxdist, ring_mask; bool stream; };
DEV IxGeom ix_geom(const JobParams& J, const ShardDesc& D) {
  IxGeom g;
  g.stream = D.ix_glen != 0u;
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
                  uint32_t w, uint32_t* lds_cnt) {
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
  for (uint32_t x0 = lo; x0 < hi; x0 += 64u) {
    const uint32_t x = x0 + (uint32_t)lane;
    if (x < hi && ix_storable(g, x + g.base)) {
      const KeyTag kt = hash_pos(ld64(data + x), J.hasher_type, J.bucket_bits);
      lds_atomic_add(&lds_cnt[kt.key >> shift], 1u);
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
  uint32_t sum = 0;
  for (uint32_t i = lo; i < hi; ++i) sum += cnt[i];
  const uint32_t incl = wave_incl_scan(sum);
  uint32_t run = incl - sum;
  wave_sync();
  for (uint32_t i = lo; i < hi; ++i) { const uint32_t v = cnt[i]; cnt[i] = run; run += v; }
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
#define IX_CHUNK 2048u
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

// ---- level 2 + window search: one wave per (shard, bucket) -------------------------------
DEV uint32_t ix_score(uint32_t len, uint32_t dist) { return 1920u + dev_mul24(135u, len) - dev_mul24(30u, log2floor(dist)); }

// The bucket loop of FindLongestMatch (..64_simd_inc.h:246-292) for the entry at index li of the
// LDS arrays (w0 / bytes 0..7 / bytes 8..15, in (key, position) order): the entries before it sit
// at li - 1, li - 2, ...  rank = same-key entries before it, nsucc = same-key entries after it.
// Matches of up to 16 bytes are decided from LDS alone; longer ones compare on in the input.
// Writes srt[] and res[].
struct IxLds { const uint32_t* w0; const uint64_t* d; const uint64_t* d2; };
// (STREAM: a chunk of a tiled stream — the window limit, the ring's end and the key table exist there only; a plain
//  shard's instantiation carries none of it)
template <bool STREAM>
DEV void ix_window(const IxGeom& g, const uint8_t* data, const IxEntry& e, bool act, uint32_t rank, uint32_t nsucc,
                   const IxLds& S, uint32_t li, uint32_t sidx, uint32_t* srt, uint64_t* res, uint32_t* kt = nullptr, uint32_t nkeys = 0) {
  if (STREAM && kt != nullptr && act) {
    // a stream's chunk: the key run's place in srt[], and how much of it lies in the chunk's own part
    const uint32_t pp = e.w0 & 0xFFFFFFu;
    if (rank == 0u) { kt[SKT_RS * nkeys + e.w1] = sidx; kt[SKT_RL * nkeys + e.w1] = nsucc + 1u; }
    if (pp >= g.ownc && (rank == 0u || (S.w0[li - 1u] & 0xFFFFFFu) < g.ownc)) kt[SKT_OWN * nkeys + e.w1] = nsucc + 1u;
  }
  const uint32_t p = e.w0 & 0xFFFFFFu, tag = e.w0 >> 24;
  const uint32_t P = p + g.base;                                // its position in the shard / stream
  // (a stream: what the 16-bit counter hides depends on the stores since the stream's start — k_tile.h finds those
  //  positions once the chunks before this one are parsed)
  const bool danger = !STREAM && rank >= 65520u;
  const bool search = act && p >= g.own && ix_searchable(g, P);
  const uint32_t max_length = search ? ix_block_end(g, P) - P : 0u;
  const uint32_t maxb = umin(P, g.maxdist);                     // max_backward (backward_references_inc.h:56-57)
  // the ring buffer's physical end (..64_simd_inc.h:243-249: no candidate is looked at once the current position is
  // within best_len of it, one that is within best_len of it is passed over): a candidate whose match does not reach
  // over the end — at either position — loses whenever one of the two rules would have applied to it (best_len is then
  // longer than its match), so only a candidate whose match does reach over it makes the search order-dependent —
  // the chain's to do
  bool ringrisk = false;
  const uint32_t rm = STREAM ? g.ring_mask : 0xFFFFFFFFu;
  uint32_t best = 0, best_len = 0, best_dist = 0;               // exact candidates
  uint32_t longmask = 0;
#if defined(IX_NOWIN)       // (timing experiments only: results are wrong)
  const uint32_t nwin = 0u;
#else
  const uint32_t nwin = search ? umin(rank, 16u) : 0u;
#endif
  const uint32_t nmax = (uint32_t)wave_max_u32(nwin);
  for (uint32_t j = 1; j <= nmax; ++j) {
    if (j > nwin) continue;
    const uint32_t qw = S.w0[li - j];
    if ((qw >> 24) != tag) continue;
    const uint64_t x = S.d[li - j] ^ e.d;
    uint32_t l = x ? ((uint32_t)dev_ctz64(x) >> 3) : 8u;
    if (l < 4u) continue;                                       // first4 != current4
    if (STREAM && p - (qw & 0xFFFFFFu) > maxb) continue;        // beyond the window (:239-241: it and everything older)
    if (l == 8u) {
      const uint64_t x2 = S.d2[li - j] ^ e.d2;
      l = x2 ? 8u + ((uint32_t)dev_ctz64(x2) >> 3) : 16u;
#if !defined(IX_NOLONG)     // (timing experiments only: results are wrong)
      if (l == 16u && max_length > 16u) { longmask |= 1u << j; continue; }
#endif
    }
    const uint32_t len = umin(l, max_length);
    if (STREAM && ((((qw & 0xFFFFFFu) + g.base) & rm) + len > rm || (P & rm) + len > rm)) ringrisk = true;
    const uint32_t dist = p - (qw & 0xFFFFFFu);
    const uint32_t k = (ix_score(len, dist) << 5) | (16u - j);
    if (k > best) { best = k; best_len = len; best_dist = dist; }
  }
  // candidates equal in the first 16 bytes: compare on in the input, four candidates per
  // round trip (bytes 16..31 first; the few that are still equal fetch 32..39)
  uint32_t ncapped = 0, cap_key = 0, cap_dist = 0;
  if (wave_ballot(longmask != 0) != 0) {
    uint64_t mine[3] = {0, 0, 0};
    if (longmask != 0) __builtin_memcpy(mine, data + p + 16u, 24);
    while (wave_ballot(longmask != 0) != 0) {
      // (two capped candidates make the search the chain's — IX_KIND_SLOW — whatever the others are: runs of zeros,
      //  where all 16 candidates are equal for as long as one looks, stop after the first round)
      if (ncapped >= 2u) longmask = 0;
      uint32_t jj[4], qp[4], ln[4];
      uint64_t c[4][2];
      bool more = false;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        jj[u] = longmask != 0 ? (uint32_t)dev_ctz32(longmask) : 0u;
        longmask &= longmask - 1u;
        qp[u] = S.w0[li - jj[u]] & 0xFFFFFFu;
        c[u][0] = c[u][1] = 0;
        if (jj[u] != 0) __builtin_memcpy(c[u], data + qp[u] + 16u, 16);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint64_t x0 = c[u][0] ^ mine[0], x1 = c[u][1] ^ mine[1];
        ln[u] = x0 ? 16u + ((uint32_t)dev_ctz64(x0) >> 3) : x1 ? 24u + ((uint32_t)dev_ctz64(x1) >> 3) : 32u;
        if (jj[u] != 0 && ln[u] == 32u && max_length > 32u) more = true;
      }
      if (wave_ballot(more) != 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (jj[u] != 0 && ln[u] == 32u && max_length > 32u) {
            const uint64_t x0 = ld64(data + qp[u] + 32u) ^ mine[2];
            ln[u] = x0 ? 32u + ((uint32_t)dev_ctz64(x0) >> 3) : 40u;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (jj[u] == 0) continue;
        const uint32_t len = umin(ln[u], max_length);
        const uint32_t dist = p - qp[u];
        {
          const uint32_t reach = (len == IX_CAP && max_length > IX_CAP) ? max_length : len;     // (a capped one: as far as it may go)
          if (STREAM && (((qp[u] + g.base) & rm) + reach > rm || (P & rm) + reach > rm)) ringrisk = true;
        }
        const uint32_t k = (ix_score(len, dist) << 5) | (16u - jj[u]);
        if (len == IX_CAP && max_length > IX_CAP) {
          ++ncapped;
          if (k > cap_key) { cap_key = k; cap_dist = dist; }
        } else if (k > best) { best = k; best_len = len; best_dist = dist; }
      }
    }
  }
  if (act) {
    srt[sidx] = e.w0;
    uint32_t kind, len = 0, dist = 0;
    if (!search) kind = IX_KIND_NONE;
    else if (danger || ringrisk || ncapped >= 2u) kind = IX_KIND_SLOW;
    else if (ncapped == 1u) {
      // the long candidate's score can only grow with its real length
      if (cap_key > best) { kind = IX_KIND_LONG; len = IX_CAP; dist = cap_dist; }
      else kind = IX_KIND_SLOW;
    } else if (best != 0) { kind = IX_KIND_EXACT; len = best_len; dist = best_dist; }
    else kind = IX_KIND_NONE;
    const uint32_t lo = (kind << 30) | (len << 24) | dist;
    const uint32_t hi = sidx | (umin(nsucc, 16u) << IX_NSUCC_SHIFT) | (danger ? IX_DANGER : 0u) |
                        ((!STREAM && rank <= 16u) ? IX_FULLRUN : 0u);
#if defined(IX_NORES)        // (timing experiments only: results are wrong)
    if (lo == 0x12345u) res[p] = (uint64_t)lo | ((uint64_t)hi << 32);
#elif defined(IX_RES_SORTED) // (timing experiments only)
    res[sidx] = (uint64_t)lo | ((uint64_t)hi << 32);
#else
    res[p] = (uint64_t)lo | ((uint64_t)hi << 32);
#endif
  }
}

// lds (words): [0, 128) bin starts, [128, 256) cursors, then w0[N], bytes 0..7 [N] (2 words each),
// bytes 8..15 [N], N = 64 * IX_LROWS entries of the sorted bucket — or, for a bigger bucket, the
// (16 + 64) staged entries of the row being searched
#define IX_BUCKET_LDS_WORDS (256u + 64u * IX_LROWS * 5u)
DEV void ix_lds_put(uint32_t* w0S, uint64_t* dS, uint64_t* d2S, uint32_t i, const IxEntry& e) {
  w0S[i] = e.w0; dS[i] = e.d; d2S[i] = e.d2;
}
// (STREAM: the index chunks of a tiled stream — a kernel of its own, k_ix_bucket_s: the plain kernel carries neither
//  the window limit / ring end / key table code nor the registers it pins.)
template <bool STREAM>
DEV void ix_bucket(const JobParams& J, const ShardDesc& D, const uint8_t* input, uint8_t* ws,
                   uint32_t bucket, uint32_t* lds) {
  const int lane = wave_lane();
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
  const uint32_t start = cnt[bucket * J.ix_slices];
  const uint32_t end = cnt[(bucket + 1u) * J.ix_slices];   // (the last one reads the total)
  const uint32_t m = end - start;
  if (m == 0) return;
  const int lowbits = J.bucket_bits - (int)J.ix_nb_log2;    // 4 .. 7
  const uint32_t lowmask = (1u << lowbits) - 1u;
  uint32_t* bins = lds;
  uint32_t* cursor = lds + 128;
  const uint32_t NL = 64u * IX_LROWS;
  uint32_t* w0S = lds + 256;
  uint64_t* dS = (uint64_t*)(lds + 256 + NL);
  uint64_t* d2S = (uint64_t*)(lds + 256 + 3u * NL);
  IxLds S;
  S.w0 = w0S; S.d = dS; S.d2 = d2S;
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
      if (i < m) row[r].w0 = ent[start + i];
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
#pragma unroll
    for (uint32_t r = 0; r < IX_LROWS; ++r) {
      if (r * 64u >= m) break;
      const bool act = r * 64u + (uint32_t)lane < m;
      const uint32_t kl = row[r].w1 & lowmask;
      const uint64_t same = ix_match_any(act, kl, lowbits);
      const uint32_t rank = (uint32_t)dev_popc64(same & ((1ull << lane) - 1ull));
      const uint32_t total = (uint32_t)dev_popc64(same);
      uint32_t at = 0;
      if (act) at = cursor[kl];
      wave_sync();
      if (act && rank + 1u == total) cursor[kl] = at + total;
      if (act) ix_lds_put(w0S, dS, d2S, at + rank, row[r]);
      wave_sync();
    }
    for (uint32_t r0 = 0; r0 < m; r0 += 64u) {
      const uint32_t i = r0 + (uint32_t)lane;
      const bool act = i < m;
      IxEntry e;
      e.w0 = e.w1 = 0xFFFFFFFFu; e.d = e.d2 = 0;
      if (act) { e.w0 = w0S[i]; e.d = dS[i]; e.d2 = d2S[i]; e.w1 = hash_pos(e.d, J.hasher_type, J.bucket_bits).key; }
      const uint32_t kl = e.w1 & lowmask;
      const uint32_t rank = act ? i - bins[kl] : 0u;
      const uint32_t nsucc = act ? (kl == lowmask ? m : bins[kl + 1u]) - i - 1u : 0u;   // entries after this one in its key run
      ix_window<STREAM>(g, data, e, act, rank, nsucc, S, act ? i : 0u, start + i, srt, res, kt, STREAM ? 1u << J.bucket_bits : 0u);
    }
    wave_sync();
    return;
  }
  // ---- a bigger bucket: sorted through HBM (ent -> ent2), searched row by row ----
  // (Every bucket of a shard of a MiB, the one bucket a run of zeros fills.  A row's chain — entries, the bytes at
  //  their positions, the search — is a wave's own here, one row after the other, so the loads run ahead of it: four
  //  rows of entries per round trip while counting, the next row's entries under way while this one is ranked, and in
  //  the search the entries two rows ahead and the bytes one row ahead.)
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
  {
    uint32_t ahead = ent[start + umin((uint32_t)lane, m - 1u)];
    for (uint32_t r0 = 0; r0 < m; r0 += 64u) {
      const uint32_t i = r0 + (uint32_t)lane;
      const bool act = i < m;
      const uint32_t ew = ahead;
      ahead = ent[start + umin(i + 64u, m - 1u)];
      const uint32_t kl = ew >> 24;
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
  wave_sync();
  // staged entries 0..15 = the last 16 entries of the previous row, 16 + lane = this row's
  {
    uint32_t w_next = ent2[start + umin((uint32_t)lane, m - 1u)];             // row 0's entries
    uint64_t b_next[2];
    __builtin_memcpy(b_next, data + (w_next & 0xFFFFFFu), 16);                 // ... and their bytes
    uint32_t w_ahead = ent2[start + umin(64u + (uint32_t)lane, m - 1u)];       // row 1's entries
    for (uint32_t r0 = 0; r0 < m; r0 += 64u) {
      const uint32_t i = r0 + (uint32_t)lane;
      const bool act = i < m;
      IxEntry e;
      e.w0 = e.w1 = 0xFFFFFFFFu; e.d = e.d2 = 0;
      if (act) {
        e.w0 = w_next; e.d = b_next[0]; e.d2 = b_next[1];
        const KeyTag kt2 = hash_pos(e.d, J.hasher_type, J.bucket_bits);
        e.w1 = kt2.key;
        e.w0 = (e.w0 & 0xFFFFFFu) | (kt2.tag << 24);
      }
      // the next row's bytes, the entries of the row behind it
      w_next = w_ahead;
      __builtin_memcpy(b_next, data + (w_next & 0xFFFFFFu), 16);
      w_ahead = ent2[start + umin(i + 128u, m - 1u)];
      ix_lds_put(w0S, dS, d2S, 16u + (uint32_t)lane, e);
      wave_sync();
      const uint32_t kl = e.w1 & lowmask;
      const uint32_t rank = act ? i - bins[kl] : 0u;
      const uint32_t nsucc = act ? (kl == lowmask ? m : bins[kl + 1u]) - i - 1u : 0u;
      ix_window<STREAM>(g, data, e, act, rank, nsucc, S, 16u + (uint32_t)lane, start + i, srt, res, kt, STREAM ? 1u << J.bucket_bits : 0u);
      wave_sync();
      if (lane >= 48) ix_lds_put(w0S, dS, d2S, (uint32_t)lane - 48u, e);   // the next row's look-back
      wave_sync();
    }
  }
}

#endif  // BROTLI_AMD_CSRC_K_INDEX_H_
