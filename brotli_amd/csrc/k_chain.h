// brotli_amd/csrc/k_chain.h — the serial half of the quality-5 LZ77 parse on an
// indexed job (JOB_FLAG_INDEXED, k_index.h): CreateBackwardReferences
// (c/enc/backward_references_inc.h:10-242) and the EncodeData glue around it
// (c/enc/encode.c:905-1173) for up to four shards per wave, 16 lanes each, with
// NO hash table: the bucket part of every FindLongestMatch was evaluated for all
// positions at once by ix_bucket; what is left for the dependency chain is
//
//   * the distance-cache candidates (they depend on the parse): the 16 lanes of a
//     group probe the 4 cache entries at FOUR consecutive positions P .. P+3 in
//     one memory round trip (lane t: position P + t/4, entry t%4);
//   * the state machine (greedy / lazy decisions, literal spree, block glue),
//     replicated per lane, which consumes those four results in order for as long
//     as the position it wants next is the next one evaluated and the distance
//     cache has not changed (a committed copy ends the step).  A typical command —
//     match at P, lazy probe of P + 1 loses, commit — is one step;
//   * book-keeping of which storable positions were NOT stored (k_index.h): a
//     bitmap in HBM, and a taint bit in res[] of the (at most 16) positions that
//     follow an unstored one in its key run — exactly the searches whose bucket
//     window it would have been part of.  A search is taken from the index only if
//     its position is not tainted, the index entry is decidable in isolation
//     (IX_KIND_*), and the bucket winner does not depend on the byte gate
//     (`unsure`, as in k_parse4.h).  Otherwise the
//     group searches the position itself, exactly, from the (key, position)-sorted
//     array: c_search_exact() reproduces the ring contents the reference would
//     hold — the last 16 STORED predecessors of the key run, masked by the 16-bit
//     counter — and resolves them like k_parse4.h does.  JOB_FLAG_FORCE_SLOW sends
//     every search down that path, which pins the two against each other in tests.
#ifndef BROTLI_AMD_CSRC_K_CHAIN_H_
#define BROTLI_AMD_CSRC_K_CHAIN_H_

#include "k_index.h"
#include "k_parse4.h"

#define C_GROUP_LDS_WORDS 24u                                          // 16 ring slots of c_search_exact + its carried store count (16..18)
                                                                       //   + the chunk before the tile's (19: its base or C_NO_OLDER, 20 / 21: its
                                                                       //   index region's offset, 22: its length) + entries walked by exact searches (23)
#define C_NO_OLDER 0xFFFFFFFFu
#define C_LDS_WORDS (Q_GROUPS * C_GROUP_LDS_WORDS)

struct CShard {
  QShard g;
  IxGeom geo;
  uint64_t* res;
  const uint32_t* srt;
  uint32_t ibase;        // position of the local position 0 of `srt`'s entries (a stream's index chunk; 0 otherwise)
  // (a stream whose chunks are shorter than the window, IxGeom::older: where the chunk before the tile's lies is kept in
  //  the group's LDS words 19..22, not here — four more registers in this struct cost the tile kernels 12 % in spills)
  uint8_t* skip;         // bit (x - geo.first): storable position x was NOT stored by the parse
  uint32_t frontier;     // every storable position below it is either stored or marked in `skip`
  uint32_t nslow;
  // tiled jobs (JOB_FLAG_TILED): this group parses [tile_lo, tile_hi) of the shard; only positions of the tile
  // are marked / tainted by it (the bitmap's words never straddle tiles: tiles are multiples of 32 positions
  // counted from geo.first)
  uint32_t tile_lo, tile_hi;
  uint32_t mode;         // C_*
};
#define C_VIEW_ALL 1u     // sweeps: the bitmap holds for every position (round 0: positions below the tile count as stored)
#define C_TILED 2u        // a settled range gets all its bits written, set and cleared (a sweep may settle a range again)
#define C_BAD 4u          // the tiled parse met something it does not handle (counter wrap): the shard goes the serial way

DEV bool c_skipped(const CShard& C, uint32_t q) {
  const uint32_t b = q - C.geo.first;
  return ((C.skip[b >> 3] >> (b & 7u)) & 1u) != 0;
}
// Bits [bi, bi + nb) of the bitmap (nb <= 16): `set` are the marked ones, the others are cleared when `all` (aligned
// dwords only: a tile's bits are whole dwords, so neighbours never write the same word).
DEV void c_bitmap_settle(uint8_t* skip, uint32_t bi, uint32_t nb, uint32_t set, bool all) {
  uint32_t* w = (uint32_t*)skip + (bi >> 5);
  const uint32_t sh = bi & 31u;
  const uint64_t rm = all ? (((1ull << nb) - 1ull) << sh) : 0ull, sm = (uint64_t)set << sh;
  const uint32_t o0 = w[0], n0 = (o0 & ~(uint32_t)rm) | (uint32_t)sm;
  if (n0 != o0) w[0] = n0;
  if (sh + nb > 32u) {
    const uint32_t o1 = w[1], n1 = (o1 & ~(uint32_t)(rm >> 32)) | (uint32_t)(sm >> 32);
    if (n1 != o1) w[1] = n1;
  }
}

DEV uint32_t c_res_hi(const CShard& C, uint32_t x) { return ((const uint32_t*)(C.res + x))[1]; }

// A tainted index result that holds all the same (plain chain only).  IX_TAINT says one of the predecessors of P in
// its key run was not stored; with IX_FULLRUN the index looked at ALL of them (<= 16), so the ring the reference holds
// at P — the stored ones — is a subset of the window the index searched, visited in the same order.  The bucket loop
// is an arg-max with the visiting order as the tie break (the byte gate's dependence on the distance cache is the
// chain's `unsure` rule either way), so: nothing found in the window = nothing found in the ring; and a winner that is
// itself stored wins in the ring too.  The winner q = P - dist is stored when the bitmap says so — or when it lies at or
// behind the frontier: the positions between the frontier and P are the ones this very step searches (and stores,
// ..64_simd_inc.h:293-295) on its way to P, and a step never uses P without having used them.
// (Floats and noise: the literal spree leaves most positions unstored and taints nearly every search; almost all of
//  them find nothing.)
DEV bool c_taint_soft(uint32_t rhi, uint32_t kind) {
  return (rhi & (IX_TAINT | IX_FULLRUN | IX_DANGER)) == (IX_TAINT | IX_FULLRUN) && kind != IX_KIND_SLOW;
}
DEV bool c_taint_holds(const CShard& C, bool soft, uint32_t kind, uint32_t P, uint32_t dist) {
  if (!soft) return false;
  if (kind == IX_KIND_NONE) return true;
  const uint32_t q = P - dist;
  return q >= C.frontier || !c_skipped(C, q);
}

// Marks the storable positions of [a, b) as not stored — except, for a literal spree
// (stride > 1), the ones the spree did store: sfirst + i * stride.  An unstored position x is
// missing from the bucket window of the (at most 16) positions that follow it in its key run:
// their index results no longer hold, so they get IX_TAINT (the chain searches them itself,
// from `srt` and this bitmap).  All of it is this group's own memory: plain loads and stores.
DEV void c_mark_range(const JobParams& J, CShard& C, bool act, uint32_t a, uint32_t b,
                      uint32_t sfirst, uint32_t stride) {
  const int t = q_t();
  (void)J;
  uint32_t cur = umax(a, umax(C.geo.first, C.tile_lo));        // (what lies below the tile is not this group's to settle)
  while (wave_any(act && cur < b)) {
    const bool on = act && cur < b;
    const uint32_t x = cur + (uint32_t)t;
    // (the last three positions of a block are stored by the next block's stitch whatever the parse does —
    //  ..64_simd_inc.h:139-151 — and nothing searches between the block's end and the stitch: never unstored.
    //  A copy of the fast path that runs to the block's end passes over them; found by tools/fuzz_stream_sim.py)
    bool sk = on && x < b && ix_storable(C.geo, x) && x + C.geo.htl <= ix_block_end(C.geo, x);
    if (sk && stride > 1u && ((x - sfirst) % stride) == 0u) sk = false;
    const uint32_t hi = sk ? c_res_hi(C, x) : 0u;
    const uint32_t s = hi & 0xFFFFFFu;
    uint32_t ns = sk ? (hi >> IX_NSUCC_SHIFT) & 31u : 0u;
    {
      // A run (zeros, a gradient's plateaus): the 16 positions are neighbours in their key run too, each with 16
      // successors — 16 dependent round trips for a set of 31 entries.  When the group's successors lie within 64
      // sorted entries they are tainted as one stretch, 16 a step (an entry too many tainted is an exact search too
      // many, never a different result).
      const uint32_t lo_s = ns != 0u ? s + 1u : 0xFFFFFFFFu, hi_s = ns != 0u ? s + ns : 0u;
      const uint32_t glo = ~q_max(~lo_s), ghi = q_max(hi_s);
      const bool dense = on && ghi >= glo && ghi - glo < 64u;
      if (wave_any(dense)) {
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
          const uint32_t i = glo + 16u * k + (uint32_t)t;
          if (dense && i <= ghi) {
            const uint32_t p = (C.srt[i] & 0xFFFFFFu) + C.ibase;
            if (p < C.tile_hi) {
              uint32_t* w = (uint32_t*)(C.res + p) + 1;
              *w = *w | IX_TAINT;
            }
          }
        }
        if (dense) ns = 0;
      }
    }
    // Four successors a round trip: their sorted entries with one load, their res[] words loaded together, then
    // stored (one successor a round was three dependent accesses each — a third of the cycles of a shard of floats,
    // where the literal spree leaves every other position unstored: profiles/r04_h).
    const uint32_t nmax = wave_max_u32(ns);
    for (uint32_t j0 = 0; j0 < nmax; j0 += 4u) {
      uint32_t e4[4] = {0, 0, 0, 0};
      if (j0 < ns) __builtin_memcpy(e4, C.srt + s + j0 + 1u, 16);      // (srt[] ends with 16 bytes of slack: k_index_layout.h)
      uint32_t* w[4];
      uint32_t v[4];
#pragma unroll
      for (uint32_t u = 0; u < 4u; ++u) {
        const uint32_t p = (e4[u] & 0xFFFFFFu) + C.ibase;
        // (a tiled job: successors in later tiles are told by k_tile_events)
        w[u] = (j0 + u < ns && p < C.tile_hi) ? (uint32_t*)(C.res + p) + 1 : nullptr;
        v[u] = w[u] ? *w[u] : 0u;
      }
#pragma unroll
      for (uint32_t u = 0; u < 4u; ++u)
        if (w[u] && !(v[u] & IX_TAINT)) *w[u] = v[u] | IX_TAINT;         // (lanes that hit the same word write the same bit)
    }
    const uint32_t m16 = q_mask16(wave_ballot(sk));
#if defined(BROTLI_AMD_SIMT_SIM)
    if (on && t == 0 && m16 != 0) { g_sim_counts[13] += (unsigned)__builtin_popcount(m16); if (getenv("SIM_SKIPS")) fprintf(stderr, "skip [%u..] mask %x (range %u..%u stride %u)\n", cur, m16, a, b, stride); }
#endif
    if (on && t == 0 && (m16 != 0 || (C.mode & C_TILED) != 0))
      c_bitmap_settle(C.skip, cur - C.geo.first, umin(16u, b - cur), m16, (C.mode & C_TILED) != 0);
    wave_sync();
    if (on) cur += 16u;
  }
}

// Sweeps: [a, b) was stored by the parse — whatever an earlier parse of the tile had marked there is taken back.
DEV void c_clear_range(CShard& C, bool act, uint32_t a, uint32_t b) {
  const int t = q_t();
  act = act && (C.mode & C_VIEW_ALL) != 0;
  uint32_t cur = umax(a, umax(C.geo.first, C.tile_lo));
  while (wave_any(act && cur < b)) {
    if (act && cur < b && t == 0) c_bitmap_settle(C.skip, cur - C.geo.first, umin(16u, b - cur), 0u, true);
    wave_sync();
    if (act && cur < b) cur += 16u;
  }
}

// The parse stored [a, b) (a single position, a StoreRange, the stitch): everything
// storable between the frontier and a was passed over.
DEV void c_stored(const JobParams& J, CShard& C, bool act, uint32_t a, uint32_t b) {
  if (wave_any(act && C.frontier < a)) c_mark_range(J, C, act && C.frontier < a, C.frontier, a, 0, 1);
  {
    const bool clr = act && a < b && (C.mode & C_VIEW_ALL) != 0;
    if (wave_any(clr)) c_clear_range(C, clr, umax(a, C.frontier), b);
  }
  if (act) C.frontier = b;
}

// Match length of data[a..] and data[b..] from offset `off` on (bytes before it are known to
// be equal), limit = bytes available at a.
DEV uint32_t c_extend_from(const uint8_t* data, uint32_t a, uint32_t b, uint32_t limit, uint32_t off) {
  while (off + 8 <= limit) {
    const uint64_t x = ld64(data + a + off) ^ ld64(data + b + off);
    if (x) return off + ((uint32_t)dev_ctz64(x) >> 3);
    off += 8;
  }
  while (off < limit && data[a + off] == data[b + off]) ++off;
  return off;
}

// Inclusive sum over the lanes 0 .. t of a group (row rotations, no LDS).
DEV uint32_t q_incl_scan(uint32_t v) {
  const int t = q_t();
  uint32_t o;
  o = wave_row_ror(v, 1); if (t >= 1) v += o;
  o = wave_row_ror(v, 2); if (t >= 2) v += o;
  o = wave_row_ror(v, 4); if (t >= 4) v += o;
  o = wave_row_ror(v, 8); if (t >= 8) v += o;
  return v;
}

// Exact FindLongestMatch (hash table part + distance cache, ..64_simd_inc.h:201-295) of
// position P for the groups in `want`, from the sorted array.  The caller runs the
// dictionary probe.
DEV QResult c_search_exact(const JobParams& J, CShard& C, bool want, uint32_t P, uint32_t* scratch) {
  QShard& g = C.g;
  const int t = q_t();
  const int ndist = J.ndist;
  const uint32_t max_length = g.pos_end - P;
  const uint32_t max_backward = umin(P, J.max_backward_limit);
  const B32 cur32 = load_b32(g.data + (want ? P : 0u));
  const uint32_t backward = q_dc_entry(g, t & 3);
  const bool d_cand = want && t < ndist && (int32_t)backward > 0 && backward <= max_backward;
  const uint32_t d_prev = P - backward;
  const B32 pd = load_b32(g.data + (d_cand ? d_prev : 0u));
  const KeyTag kt = hash_pos(cur32.q[0], J.hasher_type, J.bucket_bits);
  const uint32_t hi = want ? (uint32_t)(C.res[P] >> 32) : 0u;
  const int32_t sidx = (int32_t)(hi & 0xFFFFFFu);
  // (a stream, k_tile.h: the visible slots of a search behind a counter wrap come with the mark)
  const bool sdanger = (hi & IX_DANGER) != 0 && g.ring_mask_stream != 0u;
  const uint32_t svis = want ? (uint32_t)C.res[P] & 31u : 0u;
  const bool danger = (hi & IX_DANGER) != 0 && !sdanger;
  // the ring: the last 16 stored positions of the key run before P, newest first.  Past rank
  // 65520 the 16-bit store counter of the reference may have wrapped (:250-257), and the number
  // of stores of the whole run decides what is visible: counted once, then carried in the
  // group's LDS words 16..18 (key, index the count reaches, count) — what lies below a searched
  // position never changes — so that a run of 100 000 zeros is not walked again by every search.
  const uint32_t c_key = scratch[16], c_sidx = scratch[17], c_cnt = scratch[18];
  const bool carried = danger && c_key == kt.key && (int32_t)c_sidx <= sidx;
  const int32_t count_from = carried ? (int32_t)c_sidx : 0;
  uint32_t total = carried ? c_cnt : 0u;
  bool counted = !danger;
  // (scratch[0 .. 15]: the sorted indices of the ring's entries, newest first)
  uint32_t found = 0, j0 = 0;
  bool exhausted = false;
  while (wave_any(want && !exhausted && (found < 16u || !counted))) {
    const bool on0 = want && !exhausted && (found < 16u || !counted);
    // ---- 128 entries a round: eight per lane, the round's farthest entry still in the run (then all of them are: a
    // key's entries are one stretch of the sorted array).  A lane's eight positions usually lie within a few dozen
    // bytes of each other — runs of zeros, where a copy stores its last four positions and leaves the rest unstored, so
    // that the 16 stored predecessors lie hundreds of entries back: 13 rounds of 16 entries with three dependent loads
    // each were the whole chain time of the Silesia-style mix (profiles/r04_d) — and their bits of the bitmap come with
    // one 8-byte load. ----
    bool fast = false;
    if (C.mode == 0u) {       // (tiled jobs: measured without effect on the streams that walk far — rows, sparse zeros: profiles/r06_r)
      const int32_t hi_idx = sidx - 1 - (int32_t)j0, far_idx = hi_idx - 127;
      const bool tryf = on0 && far_idx >= 0;
      if (wave_any(tryf)) {
        if (tryf) fast = hash_pos(ld64(g.data + (C.srt[far_idx] & 0xFFFFFFu) + C.ibase), J.hasher_type, J.bucket_bits).key == kt.key;
        if (wave_any(fast)) {
          const int32_t lo_t = hi_idx - 8 * t - 7;                // this lane: entries lo_t .. lo_t + 7, the newest last
          uint32_t e8[8];
          e8[0] = e8[7] = 0;
          if (fast) __builtin_memcpy(e8, C.srt + lo_t, 32);
          const uint32_t p0 = (e8[0] & 0xFFFFFFu) + C.ibase;
          uint32_t z = 0;                                         // bit k: entry lo_t + k was stored
          const bool near = fast && (e8[7] & 0xFFFFFFu) - (e8[0] & 0xFFFFFFu) < 56u;
          if (near) {
            const uint32_t b = p0 - C.geo.first;
            uint64_t bw;
            __builtin_memcpy(&bw, C.skip + (b >> 3), 8);
            bw >>= (b & 7u);
#pragma unroll
            for (int k = 0; k < 8; ++k) z |= (((uint32_t)(bw >> ((e8[k] & 0xFFFFFFu) + C.ibase - p0)) & 1u) ^ 1u) << k;
          } else if (fast) {
            for (int k = 0; k < 8; ++k) z |= (c_skipped(C, (e8[k] & 0xFFFFFFu) + C.ibase) ? 0u : 1u) << k;
          }
          uint32_t zc = z;                                        // ... of them, the ones the store count takes in
          if (lo_t < count_from) { const int32_t sh = count_from - lo_t; zc = sh >= 8 ? 0u : (z >> sh) << sh; }
          const uint32_t cnt = (uint32_t)__builtin_popcount(z);
          const uint32_t incl = q_incl_scan(cnt);
          const uint32_t tot = q_bcast(incl, 15);
          const uint32_t totc = q_bcast(q_incl_scan((uint32_t)__builtin_popcount(zc)), 15);
          uint32_t slot = found + incl - cnt;
          while (z != 0u && slot < 16u) {                         // newest first
            const uint32_t k = 31u - (uint32_t)dev_clz32(z);
            scratch[slot++] = (uint32_t)(lo_t + (int32_t)k);
            z &= ~(1u << k);
          }
          if (fast) {
            found += tot; total += totc;
            j0 += 128u;
            if (sidx - (int32_t)j0 <= count_from) counted = true;
          }
        }
      }
    }
    // ---- entry by entry ----
    const bool on = on0 && !fast;
    const int32_t idx = sidx - 1 - (int32_t)(j0 + (uint32_t)t);
    const bool ok = on && idx >= 0;
    const uint32_t w0 = ok ? C.srt[idx] : 0u;
    const uint32_t q = (w0 & 0xFFFFFFu) + C.ibase;
    bool inrun = false, stored = false;
    if (ok) {
      inrun = hash_pos(ld64(g.data + q), J.hasher_type, J.bucket_bits).key == kt.key;
      // (round 0 of a tiled job: what other tiles skipped is not known yet — taken as stored, k_tile_events tells)
      stored = inrun && !(((C.mode & C_VIEW_ALL) != 0 || q >= C.tile_lo) && c_skipped(C, q));
    }
#if defined(BROTLI_AMD_SIMT_SIM)
    if (getenv("SIM_DBGPOS") && want && ok && P == (uint32_t)atoi(getenv("SIM_DBGPOS")))
      fprintf(stderr, "  WALK P %u idx %d q %u inrun %d stored %d skipbit %d mode %x tile_lo %u\n", P, idx, q, (int)inrun, (int)stored, (int)c_skipped(C, q), C.mode, C.tile_lo);
#endif
    const uint32_t nr16 = q_mask16(wave_ballot(on && !inrun));
    const uint32_t te = nr16 ? (uint32_t)dev_ctz32(nr16) : 16u;
    stored = stored && (uint32_t)t < te;
    const uint32_t s16 = q_mask16(wave_ballot(stored));
    const uint32_t n16 = q_mask16(wave_ballot(stored && idx >= count_from));
    const uint32_t slot = found + (uint32_t)__builtin_popcount(s16 & ((1u << t) - 1u));
    if (stored && slot < 16u) scratch[slot] = (uint32_t)idx;
    if (on) {
      found += (uint32_t)__builtin_popcount(s16);
      total += (uint32_t)__builtin_popcount(n16);
      if (te < 16u) exhausted = true;
      j0 += 16u;
      if (t == 0 && (C.mode & C_TILED) != 0) scratch[23] += 16u;
      if (sidx - (int32_t)j0 <= count_from) counted = true;     // everything from count_from up is in
    }
  }
  const uint32_t* srt_old = nullptr;
  uint32_t ibase_old = C_NO_OLDER;
  {
    // The key run of this chunk is used up and the ring is not full: the entries of the key below the chunk's base
    // are the part of the run of the chunk before that lies in ITS look-back (run start, run length and own-part length
    // from its key table), newest last.  Entries beyond the window end the walk: everything behind them is older.
    // scratch[] takes their indices with bit 31 set.
    const uint32_t nk = 1u << J.bucket_bits;
    ibase_old = scratch[19];
    bool more = want && ibase_old != C_NO_OLDER && found < 16u;   // (the groups of a wave parse tiles of different chunks)
    uint32_t rs2 = 0, n2 = 0;
    if (more) {
      IxLayout L2;
      ix_layout(scratch[22], J.ix_slices, J.ix_nb_log2, &L2);
      srt_old = (const uint32_t*)(g.wsb + (((uint64_t)scratch[21] << 32) | scratch[20]) + L2.srt);
      // (chunk j has the base (j - 1) << chunk_log2: the key table of the chunk that base belongs to)
      const uint32_t* kt_old = (const uint32_t*)(g.wsb + J.skt_off + (uint64_t)((ibase_old >> J.chunk_log2) + 1u) * skt_chunk_bytes((uint32_t)J.bucket_bits));
      rs2 = kt_old[SKT_RS * nk + kt.key];
      n2 = kt_old[SKT_RL * nk + kt.key] - kt_old[SKT_OWN * nk + kt.key];
    }
    uint32_t j2 = 0;
    while (wave_any(more && j2 < n2)) {
      const bool on = more && j2 < n2;
      const bool ok = on && j2 + (uint32_t)t < n2;
      const uint32_t idx2 = rs2 + n2 - 1u - (j2 + (uint32_t)t);
      const uint32_t q = ok ? (srt_old[idx2] & 0xFFFFFFu) + ibase_old : 0u;
      const bool inwin = ok && P - q <= max_backward;
      const bool stored = inwin && !(((C.mode & C_VIEW_ALL) != 0 || q >= C.tile_lo) && c_skipped(C, q));
      const uint32_t s16 = q_mask16(wave_ballot(stored));
      const uint32_t out16 = q_mask16(wave_ballot(ok && !inwin));
      const uint32_t slot = found + (uint32_t)__builtin_popcount(s16 & ((1u << t) - 1u));
      if (stored && slot < 16u) scratch[slot] = idx2 | 0x80000000u;
      if (on) {
        found += (uint32_t)__builtin_popcount(s16);
        j2 += 16u;
        if (t == 0) scratch[23] += 16u;
        if (found >= 16u || out16 != 0u) more = false;
      }
    }
  }
  wave_sync();
  if (want && danger && t == 0) { scratch[16] = kt.key; scratch[17] = (uint32_t)sidx; scratch[18] = total; }
  if (want && danger && (C.mode & C_TILED) != 0) C.mode |= C_BAD;      // (the carried store count is a serial matter)
  // (a tile whose exact searches walk more than 1024 entries per position of the tile — stores so sparse that the 16 stored
  //  predecessors lie thousands of entries back, search after search: a constant background — goes the serial way, where
  //  the ring is a table: k_tile.h tile_walk_over has the measurement)
  if ((C.mode & C_TILED) != 0 && want && (scratch[23] >> 10) > (C.tile_hi - C.tile_lo) + 4096u) C.mode |= C_BAD;
  // slots the 16-bit counter leaves visible (:250-257): all 16 once it has seen 16 stores,
  // and — only reachable after a wrap — count mod 65536 when that is below 16
  uint32_t nvalid = umin(found, 16u);
  if (danger) { const uint32_t n = total & 0xFFFFu; nvalid = n < 16u ? n : 16u; }
  if (sdanger) nvalid = umin(nvalid, svis);
  const uint32_t sc = (uint32_t)t < nvalid ? scratch[t] : 0u;
  const bool from_old = (sc >> 31) != 0u;                       // (set only by the walk above)
  const uint32_t w0 = (uint32_t)t < nvalid ? (from_old ? srt_old[sc & 0x7FFFFFFFu] : C.srt[sc]) : 0u;
  const uint32_t b_prev = (w0 & 0xFFFFFFu) + (from_old ? ibase_old : C.ibase);
  const bool b_cand = want && (uint32_t)t < nvalid && (w0 >> 24) == kt.tag && (P - b_prev) <= max_backward;
  wave_sync();
  uint32_t b_len = 0, d_len = 0;
  {
    B32 pb;
    pb.q[0] = pb.q[1] = pb.q[2] = pb.q[3] = 0;
    if (b_cand) pb = load_b32(g.data + b_prev);
    const uint32_t md = common_prefix32(cur32, pd);
    const uint32_t mb = common_prefix32(cur32, pb);
    bool b_ext = false, d_ext = false;
    if (b_cand) { b_len = umin(mb, max_length); b_ext = mb == 32u && max_length > 32u; }
    if (d_cand) { d_len = umin(md, max_length); d_ext = md == 32u && max_length > 32u; }
    if (wave_any(b_ext || d_ext)) {
      if (b_ext) b_len = q_extend(g.data, P, b_prev, max_length);
      if (d_ext) d_len = q_extend(g.data, P, d_prev, max_length);
    }
  }
  const uint32_t b_score = 1920u + 135u * b_len - 30u * log2floor((P - b_prev) | 1u);
  uint32_t d_score = 135u * d_len + 1935u;
  if (t != 0) d_score -= 39u + ((0x1CA10u >> ((uint32_t)t & 0xEu)) & 0xEu);
  const bool b_ok = b_cand && b_len >= 4u;
  const bool d_ok = d_cand && (d_len >= 3u || (d_len == 2u && t < 2));
  const uint32_t logical = (uint32_t)t;
  const uint32_t d_key = d_ok ? (d_score << 5) | (31u - (uint32_t)t) : 0u;
  const uint32_t b_key = b_ok ? (b_score << 5) | (27u - logical) : 0u;
  const uint32_t d_best = q_max(d_key);
  const uint32_t dc_score = d_best ? (d_best >> 5) : K_MIN_SCORE;
  const uint32_t dc_len = q_max((d_key != 0 && d_key == d_best) ? d_len : 0u);
  const uint32_t dc_len3 = dc_len < 3u ? 3u : dc_len;
  bool unsure = b_ok && b_score > dc_score && b_len <= dc_len3;
  if (g.ring_mask != 0xFFFFFFFFu) {
    // a stream longer than the ring: a search with a candidate at the ring's physical end follows the
    // order-dependent rules of q_resolve_slow
    // (only a candidate whose match reaches over the end can change the outcome: see ix_window)
    const uint32_t rm = g.ring_mask;
    unsure = unsure || (d_cand && ((d_prev & rm) + d_len > rm || (P & rm) + d_len > rm)) ||
             (b_cand && ((b_prev & rm) + b_len > rm || (P & rm) + b_len > rm));
  }
  const bool slow = q_mask16(wave_ballot(unsure)) != 0 || ((J.flags & JOB_FLAG_FORCE_SLOW) != 0 && want);
  const uint32_t best = q_max(b_key > d_key ? b_key : d_key);
  const bool win_is_d = best != 0 && d_key == best;
  const bool win_is_b = best != 0 && b_key == best;
  QResult r;
  r.len = q_max(win_is_d ? d_len : win_is_b ? b_len : 0u);
  r.distance = q_max(win_is_d ? backward : win_is_b ? P - b_prev : 0u);
  r.score = best >> 5;
  r.delta = 0;
  if (best == 0 || r.score <= K_MIN_SCORE) { r.len = 0; r.distance = 0; r.score = K_MIN_SCORE; }
  if (wave_any(slow)) {
    const QResult s = q_resolve_slow(g, want, P, max_length, 0u, ndist, d_ok || d_cand, d_len, d_prev,
                                     d_score, b_cand, b_len, b_prev, b_score);
    if (slow) r = s;
  }
#if defined(BROTLI_AMD_SIMT_SIM)
  if (getenv("SIM_DBGPOS") && want && (!getenv("SIM_DBGSHARD") || g.shard == (uint32_t)atoi(getenv("SIM_DBGSHARD"))) && P == (uint32_t)atoi(getenv("SIM_DBGPOS")))
    fprintf(stderr, "EXACT P %u t %d found %u nvalid %u b_cand %d b_prev %u b_len %u b_score %u d_cand %d d_len %u d_score %u -> len %u dist %u score %u slow %d sidx %d\n",
            P, t, found, nvalid, (int)b_cand, b_prev, b_len, b_score, (int)d_cand, d_len, d_score, r.len, r.distance, r.score, (int)slow, sidx);
#endif
  return r;
}

// One evaluation of the positions P0 .. P0 + NPOS - 1 (lane: position P0 + kpos, distance-cache
// entry idc): distance-cache winner (:201-240) against the bucket result of the index, per quad.
// e_flags = score | evaluated << 31 | needs-the-exact-search << 30.
struct CEval { uint32_t e_flags, e_len, e_dist; };
DEV CEval c_evaluate(const JobParams& J, CShard& C, bool want, uint32_t P0, int kpos, int idc,
                     bool force_slow, uint32_t htl) {
  QShard& g = C.g;
  const uint32_t Pk = P0 + (uint32_t)kpos;
  const bool ev = want && Pk + htl <= g.pos_end;
  const uint32_t max_length = g.pos_end - Pk;
#if defined(C_PD32)
  // 32-byte probes: a longer compare, fewer steps that need a second round trip
  uint64_t cb[4], pb[4];
  __builtin_memcpy(cb, g.data + (ev ? Pk : 0u), 32);
  const uint32_t backward = q_dc_entry(g, idc);
  const bool d_cand = ev && idc < J.ndist && (int32_t)backward > 0 && backward <= umin(Pk, J.max_backward_limit);
  __builtin_memcpy(pb, g.data + (d_cand ? Pk - backward : 0u), 32);
  const uint64_t rw = ev ? C.res[Pk] : 0ull;
  uint32_t d_len = 0;
  {
    const uint64_t x0 = cb[0] ^ pb[0], x1 = cb[1] ^ pb[1], x2 = cb[2] ^ pb[2], x3 = cb[3] ^ pb[3];
    const uint32_t md = x0 ? ((uint32_t)dev_ctz64(x0) >> 3) : x1 ? 8u + ((uint32_t)dev_ctz64(x1) >> 3)
                      : x2 ? 16u + ((uint32_t)dev_ctz64(x2) >> 3) : x3 ? 24u + ((uint32_t)dev_ctz64(x3) >> 3) : 32u;
    bool d_ext = false;
    if (d_cand) { d_len = umin(md, max_length); d_ext = md == 32u && max_length > 32u; }
    if (wave_any(d_ext)) { if (d_ext) d_len = c_extend_from(g.data, Pk, Pk - backward, max_length, 32u); }
  }
#else
  uint64_t cb[2], pb[2];
  __builtin_memcpy(cb, g.data + (ev ? Pk : 0u), 16);
  const uint32_t backward = q_dc_entry(g, idc);
  const bool d_cand = ev && idc < J.ndist && (int32_t)backward > 0 && backward <= umin(Pk, J.max_backward_limit);
  __builtin_memcpy(pb, g.data + (d_cand ? Pk - backward : 0u), 16);
  const uint64_t rw = ev ? C.res[Pk] : 0ull;
  uint32_t d_len = 0;
  {
    const uint64_t x0 = cb[0] ^ pb[0], x1 = cb[1] ^ pb[1];
    const uint32_t md = x0 ? ((uint32_t)dev_ctz64(x0) >> 3) : x1 ? 8u + ((uint32_t)dev_ctz64(x1) >> 3) : 16u;
    bool d_ext = false;
    if (d_cand) { d_len = umin(md, max_length); d_ext = md == 16u && max_length > 16u; }
    if (wave_any(d_ext)) { if (d_ext) d_len = c_extend_from(g.data, Pk, Pk - backward, max_length, 16u); }
  }
#endif
  // Distance-cache winner of the position (:201-240).  The score is 135 * len + 1935 - penalty(i)
  // with penalties 0, 39, 43, 43 < 135: ordering by (len, earlier entry) is ordering by score
  // with the reference's first-wins tie break, so one reduction of len << 2 | (3 - i) is enough.
  const bool d_ok = d_cand && (d_len >= 3u || (d_len == 2u && idc < 2));
  const uint32_t d_key = d_ok ? (d_len << 2) | (3u - (uint32_t)idc) : 0u;
  uint32_t d_best = umax(d_key, wave_quad_xor(d_key, 1));
  d_best = umax(d_best, wave_quad_xor(d_best, 2));
  uint32_t ring_risk = 0;
  if (g.ring_mask != 0xFFFFFFFFu) {      // (the ring's physical end: see c_search_exact)
    const uint32_t rm = g.ring_mask;
    ring_risk = (d_cand && (((Pk - backward) & rm) + d_len > rm || (Pk & rm) + d_len > rm)) ? 1u : 0u;
    ring_risk |= wave_quad_xor(ring_risk, 1);
    ring_risk |= wave_quad_xor(ring_risk, 2);
  }
  const uint32_t dc_len = d_best >> 2;
  const uint32_t dc_i = 3u - (d_best & 3u);
  const uint32_t dc_dist = q_dc_entry(g, (int)dc_i);
  uint32_t dc_score = K_MIN_SCORE;
  if (d_best != 0) {
    dc_score = 135u * dc_len + 1935u;
    if (dc_i != 0) dc_score -= 39u + ((0x1CA10u >> (dc_i & 0xEu)) & 0xEu);
  }
  // the bucket part, from the index
  const uint32_t rlo = (uint32_t)rw, rhi = (uint32_t)(rw >> 32);
  const uint32_t kind = rlo >> 30;
  uint32_t b_len = (rlo >> 24) & 63u;
  const uint32_t b_dist = rlo & 0xFFFFFFu;
  {
    const bool b_long = ev && kind == IX_KIND_LONG;
    if (wave_any(b_long)) { if (b_long) b_len = c_extend_from(g.data, Pk, Pk - b_dist, max_length, IX_CAP); }
  }
  const bool b_ok = ev && (kind == IX_KIND_EXACT || kind == IX_KIND_LONG);
  const uint32_t b_score = b_ok ? 1920u + 135u * b_len - 30u * log2floor(b_dist | 1u) : 0u;
  const bool b_wins = b_ok && b_score > dc_score;
  bool tainted = (rhi & IX_TAINT) != 0;
  {
    const bool soft = ev && C.mode == 0u && c_taint_soft(rhi, kind);
    if (wave_any(soft)) { if (c_taint_holds(C, soft, kind, Pk, b_dist)) tainted = false; }
  }
  const bool need_exact = ev && (kind == IX_KIND_SLOW || (rhi & IX_DANGER) != 0 || tainted || force_slow || ring_risk != 0 ||
                                 (b_wins && b_len <= umax(dc_len, 3u)));
#if defined(BROTLI_AMD_SIMT_SIM)
  if (ev && idc == 0) {   // (statistics of the simulator runs: why positions go to the exact path)
    if (kind == IX_KIND_SLOW) g_sim_counts[8]++;
    else if (tainted) g_sim_counts[9]++;
    else if (b_wins && b_len <= umax(dc_len, 3u)) g_sim_counts[10]++;
    if (kind == IX_KIND_LONG) g_sim_counts[11]++;
    g_sim_counts[12]++;
  }
#endif
  uint32_t e_len = 0, e_dist = 0, e_score = K_MIN_SCORE;
  if (b_wins) { e_len = b_len; e_dist = b_dist; e_score = b_score; }
  else if (d_best != 0) { e_len = dc_len; e_dist = dc_dist; e_score = dc_score; }
  const uint32_t e_flags = e_score | (ev ? 0x80000000u : 0u) | (need_exact ? 0x40000000u : 0u);

  CEval r;
  r.e_flags = e_flags; r.e_len = e_len; r.e_dist = e_dist;
  return r;
}

// ---- the common path of a 16-lane group: one whole command per step, no loops --------------
// (up to four shards per wave.)  Lane t of a group evaluates position pos + t completely: the
// four distance-cache candidates (16 bytes each; a longer one sends the position to the exact
// search) against the bucket result of the index.  With the 16 results in the lanes, what
// CreateBackwardReferences does next (:44-206) is bit arithmetic on three ballots:
//   literals   = positions before the first one with a match (bounded by the literal spree);
//   lazy chain = how many of the following positions beat their predecessor by >= 175 (at most 4);
// then ONE commit per step — distance code, distance cache, command — so every VALU instruction
// of a step advances four encoders by a whole command.  Positions the index cannot decide are
// searched exactly (c_search_exact) when they come first; block boundaries, the literal spree,
// the static dictionary and a lazy chain that runs into an undecidable position leave the loop
// for the generic step of chain_round.  Needs pos + 64 <= pos_end, so that every position of the
// step is searchable and no match of <= 16 compared bytes is cut by the block end.
struct C16 { uint32_t w[4]; };
DEV C16 c_load16(const uint8_t* p) { C16 r; __builtin_memcpy(&r, p, 16); return r; }
// Common prefix of two 16-byte strings in bytes (0 .. 16): straight-line, 16 VALU instructions.
// ffbl gives 0xFFFFFFFF for an all-equal word, so the minimum falls through to the next word.
DEV uint32_t c_prefix16(const C16& a, const C16& b) {
  uint32_t n = umin(dev_ffbl32(a.w[3] ^ b.w[3]), 32u);
  n = umin(dev_ffbl32(a.w[2] ^ b.w[2]), n + 32u);
  n = umin(dev_ffbl32(a.w[1] ^ b.w[1]), n + 32u);
  n = umin(dev_ffbl32(a.w[0] ^ b.w[0]), n + 32u);
  return n >> 3;
}

// dc[i] for a lane-varying i: three selects, no branches.
// (Values, not a pointer to the cache: a select between loads would be turned into a load from a
// selected address and pin the whole shard state in scratch memory.)
DEV uint32_t c_dc_pick(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3, uint32_t i) {
  uint32_t d = d0;
  d = i == 1u ? d1 : d;
  d = i == 2u ? d2 : d;
  d = i == 3u ? d3 : d;
  return d;
}
// ComputeDistanceCode (backward_references.c:87-109) as a chain of selects, last rule first.
DEV uint32_t c_distance_code(uint32_t distance, uint32_t max_distance, uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3) {
  const uint32_t o0 = distance + 3u - d0, o1 = distance + 3u - d1;
  uint32_t c = distance + 15u;
  uint32_t k = distance == d3 ? 3u : c;
  k = distance == d2 ? 2u : k;
  k = o1 < 7u ? (0xFDB1ACEu >> (4u * (o1 & 7u))) & 0xFu : k;
  k = o0 < 7u ? (0x9750468u >> (4u * (o0 & 7u))) & 0xFu : k;
  k = distance == d1 ? 1u : k;
  k = distance == d0 ? 0u : k;
  return distance <= max_distance ? k : c;
}

DEV void c_group_fast(const JobParams& J, const DeviceTables* T, CShard& C, bool alive, uint32_t* scratch, uint32_t& nsteps) {
  QShard& g = C.g;
  const int t = q_t();
  const uint32_t limit = J.max_backward_limit;
  const bool force_slow = (J.flags & JOB_FLAG_FORCE_SLOW) != 0;
  for (;;) {
    {
      // positions passed over since the last store are known now (see chain_round) — but not at
      // the end of a block: the next block's stitch still stores its last three positions
      const bool lag = alive && g.state == Q_SEARCH && g.st_count == 0 && C.frontier < g.position &&
                       g.position + 64u <= g.pos_end;
      if (wave_any(lag)) c_mark_range(J, C, lag, C.frontier, g.position, 0, 1);
      if (lag) C.frontier = g.position;
    }
    const bool can = alive && g.state == Q_SEARCH && g.st_count == 0 && C.frontier == g.position &&
                     g.position + 64u <= g.pos_end;
    if (wave_any(alive && g.state != Q_DONE && !can) || !wave_any(can)) break;
    ++nsteps;
    SIM_COUNT(14, 1);
    uint64_t ft = QP_NOW();
    const uint32_t pos = g.position;
    const uint32_t Pk = pos + (uint32_t)t;
    // ---- evaluation of position Pk ----
    const C16 cb = c_load16(g.data + (can ? Pk : 0u));
    const uint64_t rw = C.res[can ? Pk : C.ibase];      // (a stream: C.res is the chunk's array shifted by its base — index 0 lies gigabytes below it)
    C16 pb[4];
    const uint32_t maxb = umin(Pk, limit);
    const uint32_t dcs[4] = {(uint32_t)g.dc[0], (uint32_t)g.dc[1], (uint32_t)g.dc[2], (uint32_t)g.dc[3]};
    bool d_cand[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t dcv = dcs[i];
      d_cand[i] = can && (dcv - 1u) < maxb;                   // 0 < distance <= max_backward
      pb[i] = c_load16(g.data + (d_cand[i] ? Pk - dcv : 0u));
    }
    // (len, earlier entry) orders the cache candidates like their scores do (see c_evaluate)
    uint32_t d_key = 0;
    bool d_long = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t md = c_prefix16(cb, pb[i]);
      const bool ok = d_cand[i] && md >= (i < 2 ? 2u : 3u);
      d_long = d_long || (d_cand[i] && md == 16u);
      d_key = umax(d_key, ok ? (md << 2) | (3u - (uint32_t)i) : 0u);
    }
    const uint32_t dc_len = (d_key >> 2) & 31u, dc_i = 3u - (d_key & 3u);
    const uint32_t dc_dist = c_dc_pick(dcs[0], dcs[1], dcs[2], dcs[3], dc_i);
    uint32_t dc_score = dev_mul24(135u, dc_len) + 1935u - (dc_i == 0u ? 0u : dc_i == 1u ? 39u : 43u);
    dc_score = d_key != 0 ? dc_score : K_MIN_SCORE;
    const uint32_t rlo = (uint32_t)rw, rhi = (uint32_t)(rw >> 32);
    const uint32_t kind = rlo >> 30, b_len = (rlo >> 24) & 63u, b_dist = rlo & 0xFFFFFFu;
    const uint32_t b_score = 1920u + dev_mul24(135u, b_len) - dev_mul24(30u, log2floor(b_dist | 1u));
    const bool b_wins = kind == IX_KIND_EXACT && b_score > dc_score;
    bool ring_risk = false;
    if (g.ring_mask != 0xFFFFFFFFu) {    // (the ring's physical end: see c_search_exact)
      // (cache candidates of up to 16 compared bytes: a longer one goes to the exact search anyway)
      const uint32_t rm = g.ring_mask;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t md = c_prefix16(cb, pb[i]);
        ring_risk = ring_risk || (d_cand[i] && (((Pk - dcs[i]) & rm) + md > rm || (Pk & rm) + md > rm));
      }
    }
    bool tainted = (rhi & IX_TAINT) != 0;
    {
      const bool soft = can && C.mode == 0u && kind <= IX_KIND_EXACT && c_taint_soft(rhi, kind);
      if (wave_any(soft)) { if (c_taint_holds(C, soft, kind, Pk, b_dist)) tainted = false; }
    }
    const bool need = kind >= IX_KIND_LONG || (rhi & IX_DANGER) != 0 || tainted || d_long || force_slow || ring_risk ||
                      (b_wins && b_len <= umax(dc_len, 3u));
    uint32_t sc = b_wins ? b_score : dc_score;
    uint32_t ln = b_wins ? b_len : dc_len;
    uint32_t ds = b_wins ? b_dist : dc_dist;
    uint32_t use16 = q_mask16(wave_ballot(can && !need));
    QP_ADD(g, 8, ft);
    // ---- an undecidable first position: the exact search, then on with its result ----
    bool dead = false;                                          // this step cannot move the group
    {
      const bool fix = can && (use16 & 1u) == 0u;
      if (wave_any(fix)) {
        const QResult r = c_search_exact(J, C, fix, pos, scratch);
        if (fix) {
          ++C.nslow;
          if (t == 0) { sc = r.score; ln = r.len; ds = r.distance; }
          use16 |= 1u;
        }
      }
    }
    uint32_t hit16 = q_mask16(wave_ballot(sc > K_MIN_SCORE));
    // ---- nothing found at the first position while the static dictionary is being consulted
    // (hash.h:179-202): probe it here.  A dictionary match starts the lazy evaluation, which the
    // generic step carries on with (it may ask the dictionary again at the next position).
    bool gate_closed = g.dict_matches < (g.dict_lookups >> 7);
    bool dict0 = false;                                         // the dictionary was asked about position 0, in vain
    // literals allowed before the spree check trips (:208)
    const uint32_t room = g.apply_random_heuristics >= pos ? g.apply_random_heuristics - pos : 0u;
    {
      // (not when the literal spree trips right behind this position: that step belongs to the generic path,
      // which searches the position itself — asking here as well would count the two lookups twice and close
      // the gate earlier than the reference does; found by tools/fuzz_index_sim.py)
      const bool dq = can && !gate_closed && (hit16 & 1u) == 0u && room != 0u;
      if (wave_any(dq)) {
        QResult r;
        r.len = 0; r.distance = 0; r.score = K_MIN_SCORE; r.delta = 0;
        q_dict_search(J, T, g, dq, pos, g.pos_end - pos, r);
        if (dq) {
          if (r.score > K_MIN_SCORE) {
            g.sr_len = r.len; g.sr_dist = r.distance; g.sr_score = r.score; g.sr_delta = r.delta;
            g.delayed = 0;
            g.state = Q_LAZY;
            g.stat_searches++;
            C.frontier = pos + 1u;
            dead = true;
          } else dict0 = true;
        }
      }
    }
    QP_ADD(g, 9, ft);
    // lane t: does position t + 1 beat position t by the lazy-matching margin (:139)?
    const uint32_t sc_next = wave_row_ror(sc, 15);              // lane t reads lane (t + 1) & 15 of its group
    const uint32_t adv16 = q_mask16(wave_ballot(t < 15 && sc_next >= sc + 175u));
    const uint32_t U = (uint32_t)dev_ctz32(~use16 | 0x10000u);                  // usable prefix
    const uint32_t m = (uint32_t)dev_ctz32(hit16 | 0x10000u);                   // first match
    // ... / the dictionary has to be asked
    const uint32_t missmax = gate_closed ? room : dict0 ? umin(room, 1u) : 0u;
    const uint32_t tt = umin((uint32_t)dev_ctz32(~(adv16 >> (m & 15u)) | 0x10u), 4u);   // positions the match is delayed by
    const uint32_t last = m + umin(tt + 1u, 4u);                                // last position probed
    // while the static dictionary is being consulted, a probe that finds nothing asks it:
    // only probes with a match are decidable here
    const uint32_t probes = (0x3FFFFu >> (17u - umin(last, 16u))) & ~(0x1FFFFu >> (16u - umin(m, 16u)));   // bits m + 1 .. last
    const bool have = can && !dead && m < 16u && m <= missmax && last < U &&
                      (gate_closed || (hit16 & probes) == probes);
    const uint32_t L = umin(umin(m, U), missmax);               // literals this step may consume without a match
    if (can && !dead && !have && L != 0u) {
      g.position = pos + L;
      g.insert_length += L;
      g.stat_searches += L;
      C.frontier = pos + L;
    }
    if (can && !have && (dead || L == 0u)) { g.status |= 0x80000000u; }         // needs the generic step
#if defined(BROTLI_AMD_SIMT_SIM)
    if (can && t == 0) {
      if (have) g_sim_counts[0]++;                       // steps with a commit
      else if (dead) g_sim_counts[2]++;                  // dictionary match: lazy evaluation handed to the generic step
      else if (L != 0u) g_sim_counts[1]++;               // literal-only steps
      else if (m < 16u && m < U && m <= missmax && last >= U) g_sim_counts[3]++;   // lazy chain runs into an undecidable position
      else if (m < 16u && m < U && m <= missmax) g_sim_counts[4]++;                // lazy probe without a match, dictionary gate open
      else if (missmax == 0u) g_sim_counts[5]++;         // spree / dictionary gate at the first position
      else g_sim_counts[6]++;
    }
#endif
    QP_ADD(g, 10, ft);
    if (wave_any(have)) {
      // ---- one commit (:165-206) ----
      const uint32_t f = m + tt;
      const int src = q_base() | (int)(f & 15u);
      const uint32_t sr_len = wave_shfl(ln, src), sr_dist = wave_shfl(ds, src);
      if (have && sr_len < 2u) { g.status |= QST_ERROR | QST_DONE; g.state = Q_DONE; }   // cannot happen: fail, do not spin
      const uint32_t pc = pos + f;                              // the copy starts here
      const uint32_t ins = g.insert_length + f;
      const uint32_t newpos = pc + sr_len;
      uint32_t range_start = pc + 2u;
      const uint32_t range_end = umin(newpos, g.store_end);
      if (sr_dist < (sr_len >> 2)) range_start = umin(range_end, umax(range_start, newpos - (sr_dist << 2)));
      const bool ranged = range_start < range_end;
      if (have) C.frontier = pos + last + 1u;                   // searched positions are stored (:293-295)
      const bool gap1 = have && ranged && C.frontier != range_start;
      if (wave_any(gap1)) c_mark_range(J, C, gap1, C.frontier, range_start, 0, 1);
      if (have && ranged) C.frontier = range_end;
      const bool gap2 = have && C.frontier < newpos;
      if (wave_any(gap2)) c_mark_range(J, C, gap2, C.frontier, newpos, 0, 1);
      if (have) {
        C.frontier = umax(C.frontier, newpos);
        const uint32_t dictionary_start = umin(pc + g.stream_offset, limit);
        const uint32_t code = c_distance_code(sr_dist, dictionary_start, dcs[0], dcs[1], dcs[2], dcs[3]);
        if (sr_dist <= dictionary_start && code > 0u) {
          g.dc[3] = g.dc[2]; g.dc[2] = g.dc[1]; g.dc[1] = g.dc[0]; g.dc[0] = (int32_t)sr_dist;
        }
        const uint32_t dflags = q_dict_flags(g);
        if (t == 0) {
          Command c;
          c.insert_len = ins; c.copy_len = sr_len; c.dist_extra = code; c.cmd_prefix = CMD_RAW;
          c.dist_prefix = (uint16_t)((tt == 4u ? CMDF_NOPROBE : 0u) | dflags | (umin(tt, 3u) << CMDF_DELAYED_SHIFT));
          g.cmds[g.r.ncmds] = c;
        }
        ++g.r.ncmds;
        g.r.nlits += ins;
        g.insert_length = 0;
        g.apply_random_heuristics = pc + 2u * sr_len + J.spree_window;
        g.position = newpos;
        g.stat_searches += last + 1u;
      }
    }
    QP_ADD(g, 11, ft);
    if (wave_any((g.status & 0x80000000u) != 0)) break;
  }
  g.status &= 0x7FFFFFFFu;
}

// ---- tiled jobs: replay of a tile's previous commands (sweeps) ---------------------------------------
// A sweep (JOB_FLAG_SWEEP) walks the commands the tile's last parse produced and takes each one over as long as
// nothing it was decided from has changed: the group is at the command's boundary with the same distance cache,
// no event (k_tile.h: a position whose bucket ring differs from what the last parse saw) is pending in the
// positions the command's decision searched — its insert run, the copy's start and the lazy probe behind it — and
// neither the literal spree nor the static dictionary had a say.  Where that does not hold the generic step of
// chain_round parses for real (every search exact, with the bitmap as it stands) until its state meets the old
// command list again.
struct CReplay {
  const Command* old;    // the tile's previous commands
  uint32_t oi, on;       // next old command / one past the last copy command
  uint32_t obnd;         // position where old command oi's insert run begins
  int32_t odc[4];        // distance cache of the old parse there
  const uint8_t* ev;     // event bitmap, bit (x - first)
  uint32_t next_ev;      // no event in [the group's position, next_ev); next_ev is one, or the tile's end
  uint32_t changed;      // generic commits of this sweep
};
// Distance an old command's code stands for (ComputeDistanceCode backwards, backward_references.c:87-109).
DEV uint32_t c_code_distance(uint32_t code, int32_t d0, int32_t d1, int32_t d2, int32_t d3) {
  if (code >= 16u) return code - 15u;
  if (code < 4u) return c_dc_pick((uint32_t)d0, (uint32_t)d1, (uint32_t)d2, (uint32_t)d3, code);
  const uint32_t k = code - 4u, kk = k < 6u ? k : k - 6u;
  const uint32_t base = (uint32_t)(k < 6u ? d0 : d1), mag = (kk >> 1) + 1u;
  return (kk & 1u) ? base + mag : base - mag;
}
// First event at or behind `from` (for the groups in `act`), `limit` if there is none below it: the 16 lanes of
// a group look at 16 words of the bitmap per round.
DEV uint32_t c_next_event(const uint8_t* ev, uint32_t first, bool act, uint32_t from, uint32_t limit) {
  const uint32_t t = (uint32_t)q_t();
  const uint32_t* w = (const uint32_t*)ev;
  uint32_t word0 = (from - first) >> 5;
  uint32_t found = limit;
  bool scanning = act && from < limit;
  while (wave_any(scanning)) {
    const uint32_t i = word0 + t;
    uint32_t v = 0;
    if (scanning && i * 32u + first < limit) v = w[i];
    if (i == ((from - first) >> 5)) v &= 0xFFFFFFFFu << ((from - first) & 31u);
    const uint32_t m16 = q_mask16(wave_ballot(v != 0));
    const int j = m16 ? dev_ctz32(m16) : 0;
    const uint32_t vj = q_bcast(v, j);
    if (scanning) {
      if (m16 != 0) { found = umin(limit, first + (word0 + (uint32_t)j) * 32u + (uint32_t)dev_ctz32(vj | 0x80000000u)); scanning = false; }
      else { word0 += 16u; if (word0 * 32u + first >= limit) scanning = false; }
    }
  }
  return found;
}
// Sixteen old commands per round, one per lane: their boundaries by a scan, their distances and the distance
// cache by a walk over the lanes in registers, the conditions per lane; the leading commands that hold are
// taken over at once.
// Score FindLongestMatch gave a match of (len, distance) under the distance cache d0..d3 (hash.h:123-138,
// ..64_simd_inc.h:201-240): a distance the cache holds was found as a cache candidate first, with the bonus.
DEV uint32_t c_match_score(uint32_t len, uint32_t dist, int32_t d0, int32_t d1, int32_t d2, int32_t d3) {
  if (dist == (uint32_t)d0) return 135u * len + 1935u;
  if (dist == (uint32_t)d1) return 135u * len + 1935u - 39u;
  if (len >= 3u && (dist == (uint32_t)d2 || dist == (uint32_t)d3)) return 135u * len + 1935u - 43u;
  return 1920u + 135u * len - 30u * log2floor(dist | 1u);
}
DEV void c_group_replay(const JobParams& J, CShard& C, CReplay& R, bool alive, uint32_t htl, uint32_t* scratch) {
  QShard& g = C.g;
  const int t = q_t();
  const uint32_t limit = J.max_backward_limit;
  for (;;) {
    bool can = alive && g.state == Q_SEARCH && g.st_count == 0;
    const uint32_t nb = g.position - g.insert_length;
    // the old parse's state at (or past) this boundary: only behind generic commits is there anything to skip
    while (can && R.oi < R.on && R.obnd < nb) {
      const Command c = R.old[R.oi];
      const uint32_t code = c.dist_extra, pc = R.obnd + c.insert_len;
      const uint32_t dist = c_code_distance(code, R.odc[0], R.odc[1], R.odc[2], R.odc[3]);
      if (dist <= umin(pc + g.stream_offset, limit) && code > 0u) { R.odc[3] = R.odc[2]; R.odc[2] = R.odc[1]; R.odc[1] = R.odc[0]; R.odc[0] = (int32_t)dist; }
      R.obnd = pc + (c.copy_len & 0x1FFFFFFu);
      ++R.oi;
    }
    can = can && R.oi < R.on && R.obnd == nb && R.odc[0] == g.dc[0] && R.odc[1] == g.dc[1] && R.odc[2] == g.dc[2] && R.odc[3] == g.dc[3];
    // the next event at or behind the group's position
    {
      const bool rescan = can && g.position > R.next_ev;
      if (wave_any(rescan)) {
        const uint32_t ne = c_next_event(R.ev, C.geo.first, rescan, g.position, C.tile_hi);
        if (rescan) R.next_ev = ne;
      }
    }
    const uint32_t k = R.oi + (uint32_t)t;
    const bool have = can && k < R.on;
    Command c;
    c.insert_len = c.copy_len = c.dist_extra = 0; c.cmd_prefix = c.dist_prefix = 0;
    if (have) c = R.old[k];
    const uint32_t I = c.insert_len, L = c.copy_len & 0x1FFFFFFu, code = c.dist_extra;
    const uint32_t span = have ? I + L : 0u;
    const uint32_t bnd = R.obnd + q_incl_scan(span) - span;      // the boundary before this lane's command
    const uint32_t pc = bnd + I;
    const uint32_t dictionary_start = umin(pc + g.stream_offset, limit);
    // distances and the cache behind every command, lane by lane (registers only)
    int32_t x0 = g.dc[0], x1 = g.dc[1], x2 = g.dc[2], x3 = g.dc[3];
    int32_t a0 = x0, a1 = x1, a2 = x2, a3 = x3;                  // ... behind this lane's command
    uint32_t dist = 0;
    const uint32_t nhave = (uint32_t)__builtin_popcount(q_mask16(wave_ballot(have)));
    const uint32_t nmax = wave_max_u32(nhave);
    for (uint32_t j = 0; j < nmax; ++j) {
      const uint32_t cj = q_bcast(code, (int)j), dsj = q_bcast(dictionary_start, (int)j);
      const uint32_t dj = c_code_distance(cj, x0, x1, x2, x3);
      if (j < nhave && dj <= dsj && cj > 0u) { x3 = x2; x2 = x1; x1 = x0; x0 = (int32_t)dj; }
      if ((uint32_t)t == j) { dist = dj; a0 = x0; a1 = x1; a2 = x2; a3 = x3; }
    }
    const uint32_t L0 = umin(L, g.pos_end - umin(pc, g.pos_end));        // (ExtendLastCommand adds the rest again at the next block)
    const uint32_t arh_next = pc + 2u * L0 + J.spree_window;
    const uint32_t arh_prev = wave_row_ror(arh_next, 1);
    const uint32_t arh = t == 0 ? g.apply_random_heuristics : arh_prev;
    const uint32_t from = t == 0 ? g.position : bnd;
    const bool ok = have && pc >= from && pc + htl < g.pos_end && pc <= arh && (c.dist_prefix & CMDF_SPREE) == 0 &&
                    (c.copy_len >> 25) == 0u && dist <= dictionary_start && L >= 2u && c.cmd_prefix == CMD_RAW &&
                    umin(pc + 1u, g.pos_end - 1u) < R.next_ev;
    const bool okn = have && pc >= from && pc + htl < g.pos_end && pc <= arh && (c.dist_prefix & CMDF_SPREE) == 0 &&
                     (c.copy_len >> 25) == 0u && dist <= dictionary_start && L >= 2u && c.cmd_prefix == CMD_RAW;
    const uint32_t m = (uint32_t)dev_ctz32(~q_mask16(wave_ballot(ok)) | 0x10000u);   // leading commands that hold
    SIM_COUNT(15, 1);
    // The group's first command holds but for an event in the positions it searched: instead of parsing it again, the
    // searches at the event positions are done again (exactly, with the bitmap as it stands) and compared with what the
    // command says their results were — no match at a literal in front of the lazy chain, the copy itself at the copy's
    // start, a loser (by the margin of 175, backward_references_inc.h:139) at the probe behind it.  Only a result that
    // differs — or one the command does not tell, a position inside the lazy chain — sends the group to the generic step.
    {
      const bool first_okn = q_bcast(okn ? 1u : 0u, 0) != 0;
      bool qc = m == 0u && first_okn && g.dict_matches < (g.dict_lookups >> 7);
      if (wave_any(qc)) {
        const uint32_t pc0 = q_bcast(pc, 0), L0c = q_bcast(L0, 0), dist0 = q_bcast(dist, 0), fl0 = q_bcast((uint32_t)c.dist_prefix, 0);
        const uint32_t delayed = (fl0 >> CMDF_DELAYED_SHIFT) & 3u;
        const uint32_t rend = umin(pc0 + 1u, g.pos_end - 1u);
        const uint32_t sc_commit = c_match_score(L0c, dist0, g.dc[0], g.dc[1], g.dc[2], g.dc[3]);
        while (wave_any(qc)) {
          const uint32_t e = R.next_ev;
          bool done = qc && e > rend;                      // every event of the range is dealt with: the command holds
          bool bad = qc && !done && e < pc0 && e + delayed >= pc0;      // inside the lazy chain
          if (qc && !done && e == pc0 && delayed == 3u) bad = true;     // (3 = three or more: the chain's start is not known)
          const bool skip_probe = qc && !done && !bad && e == pc0 + 1u && (fl0 & CMDF_NOPROBE) != 0;   // (was not searched)
          const bool srch = qc && !done && !bad && !skip_probe;
          if (wave_any(srch)) {
            const QResult r = c_search_exact(J, C, srch, e, scratch);
            if (srch) {
              if (e < pc0) bad = r.score != K_MIN_SCORE;
              else if (e == pc0) bad = !(r.len == L0c && r.distance == dist0);
              else bad = r.score >= sc_commit + 175u;
            }
          }
          const bool next = qc && !done && !bad;
          if (wave_any(next)) {
            const uint32_t ne = c_next_event(R.ev, C.geo.first, next, e + 1u, C.tile_hi);
            if (next) R.next_ev = ne;
          }
          if (done || bad) qc = false;
          if (bad) R.changed |= 0x80000000u;               // (statistics: quick checks that failed)
        }
      }
    }
    // (a group whose quick check passed has its first command hold now: it takes it in the next round)
    const bool again = m == 0u && have && okn && umin(pc + 1u, g.pos_end - 1u) < R.next_ev && t == 0;
    const bool retry = q_bcast(again ? 1u : 0u, 0) != 0;
    if (!wave_any(m != 0) && !wave_any(retry)) break;
    const uint32_t ins_sum = q_incl_scan(have ? I : 0u);
    // the static dictionary's two counters move as they did when the commands were decided (hash.h:49-50, 186)
    const uint32_t dict_sum = q_incl_scan(have ? ((uint32_t)(c.dist_prefix >> CMDF_LOOKUPS_SHIFT) & 255u) | (((uint32_t)c.dist_prefix >> 10) & 15u) << 16 : 0u);
    if ((uint32_t)t < m) {
      Command n;
      n.insert_len = I; n.copy_len = L0; n.dist_extra = code; n.cmd_prefix = CMD_RAW; n.dist_prefix = c.dist_prefix;
      g.cmds[g.r.ncmds + (uint32_t)t] = n;
    }
    {
      // the state behind the last command taken over (every lane takes part in the shuffles)
      const int last = q_base() | (int)(m != 0 ? m - 1u : 0u);
      const uint32_t n0 = wave_shfl((uint32_t)a0, last), n1 = wave_shfl((uint32_t)a1, last);
      const uint32_t n2 = wave_shfl((uint32_t)a2, last), n3 = wave_shfl((uint32_t)a3, last);
      const uint32_t nl = wave_shfl(ins_sum, last), na = wave_shfl(arh_next, last);
      const uint32_t np = wave_shfl(pc + L0, last), no = wave_shfl(pc + L, last), nd = wave_shfl(dict_sum, last);
      if (m != 0) {
        g.dc[0] = (int32_t)n0; g.dc[1] = (int32_t)n1; g.dc[2] = (int32_t)n2; g.dc[3] = (int32_t)n3;
        R.odc[0] = g.dc[0]; R.odc[1] = g.dc[1]; R.odc[2] = g.dc[2]; R.odc[3] = g.dc[3];
        g.r.ncmds += m;
        g.r.nlits += nl;
        // (counted from the mark = the counters behind the last commit: the generic steps may have searched — and
        //  looked up — a part of the first command's insert run already, and the command's counts cover all of it)
        g.dict_lookups = g.dict_mark_l + (nd & 0xFFFFu); g.dict_matches = g.dict_mark_m + (nd >> 16);
        g.dict_mark_l = g.dict_lookups; g.dict_mark_m = g.dict_matches;
        g.insert_length = 0;
        g.apply_random_heuristics = na;
        g.position = np;
        C.frontier = umax(C.frontier, g.position);
        R.obnd = no;
        R.oi += m;
      }
    }
    if (wave_any(alive && g.state != Q_DONE && m == 0u && !retry)) break;      // somebody needs the generic step
  }
}

// ---- the kernel body ---------------------------------------------------------------------
// Up to four units per wave (q_groups_per_wave): group gi of wave w serves unit w * gpw + gi — a shard, or, in a
// tiled job, a tile of a shard (`tiles`).
DEV Command* c_tile_slot(uint8_t* ws, const ShardDesc& D, const JobParams& J, uint32_t buf, uint32_t tt) {
  return (Command*)(ws + D.cmds2_off) + ((uint64_t)buf * D.ntiles + tt) * tile_slot_cmds(J.tile_log2, (uint32_t)J.lgblock);
}
// MODE 0: the plain chain (one unit = one shard), 1: the tiles' first parse, 2: a sweep — three kernels, so that
// the plain chain does not carry the registers of the other two.
template <int MODE>
DEV void chain_round(const JobParams& J, const ShardDesc* shards, ShardState* states,
                     uint32_t nshards, const DeviceTables* T, const uint8_t* input, uint8_t* ws,
                     uint32_t wave_index, uint32_t* lds, const TileDesc* tiles, TileRec* trecs, uint32_t ntiles,
                     const ShardDesc* chunks = nullptr) {
  const int t = q_t();
  const uint32_t gpw = q_groups_per_wave(J);
  const uint32_t gi = (uint32_t)(wave_lane() >> 4);
  constexpr bool tiled = MODE != 0, sweep = MODE == 2;
  const uint32_t unit = wave_index * gpw + gi;
  bool alive = gi < gpw && unit < (tiled ? ntiles : nshards);
  uint32_t shard = unit, tt = 0;
  if (tiled) { shard = tiles[alive ? unit : 0u].shard; tt = tiles[alive ? unit : 0u].t; }
  const bool writer = alive && t == 0;
  constexpr int NPOS = 4;
  const uint32_t htl = hasher_htl(J.hasher_type);
  const ShardDesc& D = shards[alive ? shard : 0];
  const ShardState* S0 = &states[alive ? shard : 0];
  uint32_t* scratch = lds + gi * C_GROUP_LDS_WORDS;
  if (t == 0) scratch[16] = 0xFFFFFFFFu;               // no carried store count yet (c_search_exact)
  const int kpos = t >> 2, idc = t & 3;   // this lane's probe: position P0 + kpos, cache entry idc
  // a shard of one tile is parsed the plain way even in a tiled job
  const bool tile_mode = tiled && D.ntiles > 1u;
  const bool last_tile = !tile_mode || tt + 1u == D.ntiles;
  TileRec* TR = trecs + (tile_mode && alive ? unit : 0u);
  if (tiled && !tile_mode && sweep) alive = false;     // (nothing to sweep)

  CShard C;
  QShard& g = C.g;
  g.data = input + D.in_off;
  g.table = nullptr;
  g.nums = nullptr;
  g.cmds = (Command*)(ws + D.cmds_off);
  g.descs = shards;
  g.wsb = ws;
  g.shard = alive ? shard : 0u;
  g.stream_offset = D.stream_offset;
  g.raw_cmds = 1;
  regs_load(g.r, S0);
  for (int i = 0; i < 4; ++i) g.dc[i] = S0->dist_cache[i];
  g.dict_lookups = S0->dict_lookups;
  g.dict_matches = S0->dict_matches;
  g.blk_flags = g.blk_bytes = g.blk_pos = 0;
  g.position = g.pos_end = g.store_end = g.insert_length = g.apply_random_heuristics = 0;
  g.sr_len = g.sr_dist = 0; g.sr_score = K_MIN_SCORE; g.sr_delta = 0; g.delayed = 0;
  g.st_first = g.st_count = 0; g.st_stride = 1;
  g.st_x = 0; g.st_x_valid = 0;
  g.n32.q[0] = g.n32.q[1] = g.n32.q[2] = g.n32.q[3] = 0;
  g.n32_pos = 0xFFFFFFFFu;
  g.status = 0;
  g.stat_searches = 0;
  g.role = 0;
  g.pf_val = g.pf_acc = 0;
  for (int i = 0; i < 12; ++i) g.prof[i] = 0;
  g.state = (alive && !S0->done && !S0->mb_valid && !S0->error) ? Q_PRE : Q_DONE;
  C.geo = ix_geom(J, D);
  if (t == 0 && gi < gpw) { scratch[19] = C_NO_OLDER; scratch[23] = 0; }
  IxLayout L;
  uint8_t* evb;                                        // the event bitmap (sweeps)
  if (tiled && (J.flags & JOB_FLAG_STREAMT) != 0) {
    // a stream: the index chunk the tile's positions are searched from, addressed by stream position
    const uint32_t lo = alive ? tile_lo(C.geo.first, tt, J.tile_log2) : 0u;
    const ShardDesc& K = chunks[umin(lo >> J.chunk_log2, J.nchunks - 1u)];
    ix_layout(K.len, J.ix_slices, J.ix_nb_log2, &L);
    uint8_t* ixb = ws + K.ix_off;
    C.res = (uint64_t*)(ixb + L.res) - K.ix_base;
    C.srt = (const uint32_t*)(ixb + L.srt);
    C.ibase = K.ix_base;
    if (J.chunk_log2 < (uint32_t)J.lgwin && K.ix_base != 0u) {
      // (chunks of half a window: chunk cj - 1 holds what lies between the window's far end and this chunk's base)
      const uint32_t cj = umin(lo >> J.chunk_log2, J.nchunks - 1u);
      const ShardDesc& K2 = chunks[cj - 1u];
      if (t == 0) { scratch[19] = K2.ix_base; scratch[20] = (uint32_t)K2.ix_off; scratch[21] = (uint32_t)(K2.ix_off >> 32); scratch[22] = K2.len; }
    }
    C.skip = ws + J.sbm_off;
    evb = ws + J.sbm_off + 2u * J.sbm_stride;
    if (D.len > J.ring_mask) g.ring_mask = J.ring_mask;
    g.ring_mask_stream = 1u;
  } else {
    ix_layout(D.len, J.ix_slices, J.ix_nb_log2, &L);
    uint8_t* ixb = ws + D.ix_off;
    C.res = (uint64_t*)(ixb + L.res);
    C.srt = (const uint32_t*)(ixb + L.srt);
    C.ibase = 0;
    C.skip = ixb + L.skip;
    evb = ixb + L.ev;
  }
  C.frontier = S0->ix_frontier;
  C.nslow = 0;
  C.tile_lo = 0;
  C.tile_hi = D.len;
  C.mode = tiled ? C_TILED : 0u;
  bool force_slow = (J.flags & JOB_FLAG_FORCE_SLOW) != 0;

  // ---- a tile of a tiled job: where it starts from (wave operations stay outside the per-group branches) ----
  uint32_t warm = 0;                                   // 1: the group is in the warm-up of a speculative start
  const bool stream = tiled && (J.flags & JOB_FLAG_STREAMT) != 0;
  // the gate hypothesis of the shard's tiles t > 0 (enc_types.h: TILE_GATE_OPEN), and the counter values that stand
  // for it at a tile's start: closed = 256 lookups without a match; open for good = a match count no lookup count reaches
  const uint32_t hyp = (tile_mode && alive && tt != 0u) ? TR->hyp : 0u;
  const uint32_t gate_l0 = hyp == 2u ? TR->in_l : hyp == 1u ? 0u : 256u;
  const uint32_t gate_m0 = hyp == 2u ? TR->in_m : hyp == 1u ? 0x40000000u : 0u;
  bool cut_in = false;                                 // a stream's tile that begins a meta-block: no ExtendLastCommand
  CReplay R;
  R.old = g.cmds; R.oi = R.on = 0; R.obnd = 0; R.changed = 0; R.next_ev = 0;
  R.odc[0] = R.odc[1] = R.odc[2] = R.odc[3] = 0;
  R.ev = evb;
  {
    if (tile_mode) {
      C.tile_lo = tile_lo(C.geo.first, tt, J.tile_log2);
      C.tile_hi = tile_hi(D.len, C.geo.first, tt, J.tile_log2);
    }
    const uint32_t ev_w0 = (C.tile_lo - umin(C.tile_lo, C.geo.first)) >> 5, ev_w1 = (C.tile_hi - C.geo.first + 31u) >> 5;
    bool run = alive && tile_mode && !S0->error;
    if (!sweep && run && (TR->flags & TILE_RAN) != 0) run = false;       // (a later launch parses the tiles sent back only, k_tile.h)
    if (sweep && run && (TR->flags & TILE_RAN) == 0) run = false;        // (... and a sweep leaves those alone)
    uint32_t any = 0;
    if (sweep && run) {
      // only tiles with something pending run: a replaced in-state or an event among their positions
      if (TR->flags & TILE_START_EVENT) any = 1;
      const uint32_t* evw = (const uint32_t*)R.ev;
      for (uint32_t i = ev_w0 + (uint32_t)t; i < ev_w1; i += 16u) any |= evw[i];
    }
    any = q_or(any);
    if (sweep) run = run && any != 0 && (trecs[D.tile_base].flags & TILE_BAD) == 0;
    int32_t used_dc[4] = {0, 0, 0, 0};
    uint32_t used_insert = 0, used_ext = 0;
    if (tile_mode && stream) g.no_cut = last_tile ? 0u : 1u;     // (cuts are k_stream_cuts' to find, k_tile.h)
    if (tile_mode) {
      g.lim_len = C.tile_hi;
      g.lim_op = last_tile ? D.final_op : 0u;
      g.lim_cmd_cap = tile_slot_cmds(J.tile_log2, (uint32_t)J.lgblock);
      const uint32_t oldbuf = sweep ? TR->buf : 1u;
      g.cmds = c_tile_slot(ws, D, J, oldbuf ^ 1u, tt);
      if (!sweep && (J.flags & JOB_FLAG_VIEWALL) != 0) { C.mode |= C_VIEW_ALL; force_slow = true; }     // (every search exact, as in a sweep)
      if (sweep) {
        C.mode |= C_VIEW_ALL;
        force_slow = true;
        R.old = c_tile_slot(ws, D, J, oldbuf, tt);
        R.oi = tt == 0 ? 0u : 1u;
        R.on = R.oi + (run ? TR->out_ncmds : 0u);
        if (run && R.on > R.oi && R.old[R.on - 1u].cmd_prefix != CMD_RAW) --R.on;     // (the trailing insert-only command)
        for (int i = 0; i < 4; ++i) used_dc[i] = TR->used_dc[i];
        used_insert = TR->used_insert; used_ext = TR->used_ext;
        cut_in = stream && run && TR->cut != 0;
      }
      alive = run;
      // every tile's parse begins from a state of its own making: nothing is taken from ShardState
      ShardState init;
      init_shard_state(J, D, &init);
      regs_load(g.r, &init);
      for (int i = 0; i < 4; ++i) g.dc[i] = init.dist_cache[i];
      g.dict_lookups = g.dict_matches = 0;
      C.frontier = 0;
      g.state = run ? Q_PRE : Q_DONE;
      if (tt == 0) {
        R.obnd = C.geo.first;
        for (int i = 0; i < 4; ++i) R.odc[i] = init.dist_cache[i];
      } else {
        const uint32_t B = C.tile_lo;
        g.r.flint = -2;
        // (a stream's tile: nothing of its own block is "flushed" yet, so the next block always fits; the last one
        //  ends the stream's last meta-block, whose size the tile need not know)
        g.r.last_flush_pos = stream ? (last_tile ? 0u : B) : C.geo.first;
        g.r.last_bytes = g.r.last_bytes_bits = 0;
        g.dict_lookups = gate_l0;                      // the gate is taken as closed (hash.h:186) or as open for good;
        g.dict_matches = gate_m0;                      //   k_tile_verify / k_stream_cuts check
        if (!sweep) {
          // round 0: the state at B is what a parse of the `tile_warm` bytes before it arrives with
          const uint32_t S = B - umin(J.tile_warm, 1u << (J.lgblock - 1));
          g.dc[0] = 4; g.dc[1] = 11; g.dc[2] = 15; g.dc[3] = 16;
          g.r.input_pos = B;
          g.r.last_processed_pos = S;
          g.blk_flags = 0; g.blk_bytes = B - S; g.blk_pos = S; g.pos_end = B;
          C.frontier = S;
          if (run) { g.state = Q_SETUP; warm = 1; }
        } else {
          // sweeps: the state k_tile_verify put into the record (the true one as far as it is known)
          for (int i = 0; i < 4; ++i) { R.odc[i] = used_dc[i]; g.dc[i] = TR->in_dc[i]; }
          R.obnd = B + used_ext - used_insert;
          g.r.last_insert_len = TR->in_insert;
          g.r.input_pos = g.r.last_processed_pos = B;
          g.r.ncmds = 1;
          C.frontier = B;
        }
      }
    }
    wave_sync();
    if (tile_mode && sweep && tt != 0u && alive && t == 0) {
      Command gh;
      gh.insert_len = 0; gh.copy_len = TR->in_copy_len; gh.dist_extra = TR->in_code; gh.cmd_prefix = CMD_RAW; gh.dist_prefix = 0;
      g.cmds[0] = gh;
      for (int i = 0; i < 4; ++i) TR->used_dc[i] = TR->in_dc[i];
      TR->used_insert = TR->in_insert;
      TR->used_cut = cut_in ? 1u : 0u;
    }
    wave_sync();
    if (sweep) {
      const uint32_t ne = c_next_event(R.ev, C.geo.first, alive, umax(C.tile_lo, C.geo.first), C.tile_hi);
      R.next_ev = alive ? ne : 0u;
    }
  }
  const bool participated = g.state != Q_DONE;

  uint32_t nsteps = 0;
  while (wave_any(g.state != Q_DONE)) {
    SIM_COUNT(7, 1);                                   // chain steps (wave level)
    uint64_t qt = QP_NOW();
    if (sweep) c_group_replay(J, C, R, alive, htl, scratch);
    else {
      // (tiles parsed again in a later pass — JOB_FLAG_VIEWALL — take the generic step only, like a sweep: it re-settles
      //  every bit of what it passes over)
#if defined(BROTLI_AMD_SIMT_SIM)
      if (!getenv("SIM_NOFAST") && !(J.flags & JOB_FLAG_VIEWALL)) c_group_fast(J, T, C, alive, scratch, nsteps);   // (test knob: generic steps only)
#else
      if (!(J.flags & JOB_FLAG_VIEWALL)) c_group_fast(J, T, C, alive, scratch, nsteps);
#endif
    }
    if (g.state == Q_PRE) q_driver_pre(J, g);
    if (wave_any(g.state == Q_SETUP)) {
      const bool su = g.state == Q_SETUP;
      // StitchToPreviousBlock (..64_simd_inc.h:139-151) stores the last three positions of the
      // previous block
      if (wave_any(su && (g.blk_flags & QBLK_STITCH)))
        c_stored(J, C, su && (g.blk_flags & QBLK_STITCH), g.blk_pos - 3u, g.blk_pos);
      if (cut_in && su && g.blk_pos == C.tile_lo) g.blk_flags &= ~QBLK_EXTEND;      // (num_commands_ is 0 behind a cut, encode.c:1103)
      const bool first_blk = tile_mode && tt != 0u && su && g.blk_pos == C.tile_lo && (g.blk_flags & QBLK_EXTEND) != 0;
      const uint32_t ghost_len = first_blk ? (g.cmds[0].copy_len & 0x1FFFFFFu) : 0u;
      wave_sync();
      q_setup_extend(J, g, su);
      // a tile's first block: what ExtendLastCommand added to the last command of the tile before it
      const uint32_t ext_len = first_blk ? (g.cmds[0].copy_len & 0x1FFFFFFu) - ghost_len : 0u;
      if (tile_mode && tt != 0u && su && g.blk_pos == C.tile_lo && writer) { TR->in_ext = ext_len; TR->used_ext = ext_len; }
    }
    // block finished? (loop guard of CreateBackwardReferences, :44 and :239-241)
    if (g.state == Q_SEARCH && !(g.position + htl < g.pos_end)) {
      g.insert_length += g.pos_end - g.position;
      g.r.last_insert_len = g.insert_length;
      g.state = Q_POST;
    }
    const bool want = g.state == Q_SEARCH || g.state == Q_LAZY;
    if (wave_any(want)) {
      const uint32_t P0 = g.position + (g.state == Q_LAZY ? 1u : 0u);
      // positions passed over since the last store are known now
      if (wave_any(want && C.frontier < P0)) c_mark_range(J, C, want && C.frontier < P0, C.frontier, P0, 0, 1);
      if (want) C.frontier = umax(C.frontier, P0);
      wave_sync();

      ++nsteps;
      QP_ADD(g, 0, qt);
      // ---- evaluation: lane (kpos, idc) ----
      CEval ce;
      if (sweep) {      // every search of a sweep is exact: nothing to evaluate from the index
        ce.e_flags = K_MIN_SCORE | ((want && P0 + (uint32_t)kpos + htl <= g.pos_end) ? 0xC0000000u : 0u);
        ce.e_len = ce.e_dist = 0;
      } else ce = c_evaluate(J, C, want, P0, kpos, idc, force_slow, htl);
      const uint32_t e_flags = ce.e_flags, e_len = ce.e_len, e_dist = ce.e_dist;
      QP_ADD(g, 1, qt);
      QP_ADD(g, 2, qt);
      // ---- the common transitions, all four positions in registers (:44-164) ----
      // SEARCH + hit -> LAZY; SEARCH + miss -> one more literal (only while neither the static
      // dictionary nor the literal spree can come into play); LAZY -> stay lazy or commit.
      uint32_t consumed = 0, sr_from = 0xFFu;
      bool commit = false, stop = !want;
      uint32_t fl0 = 0;
      for (int k = 0; k < NPOS; ++k) {
        if (k >= 2 && !wave_any(!stop)) break;                         // (most commands end at the second position)
        const uint32_t flk = q_bcast(e_flags, 4 * k);
        if (k == 0) fl0 = flk;
        const uint32_t sk = flk & 0x3FFFFFFFu;
        const bool usable = (flk >> 30) == 2u;                         // evaluated and decidable from the index
        if (!stop && !usable) stop = true;
        if (!stop) {
          if (g.state == Q_SEARCH) {
            if (!(g.position + htl < g.pos_end)) stop = true;
            else if (sk > K_MIN_SCORE) {
              g.sr_score = sk; sr_from = (uint32_t)k; g.delayed = 0; g.sr_delta = 0; g.state = Q_LAZY;
              ++consumed;
            } else if (g.dict_matches < (g.dict_lookups >> 7) && g.position + 1u <= g.apply_random_heuristics) {
              ++g.insert_length; ++g.position; ++consumed;
            } else stop = true;
          } else if (sk == K_MIN_SCORE && !(g.dict_matches < (g.dict_lookups >> 7))) {
            stop = true;   // Q_LAZY, nothing found: the static dictionary is asked next (generic path)
          } else {         // Q_LAZY
            ++consumed;
            if (sk >= g.sr_score + 175u) {
              ++g.position; ++g.insert_length;
              g.sr_score = sk; sr_from = (uint32_t)k; g.sr_delta = 0;
              if (!(++g.delayed < 4u && g.position + htl < g.pos_end)) { commit = true; stop = true; g.cmd_flags |= CMDF_NOPROBE; }
            } else { commit = true; stop = true; }
          }
        }
      }
      QP_ADD(g, 3, qt);
      if (want && consumed != 0) { C.frontier = P0 + consumed; g.stat_searches += consumed; }
      // the pending match sits in the quad that evaluated it
      if (wave_any(sr_from != 0xFFu)) {
        const int src = q_base() | (int)((sr_from & 3u) << 2);
        const uint32_t l = wave_shfl(e_len, src), d = wave_shfl(e_dist, src);
        if (sr_from != 0xFFu) { g.sr_len = l; g.sr_dist = d; }
      }
      bool committed = commit;
      if (wave_any(commit)) q_commit(J, g, commit, htl);

      QP_ADD(g, 4, qt);
      // ---- everything else, one position: exact search, dictionary, spree, block end ----
      const bool gen = want && consumed == 0;
      if (wave_any(gen)) {
        QResult cur;
        cur.len = q_bcast(e_len, 0);
        cur.distance = q_bcast(e_dist, 0);
        cur.score = fl0 & 0x3FFFFFFFu;
        cur.delta = 0;
        bool take = gen && (fl0 & 0x80000000u) != 0;
        if (take) {
          if (g.state == Q_SEARCH) take = g.position == P0 && g.position + htl < g.pos_end;
          else take = g.state == Q_LAZY && g.position + 1u == P0;
        }
        if (gen && !take) { g.status |= QST_ERROR | QST_DONE; g.state = Q_DONE; }   // cannot happen: fail, do not spin
        const bool exact = take && (fl0 & 0x40000000u) != 0;
        if (wave_any(exact)) {
          const QResult sx = c_search_exact(J, C, exact, P0, scratch);
          if (exact) { cur = sx; ++C.nslow; }
        }
        if ((sweep || (J.flags & JOB_FLAG_VIEWALL) != 0) && wave_any(take)) c_clear_range(C, take, P0, P0 + 1u);
        if (take) C.frontier = P0 + 1u;          // FindLongestMatch stores the position it searched
        // static dictionary when nothing was found (hash.h:179-202)
        q_dict_search(J, T, g, take && cur.score == K_MIN_SCORE, P0, g.pos_end - P0, cur);
        if (take) g.stat_searches++;
        committed = q_transition(J, g, take, cur, htl) || committed;
      }
      QP_ADD(g, 5, qt);
      if (committed) ++R.changed;
      // what the step stored: the copied range (StoreRange) or the literal spree
      if (wave_any(want && g.st_count != 0)) {
        const bool st = want && g.st_count != 0;
        if (wave_any(st && g.st_stride == 1u))
          c_stored(J, C, st && g.st_stride == 1u, g.st_first, g.st_first + g.st_count);
        if (wave_any(st && g.st_stride != 1u)) {
          const bool sp = st && g.st_stride != 1u;
          c_stored(J, C, sp, g.st_first, g.st_first);
          c_mark_range(J, C, sp, g.st_first, g.st_first + g.st_count * g.st_stride, g.st_first, g.st_stride);
          if (sp) C.frontier = g.st_first + g.st_count * g.st_stride;
        }
        if (st) g.st_count = 0;
      }
      QP_ADD(g, 6, qt);
    }
    {
      // ---- a warm-up that has reached the tile's first block boundary: this is the state the tile starts from ----
      const bool wpost = g.state == Q_POST && warm != 0;
      if (wave_any(wpost)) {
        Command gh;
        gh.insert_len = 0; gh.copy_len = 0; gh.dist_extra = 0; gh.cmd_prefix = CMD_RAW; gh.dist_prefix = 0;
        if (wpost && g.r.ncmds != 0) gh = g.cmds[g.r.ncmds - 1u];
        wave_sync();
        if (wpost && writer) {
          gh.copy_len &= 0x1FFFFFFu;
          g.cmds[0] = gh;
          for (int i = 0; i < 4; ++i) { TR->in_dc[i] = g.dc[i]; TR->used_dc[i] = g.dc[i]; }
          TR->in_insert = TR->used_insert = g.r.last_insert_len;
          TR->in_copy_len = gh.copy_len;
          TR->in_code = gh.dist_extra;
          TR->in_ext = TR->used_ext = 0;
          TR->used_cut = 0;
        }
        wave_sync();
        if (wpost) {
          g.r.ncmds = 1;
          g.r.nlits = 0;
          g.r.input_pos = g.r.last_processed_pos = C.tile_lo;
          g.dict_lookups = gate_l0; g.dict_matches = gate_m0;      // (the warm-up's lookups are not the tile's)
          g.dict_mark_l = g.dict_lookups; g.dict_mark_m = g.dict_matches;
          g.cmd_flags = 0;
          C.frontier = C.tile_lo;
          warm = 0;
          g.state = Q_PRE;
        }
      }
    }
#if defined(BROTLI_AMD_SIMT_SIM)
    if (g.state == Q_POST && t == 0 && alive && getenv("SIM_GATE_LOG"))
      fprintf(stderr, "GATE mode %d tile %u block end %u: lookups %u matches %u\n", MODE, tt, g.pos_end, g.dict_lookups, g.dict_matches);
#endif
    if (g.state == Q_POST) q_driver_post(J, g, writer);
    // a tile that met a counter wrap gives up at once: the shard goes the plain way anyway, and the counted
    // searches (c_search_exact walks a whole key run) are what makes such shards slow
    if (tiled && (C.mode & C_BAD) != 0 && g.state != Q_DONE) { g.status |= QST_ERROR | QST_DONE; g.state = Q_DONE; }
    QP_ADD(g, 7, qt);
  }

  wave_sync();
  {
    // ---- a tile's result: the state it ended with (k_tile.h takes it from here) ----
    const bool tp = tile_mode && participated;
    // the storable positions behind the last store stay unstored — but for the last three, which the next
    // tile's first block stitches in (..64_simd_inc.h:139-151; ix_storable knows when it does)
    const bool m = tp && !last_tile && !(g.status & QST_ERROR) && C.frontier + 3u < C.tile_hi;
    if (wave_any(m)) c_mark_range(J, C, m, C.frontier, C.tile_hi - 3u, 0, 1);
    const uint32_t base = tt == 0 ? 0u : 1u;
    Command lastc;
    lastc.insert_len = lastc.copy_len = lastc.dist_extra = 0; lastc.cmd_prefix = lastc.dist_prefix = 0;
    if (tp && g.r.ncmds > base) lastc = g.cmds[g.r.ncmds - 1u];
    wave_sync();
    if (tp && t == 0) {
      for (int i = 0; i < 4; ++i) TR->out_dc[i] = g.dc[i];
      TR->out_insert = g.r.last_insert_len;
      TR->out_copy_len = lastc.cmd_prefix == CMD_RAW ? (lastc.copy_len & 0x1FFFFFFu) : 0u;
      TR->out_code = lastc.dist_extra;
      TR->out_ncmds = g.r.ncmds - base;
      TR->out_nlits = g.r.nlits;
      TR->out_gate = (g.dict_matches < (g.dict_lookups >> 7)) ? 1u : 0u;
      TR->dlookups = g.dict_lookups - (tt == 0 ? 0u : gate_l0);
      TR->dmatches = g.dict_matches - (tt == 0 ? 0u : gate_m0);
      TR->out_lpp = g.r.last_processed_pos;
      TR->out_mb = ((g.status & QST_HAVE_MB) ? 1u : 0u) | ((g.blk_flags & QBLK_LAST) ? 2u : 0u) |
                   ((g.blk_flags & QBLK_FLUSH) ? ((g.blk_flags & QBLK_NOSEAL) ? 8u : 4u) : 0u);
      TR->buf = sweep ? (TR->buf ^ 1u) : 0u;
      uint32_t fl = (TR->flags | TILE_RAN) & ~(TILE_START_EVENT | TILE_CHANGED);
      if ((C.mode & C_BAD) != 0) fl |= TILE_BAD | TILE_WHY_WRAP;
      if ((g.status & QST_ERROR) != 0) fl |= TILE_BAD | TILE_WHY_ERROR;
      if (last_tile && !(g.status & QST_HAVE_MB)) fl |= TILE_BAD | TILE_WHY_NO_MB;
      if (R.changed != 0) fl |= TILE_CHANGED;
      TR->flags = fl;
      if (tt == 0) {
        // what the shard's first blocks left behind (the flint bytes' meta-block, the stream header bits)
        ShardState* S = &states[shard];
        regs_save(g.r, S);
        S->dict_lookups = g.dict_lookups;
        S->dict_matches = g.dict_matches;
        S->done = 0; S->mb_valid = 0;
      }
    }
    if (tp && sweep) {
      // the events of the tile are dealt with
      uint32_t* evw = (uint32_t*)evb;
      const uint32_t w0 = (C.tile_lo - umin(C.tile_lo, C.geo.first)) >> 5, w1 = (C.tile_hi - C.geo.first + 31u) >> 5;
      for (uint32_t i = w0 + (uint32_t)t; i < w1; i += 16u) evw[i] = 0;
    }
    wave_sync();
  }
  if (!tile_mode && writer && participated) {
    ShardState* S = &states[shard];
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
    S->stat_pairs += nsteps;
    S->stat_b_used = g.pf_acc;
#if defined(Q_PROFILE)
    for (int i = 0; i < 12; ++i) S->prof[i] += g.prof[i];
#endif
    S->ix_frontier = C.frontier;
    S->ix_slow += C.nslow;
  }
  wave_sync();
}

#endif  // BROTLI_AMD_CSRC_K_CHAIN_H_
