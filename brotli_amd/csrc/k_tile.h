// brotli_amd/csrc/k_tile.h — tiled quality-5 jobs (JOB_FLAG_TILED): what stands between the tiles' parses.
//
// The reference's parse of a shard is one dependency chain (c/enc/backward_references_inc.h:44-238): the distance
// cache, the pending literals and the last command cross every block boundary (c/enc/encode.c:905-971, 1103-1139),
// and which positions the hasher holds depends on every decision before (hash_longest_match64_simd_inc.h:114-137).
// k_chain parses the tiles of a shard all at once all the same: each tile from a state its own warm-up arrived at,
// with the unstored positions of OTHER tiles taken as stored.  Both guesses are checked here and, where they do not
// hold, turned into work for the next sweep of k_chain (which parses again only around such places and re-joins its
// earlier result):
//   k_tile_verify  — per shard, tile after tile: does the state a tile started from equal the state the tile before
//                    it ended with?  If not the true state goes into the tile's record (TILE_START_EVENT).  Totals,
//                    command offsets, and the conditions under which a shard cannot be done in tiles at all
//                    (TILE_BAD: the static-dictionary gate still open behind the first tile, a meta-block cut inside
//                    the shard, a tile without commands, a counter wrap).
//   k_tile_events  — every position whose "unstored" bit changed since the last pass (the first pass: every unstored
//                    position) invalidates the searches of the <= 16 positions behind it in its key run, whichever
//                    tile they are in: their bits go into the event bitmap.
//   k_tile_finish  — the tiles' commands move into the shard's command array (encoded, what k_cmd_encode does
//                    for the plain chain), a tile's last command gets the bytes the next tile's ExtendLastCommand gave
//                    it, and the shard's state is put together for k_build / k_store.
// The loop (hip_layer.hip): chain, { verify, events, stop if nothing is pending, sweep }, finish.  It ends with every
// search consistent with the final bitmap and every join equal — the parse the reference makes, since a decision
// depends on earlier positions only.
#ifndef BROTLI_AMD_CSRC_K_TILE_H_
#define BROTLI_AMD_CSRC_K_TILE_H_

#include "k_chain.h"

// counters[]: [2] tiles with a start event, [3] changed skip bits, [4] shards that left the tiled path
#define TILE_CNT_START 2
#define TILE_CNT_FLIPS 3
#define TILE_CNT_BAD 4

// grid = nshards, block = 64: lane 0 walks the shard's tiles.
DEV void tile_verify(const JobParams& J, const ShardDesc& D, ShardState* S, TileRec* trecs, uint32_t* counters) {
  if (D.ntiles <= 1u || wave_lane() != 0) return;
  TileRec* R = trecs + D.tile_base;
  if (R[0].flags & TILE_BAD) return;                    // (decided in an earlier pass)
  uint32_t why = 0;
  uint32_t ncmds = 0, nlits = 0, starts = 0;
  // a shard most of whose searches the tiles would have to do twice is better off on the plain chain
  if (R[0].nflips > D.len / 32u + 64u) why |= TILE_WHY_EVENTS;
  R[0].nflips = 0;
  for (uint32_t t = 0; t < D.ntiles; ++t) {
    TileRec& c = R[t];
    if (c.flags & TILE_BAD) why |= TILE_WHY_TILE;
    if (!(c.flags & TILE_RAN)) why |= TILE_WHY_NOT_RUN;
    c.cmd_off = ncmds;
    ncmds += c.out_ncmds;
    nlits += c.out_nlits;
    if (t + 1u == D.ntiles) break;
    // what the next tile has to start from
    TileRec& n = R[t + 1u];
    if (c.out_ncmds == 0u) why |= TILE_WHY_NO_CMD;          // (no command ExtendLastCommand could lengthen)
    if (c.out_gate == 0u) why |= TILE_WHY_GATE;             // (the dictionary still consulted)
    const bool same_cmd = c.out_insert != 0u || (n.in_copy_len == c.out_copy_len && n.in_code == c.out_code);
    const bool same = n.in_dc[0] == c.out_dc[0] && n.in_dc[1] == c.out_dc[1] && n.in_dc[2] == c.out_dc[2] &&
                      n.in_dc[3] == c.out_dc[3] && n.in_insert == c.out_insert && same_cmd;
    if (!same) {
      for (int i = 0; i < 4; ++i) n.in_dc[i] = c.out_dc[i];
      n.in_insert = c.out_insert;
      n.in_copy_len = c.out_copy_len;
      n.in_code = c.out_code;
      n.flags |= TILE_START_EVENT;
      ++starts;
    }
  }
  // a meta-block cut inside the shard (encode.c:1141-1166) is not something the tiles know about
  if (nlits >= J.max_literals || ncmds >= J.max_commands) why |= TILE_WHY_CUT;
  if (S->error != 0) why |= TILE_WHY_ERROR;
  const bool bad = why != 0;
#if defined(BROTLI_AMD_SIMT_SIM)
  if (getenv("SIM_TILE_LOG")) {
    for (uint32_t t = 0; t < D.ntiles; ++t)
      fprintf(stderr, "  tile %u: flags %x ncmds %u nlits %u gate %u in(dc %d %d %d %d ins %u cl %u code %u ext %u) out(dc %d %d %d %d ins %u cl %u code %u) buf %u\n", t, R[t].flags,
              R[t].out_ncmds, R[t].out_nlits, R[t].out_gate, R[t].in_dc[0], R[t].in_dc[1], R[t].in_dc[2], R[t].in_dc[3], R[t].in_insert, R[t].in_copy_len, R[t].in_code, R[t].in_ext,
              R[t].out_dc[0], R[t].out_dc[1], R[t].out_dc[2], R[t].out_dc[3], R[t].out_insert, R[t].out_copy_len, R[t].out_code, R[t].buf);
    if (bad) fprintf(stderr, "  -> shard leaves the tiled path (nlits %u ncmds %u error %u)\n", nlits, ncmds, S->error);
  }
#endif
  if (bad) {
    R[0].flags |= TILE_BAD | why;
    glb_atomic_add(&counters[TILE_CNT_BAD], 1u);
  } else if (starts != 0) glb_atomic_add(&counters[TILE_CNT_START], starts);
}

// grid = nshards * ix_slices, block = 64: the slice's words of the bitmap.
DEV void tile_events(const JobParams& J, const ShardDesc& D, const uint8_t* input, uint8_t* ws, TileRec* trecs, uint32_t w, uint32_t* counters) {
  if (D.ntiles <= 1u || (trecs[D.tile_base].flags & TILE_BAD)) return;
  const uint32_t lane = (uint32_t)wave_lane();
  const uint32_t first = D.stream_offset != 0 ? 2u : 0u;
  IxLayout L;
  ix_layout(D.len, J.ix_slices, J.ix_nb_log2, &L);
  uint8_t* base = ws + D.ix_off;
  const uint32_t* skip = (const uint32_t*)(base + L.skip);
  uint32_t* prev = (uint32_t*)(base + L.skip_prev);
  uint32_t* ev = (uint32_t*)(base + L.ev);
  const uint32_t* srt = (const uint32_t*)(base + L.srt);
  const uint64_t* res = (const uint64_t*)(base + L.res);
  const uint8_t* data = input + D.in_off;
  const uint32_t total = ((const uint32_t*)(base + L.cnt))[J.ix_slices << J.ix_nb_log2];    // sorted entries of the shard
  const uint32_t per = ix_slice_len(D.len, J.ix_slices);
  const uint32_t w_lo = (w * per) / 32u, w_hi = umin(((w + 1u) * per) / 32u, (D.len + 31u) / 32u);
  uint32_t flips = 0;
  for (uint32_t i = w_lo + lane; i < w_hi; i += 64u) {
    const uint32_t cur = skip[i];
    uint32_t diff = cur ^ prev[i];
    if (diff == 0) continue;
    prev[i] = cur;
    flips += (uint32_t)__builtin_popcount(diff);
    for (; diff != 0; diff &= diff - 1u) {
      // x changed: every later position of its key run whose ring — the 16 nearest STORED predecessors,
      // hash_longest_match64_simd_inc.h:246-262 — reaches back to x has to be searched again: the successors of x up
      // to the 16th one that is stored (behind a run of unstored positions that is more than 16 entries away)
      const uint32_t x = first + i * 32u + (uint32_t)dev_ctz32(diff);
      const uint32_t s = (uint32_t)(res[x] >> 32) & 0xFFFFFFu;
      const uint32_t key = hash_pos(ld64(data + x), J.hasher_type, J.bucket_bits).key;
      // (the first pass: a tile's own parse already knew what it had left unstored itself — only the searches
      //  of LATER tiles went without it)
      const uint32_t xt = (J.flags & JOB_FLAG_SWEEP) ? 0xFFFFFFFFu : (x - first) >> J.tile_log2;
      uint32_t stored = 0;
      for (uint32_t j = s + 1u; j < total && stored < 16u; ++j) {
        const uint32_t q = srt[j] & 0xFFFFFFu;
        if (hash_pos(ld64(data + q), J.hasher_type, J.bucket_bits).key != key) break;
        const uint32_t b = q - first;
        if ((b >> J.tile_log2) != xt) glb_atomic_or(&ev[b >> 5], 1u << (b & 31u));
        if (!((skip[b >> 5] >> (b & 31u)) & 1u)) ++stored;
      }
    }
  }
  if (flips != 0) {
    glb_atomic_add(&counters[TILE_CNT_FLIPS], flips);
    glb_atomic_add((uint32_t*)&trecs[D.tile_base].nflips, flips);
  }
}

// grid = ntiles, block = 64: a tile's commands to their place in the shard's array, encoded; tile 0's wave also
// completes the shard's state.  A shard of one tile was parsed by the plain chain into the array: encoded in place.
DEV void tile_finish(const JobParams& J, const ShardDesc& D, ShardState* S, uint8_t* ws, const TileRec* trecs, uint32_t tt) {
  const uint32_t lane = (uint32_t)wave_lane();
  Command* dst = (Command*)(ws + D.cmds_off);
  if (D.ntiles <= 1u) {
    const uint32_t n = S->ncmds;
    for (uint32_t i = lane; i < n; i += 64u) {
      const Command c = dst[i];
      if (c.cmd_prefix != CMD_RAW) continue;
      const uint32_t m = c.copy_len >> 25;
      const int32_t delta = (int8_t)((uint8_t)(m | ((m & 0x40) << 1)));
      dst[i] = make_command(c.insert_len, c.copy_len & 0x1FFFFFFu, delta, c.dist_extra);
    }
    return;
  }
  const TileRec* R = trecs + D.tile_base;
  if (R[0].flags & TILE_BAD) return;
  const TileRec& r = R[tt];
  const Command* src = c_tile_slot(ws, D, J, r.buf, tt) + (tt == 0 ? 0u : 1u);
  const uint32_t n = r.out_ncmds;
  const uint32_t ext = tt + 1u < D.ntiles ? R[tt + 1u].in_ext : 0u;
  for (uint32_t i = lane; i < n; i += 64u) {
    Command c = src[i];
    if (c.cmd_prefix == CMD_RAW) {
      uint32_t len = c.copy_len & 0x1FFFFFFu;
      if (i + 1u == n) len += ext;
      const uint32_t m = c.copy_len >> 25;
      const int32_t delta = (int8_t)((uint8_t)(m | ((m & 0x40) << 1)));
      c = make_command(c.insert_len, len, delta, c.dist_extra);
    }
    dst[r.cmd_off + i] = c;
  }
  if (tt == 0 && lane == 0) {
    const TileRec& z = R[D.ntiles - 1u];
    uint32_t ncmds = 0, nlits = 0;
    for (uint32_t k = 0; k < D.ntiles; ++k) { ncmds += R[k].out_ncmds; nlits += R[k].out_nlits; }
    S->input_pos = D.len;
    S->last_processed_pos = z.out_lpp;
    S->last_insert_len = z.out_insert;
    S->ncmds = ncmds;
    S->nlits = nlits;
    for (int i = 0; i < 4; ++i) S->dist_cache[i] = z.out_dc[i];
    S->done = 0;
    S->mb_valid = (z.out_mb & 1u) ? 1u : 0u;
    S->mb_start = S->last_flush_pos;
    S->mb_bytes = D.len - S->last_flush_pos;
    S->mb_is_last = (z.out_mb & 2u) ? 1u : 0u;
    S->mb_force_flush = (z.out_mb & 8u) ? 2u : (z.out_mb & 4u) ? 1u : 0u;
    S->mb_raw = 0;
    if (!(z.out_mb & 1u)) S->error = 3;
  }
}

#endif  // BROTLI_AMD_CSRC_K_TILE_H_
