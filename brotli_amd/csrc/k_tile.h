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
#include "mb_layout.h"

// counters[]: [2] tiles with a start event, [3] changed skip bits, [4] shards that left the tiled path
#define TILE_CNT_START 2
#define TILE_CNT_FLIPS 3
#define TILE_CNT_BAD 4

// ---- the gate hypothesis (TILE_GATE_OPEN) ---------------------------------------------------------------------------
// Tiles t > 0 are first parsed with the static dictionary's gate taken as closed — right for everything but text
// the dictionary keeps matching (real English), where tile 0 — the one that starts from the true counters — ends with
// the gate open.  Then the shard's other tiles start over with the gate taken as open for good: k_tile_restart
// (grid = nshards, block = 64; between the tiles' first launch and a second one that parses the restarted tiles
// only) un-runs them, k_tile_restart_clear (grid = units * ix_slices) wipes their bitmaps.
#define TILE_CNT_RESTART 7
#define TILE_CNT_PREFLIPS 10  // a stream: changed store bits counted BEFORE k_stream_events walks their successors (k_stream_flips)
// The successor walks of the event kernels (up to the 16th STORED successor of a changed position) are as long as the
// runs of unstored entries they cross: on a constant background with a rare token now and then — one key, nearly
// every position inside a clipped copy — that is the rest of the key run for every changed bit, 800 s for 8 MB before
// the "too many changes" rule could send the stream to the serial path (tools/gpu_fuzz_windows.py, seed 2).  Three
// things keep that from happening: a stream counts its changed bits before it walks (k_stream_flips); a changed,
// unstored position whose predecessor in the key run changed too does not walk at all (the predecessor's walk passes
// over it and ends where its own would: stream_events); and, as the net under both, the walks of one pass are charged
// to tile 0 of the shard in units of 4096 entries (TileRec::pad) — beyond 1024 entries per input byte the shard goes
// the serial way at once (a stream with 150 KB of noise and 200 KB of zeros in 770 KB walks 65 per byte and settles).
DEV bool tile_walk_over(TileRec* t0, uint32_t len, uint32_t* counters) {
  const uint32_t units = glb_atomic_add(&t0->pad, 1u) + 1u;
  if (units > len / 4u + 4096u) {
    if (!(t0->flags & TILE_BAD)) { glb_atomic_or(&t0->flags, TILE_BAD | TILE_WHY_EVENTS); glb_atomic_add(&counters[TILE_CNT_BAD], 1u); }
    return true;
  }
  return (t0->flags & TILE_BAD) != 0;
}
DEV void tile_unrun(TileRec& r, uint32_t hyp) {
  r.hyp = hyp;
  r.flags &= ~(TILE_RAN | TILE_START_EVENT | TILE_CHANGED);
}
DEV void tile_restart(const ShardDesc& D, TileRec* trecs, uint32_t* counters) {
  if (D.ntiles <= 1u) return;
  TileRec* R = trecs + D.tile_base;
  const uint32_t f0 = R[0].flags;
  if (!(f0 & TILE_RAN) || (f0 & (TILE_BAD | TILE_GATE_OPEN)) != 0 || R[0].out_gate != 0u) return;
  for (uint32_t t = 1u + (uint32_t)wave_lane(); t < D.ntiles; t += 64u) tile_unrun(R[t], 1u);
  wave_sync();
  if (wave_lane() == 0) {
    R[0].flags = f0 | TILE_GATE_OPEN;
    glb_atomic_add(&counters[TILE_CNT_RESTART], 1u);
  }
}
// grid = ntiles, block = 64: a tile that is to be parsed again from scratch forgets what it marked.
DEV void tile_restart_clear(const JobParams& J, const ShardDesc& D, uint8_t* ws, const TileRec* trecs, uint32_t tt) {
  if (D.ntiles <= 1u || tt == 0u || (trecs[D.tile_base + tt].flags & TILE_RAN) != 0) return;
  const uint32_t first = D.stream_offset != 0 ? 2u : 0u;
  uint32_t *skip, *prev, *ev;
  if (J.flags & JOB_FLAG_STREAMT) {
    skip = (uint32_t*)(ws + J.sbm_off); prev = (uint32_t*)(ws + J.sbm_off + J.sbm_stride); ev = (uint32_t*)(ws + J.sbm_off + 2u * J.sbm_stride);
  } else {
    IxLayout L;
    ix_layout(D.len, J.ix_slices, J.ix_nb_log2, &L);
    uint8_t* base = ws + D.ix_off;
    skip = (uint32_t*)(base + L.skip); prev = (uint32_t*)(base + L.skip_prev); ev = (uint32_t*)(base + L.ev);
  }
  const uint32_t w0 = (tile_lo(first, tt, J.tile_log2) - first) >> 5;
  const uint32_t w1 = tt + 1u == D.ntiles ? (D.len + 128u + 31u) >> 5 : (tile_lo(first, tt + 1u, J.tile_log2) - first) >> 5;
  // (in a later pass — JOB_FLAG_VIEWALL — the tile's new parse re-settles every bit of its range and the difference to
  //  `prev` is what the other tiles have to hear about: only its pending events go, which an exact parse has no use for)
  const bool late = (J.flags & JOB_FLAG_VIEWALL) != 0;
  for (uint32_t i = w0 + (uint32_t)wave_lane(); i < w1; i += 64u) {
    if (!late) { skip[i] = 0; prev[i] = 0; }
    ev[i] = 0;
  }
}
// The gate along a shard whose tile 0 ended with it open (one lane): tiles parsed as "open for good" are confirmed
// with the summed counters; the first one that cannot be (the gate may close in it) is parsed again from the exact
// counters, and once a tile has ended with the gate closed everything behind it is parsed as closed.  Returns the
// tiles sent back to be parsed again (the caller's loop goes on while there are any).
DEV uint32_t gate_walk(TileRec* R, uint32_t ntiles) {
  uint32_t gl = R[0].dlookups, gm = R[0].dmatches, again = 0;
  bool closed = false;
  for (uint32_t t = 1; t < ntiles; ++t) {
    TileRec& c = R[t];
    if (closed) {
      if (c.hyp != 0u) { tile_unrun(c, 0u); ++again; }
      continue;
    }
    if (!(c.flags & TILE_RAN)) break;                     // (on its way already: nothing behind it can be told yet)
    if (c.hyp == 1u) {
      if (gm >= ((gl + c.dlookups) >> 7)) { gl += c.dlookups; gm += c.dmatches; continue; }
      c.in_l = gl; c.in_m = gm;
      tile_unrun(c, 2u); ++again;
      break;
    }
    if (c.hyp == 2u) {
      if (c.in_l != gl || c.in_m != gm) { c.in_l = gl; c.in_m = gm; tile_unrun(c, 2u); ++again; break; }
      if (c.out_gate != 0u) closed = true; else { gl += c.dlookups; gm += c.dmatches; }
      continue;
    }
    tile_unrun(c, 1u); ++again;                           // parsed as closed, but it is not closed here
    break;
  }
  return again;
}

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
  R[0].pad = 0;                  // (tile_walk_over: the walks of the next pass)
  // the static dictionary's gate (hash.h:186): with tiles parsed as if it stayed open for good, the counters at a
  // tile's start are tile 0's plus the tiles' before — it cannot close inside a tile whose start has
  // matches >= (lookups + the tile's lookups) >> 7 (matches only grow, lookups end at that sum): gate_walk
  const bool gate_open = (R[0].flags & TILE_GATE_OPEN) != 0;
  if (gate_open) {
    const uint32_t again = gate_walk(R, D.ntiles);
    if (again != 0) { glb_atomic_add(&counters[TILE_CNT_RESTART], again); return; }     // (joins: once every tile has run)
  }
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
    // (a tile without a command: literals pending at its end — nobody asks for the last command then — or a block that
    //  ExtendLastCommand consumed whole, which hands on the command it was given, longer by what it added)
    const bool hollow = c.out_ncmds == 0u && c.out_insert == 0u;
    const uint32_t eff_copy_len = hollow ? c.in_copy_len + c.in_ext : c.out_copy_len;
    const uint32_t eff_code = hollow ? c.in_code : c.out_code;
    if (!gate_open && c.out_gate == 0u) why |= TILE_WHY_GATE;   // (cannot happen: tile 0 open restarts the shard, a closed gate stays closed)
    const bool same_cmd = c.out_insert != 0u || (n.in_copy_len == eff_copy_len && n.in_code == eff_code);
    const bool same = n.in_dc[0] == c.out_dc[0] && n.in_dc[1] == c.out_dc[1] && n.in_dc[2] == c.out_dc[2] &&
                      n.in_dc[3] == c.out_dc[3] && n.in_insert == c.out_insert && same_cmd;
    if (!same) {
      for (int i = 0; i < 4; ++i) n.in_dc[i] = c.out_dc[i];
      n.in_insert = c.out_insert;
      n.in_copy_len = eff_copy_len;
      n.in_code = eff_code;
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
  uint32_t flips = 0, walked = 0;
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
        if ((++walked & 4095u) == 0u && tile_walk_over(&trecs[D.tile_base], D.len, counters)) return;
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
  // what ExtendLastCommand added to the tile's last command at the next block — and at the blocks behind that one
  // as long as it consumed them whole
  uint32_t ext = 0;
  for (uint32_t u = tt + 1u; u < D.ntiles; ++u) {
    ext += R[u].in_ext;
    if (!(R[u].out_ncmds == 0u && R[u].out_insert == 0u)) break;
  }
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

// ---- a tiled stream (JOB_FLAG_STREAMT): one encoder instance longer than the window ------------------------------
// The stream is one shard whose tiles are its input blocks (tile_log2 == lgblock).  Besides the joins of the tiled
// shards above, a tile's parse depends on whether a meta-block was cut in front of it (encode.c:1141-1216: the
// pending literals become an insert-only command of the meta-block that ends, last_insert_len_ and num_commands_
// start from zero, so the block's ExtendLastCommand does nothing) — and where the cuts fall depends on the counts
// of all the tiles before: k_stream_cuts walks the tiles' counts (one wave), k_stream_verify compares every tile's
// assumed in-state (cut included) with what its predecessor left, sweeps repair, until nothing changes.  Then the
// meta-blocks are known: k_stream_cuts (finalize) describes each one as a shard of its own for k_build / k_store
// (its commands, its literals' context bytes, workspace carved out by its position), written as if it began at bit
// 0 of its own output, and k_stream_place shifts the results to their bit offsets in the stream.
#define TILE_CNT_RAW 5
#define TILE_CNT_NMB 6       // meta-blocks of the stream (k_stream_cuts, finalize)

DEV uint64_t st_al(uint64_t x) { return (x + 255u) & ~(uint64_t)255u; }
// One meta-block [tile s, tile e] of the stream as a shard description + state (lane 0 of the caller).
DEV void stream_emit_mb(const JobParams& J, const ShardDesc& D, const uint8_t* input, ShardDesc* md, ShardState* ms,
                        uint32_t m, uint32_t s, uint32_t e, uint32_t cmd_lo, uint32_t ncmds, uint32_t nlits, bool is_last) {
  const uint32_t start = tile_lo(0u, s, J.tile_log2), end = tile_hi(D.len, 0u, e, J.tile_log2), bytes = end - start;
  MbLayout ML;
  mb_layout(J.max_metablock_size, &ML);
  ShardDesc X;
  __builtin_memset(&X, 0, sizeof(X));
  X.in_off = D.in_off;
  X.len = bytes;
  X.final_op = 2u;
  X.cmd_cap = ncmds;
  X.cmds_off = D.cmds_off + 16ull * cmd_lo;
  X.lits_off = st_al(D.lits_off + 2ull * start + 512ull * m);
  X.dsym_off = st_al(D.dsym_off + 2ull * cmd_lo + 512ull * m);
  X.mb_off = D.mb_off + (uint64_t)m * st_al(ML.total);
  X.scratch_off = st_al(D.scratch_off + 8ull * start + (start >> 5) + 1280ull * m);
  X.out_off = st_al(D.out_off + 2ull * start + (start >> 10) + 4096ull * m);
  X.out_cap = 2ull * bytes + 1024u;
  md[m] = X;
  ShardState Y;
  __builtin_memset(&Y, 0, sizeof(Y));
  if (m == 0) { ShardState init; init_shard_state(J, D, &init); Y.last_bytes = init.last_bytes; Y.last_bytes_bits = init.last_bytes_bits; }
  Y.input_pos = end;
  Y.last_processed_pos = Y.last_flush_pos = start;
  Y.ncmds = ncmds;
  Y.nlits = nlits;
  Y.flint = -2;
  const uint8_t* data = input + D.in_off;
  if (start > 0) Y.prev_byte = data[start - 1];
  if (start > 1) Y.prev_byte2 = data[start - 2];
  Y.mb_valid = 1;
  Y.mb_start = start;
  Y.mb_bytes = bytes;
  Y.mb_is_last = is_last ? 1u : 0u;
  ms[m] = Y;
}

// grid = 1, block = 64.  Where the reference cuts its meta-blocks, given the tiles' counts as they stand: after the
// block of tile t when the literals / commands gathered since the last cut reach the limits or the next block would
// not fit (encode.c:1141-1166).  Writes TileRec::cut (tile t + 1 begins a meta-block) and TileRec::cmd_off; with
// `finalize` also the meta-blocks' descriptions.
#define TILE_CNT_TAILCLOSED 9 // the stream's last block closes its meta-block by the rule of encode.c:1141-1166 (see stream_tail_fix)
DEV void stream_cuts(const JobParams& J, const ShardDesc& D, TileRec* R, const uint8_t* input, ShardDesc* md, ShardState* ms,
                     uint32_t mcap, uint32_t* counters, bool finalize, uint8_t* ws) {
  const uint32_t lane = (uint32_t)wave_lane();
  const uint32_t nt = D.ntiles;
  const uint64_t block = 1ull << J.lgblock;
  uint32_t s_tile = 0, accL = 0, accC = 0, m = 0, cmd_row = 0, mb_cmd_lo = 0;
  uint32_t prev_cut = 0;                     // the row before ended with a cut behind its last tile
  bool overflow = false;
  // the gate along a stream whose first block left it open (gate_walk; one lane: only such streams pay for it)
  if ((R[0].flags & TILE_GATE_OPEN) != 0 && !finalize) {
    uint32_t again = 0;
    if (lane == 0) again = gate_walk(R, nt);
    again = wave_bcast(again, 0);
    if (again != 0 && lane == 0) glb_atomic_add(&counters[TILE_CNT_RESTART], again);
  }
  wave_sync();
  for (uint32_t r0 = 0; r0 < nt; r0 += 64u) {
    const uint32_t t = r0 + lane;
    const bool have = t < nt;
    const uint32_t nl = have ? R[t].out_nlits : 0u, nc = have ? R[t].out_ncmds : 0u, oi = have ? R[t].out_insert : 0u;
    uint32_t startlane = 0;
    uint64_t cutmask = 0;
    uint32_t carryL = accL, carryC = accC;
    for (;;) {
      const bool on = have && lane >= startlane;
      const uint32_t sl = carryL + wave_incl_scan(on ? nl : 0u), sc = carryC + wave_incl_scan(on ? nc : 0u);
      const uint64_t nb = (uint64_t)t - s_tile + 1u;
      const bool cond = on && t + 1u < nt &&
                        (sl >= J.max_literals || sc >= J.max_commands || (nb + 1u) * block > (uint64_t)J.max_metablock_size);
      const uint64_t mask = wave_ballot(cond);
      if (mask == 0) { accL = wave_bcast(sl, 63); accC = wave_bcast(sc, 63); break; }
      const int f = dev_ctz64(mask);
      cutmask |= 1ull << f;
      const uint32_t fl = wave_bcast(sl, f), fc = wave_bcast(sc, f), fi = wave_bcast(oi, f);
      const uint32_t ncm = fc + (fi != 0 ? 1u : 0u);
      if (finalize) {
        if (m < mcap) { if (lane == 0) stream_emit_mb(J, D, input, md, ms, m, s_tile, r0 + (uint32_t)f, mb_cmd_lo, ncm, fl + fi, false); }
        else overflow = true;
      }
      mb_cmd_lo += ncm;
      ++m;
      s_tile = r0 + (uint32_t)f + 1u;
      startlane = (uint32_t)f + 1u;
      carryL = carryC = 0;
    }
    // the row's cut flags and command offsets
    const uint64_t flushmask = cutmask & wave_ballot(oi != 0);
    const uint32_t ex = wave_incl_scan(nc) - nc;
    const uint32_t before = (uint32_t)dev_popc64(flushmask & ((1ull << lane) - 1ull));
    if (have) {
      R[t].cmd_off = cmd_row + ex + before;
      R[t].cut = lane == 0 ? prev_cut : (uint32_t)((cutmask >> (lane - 1u)) & 1ull);
    }
    cmd_row += wave_bcast(ex + nc, 63) + (uint32_t)dev_popc64(flushmask);
    prev_cut = (uint32_t)((cutmask >> 63) & 1ull);
  }
  if (finalize) {
    if (m < mcap) { if (lane == 0) stream_emit_mb(J, D, input, md, ms, m, s_tile, nt - 1u, mb_cmd_lo, accC, accL, true); }
    else overflow = true;
    if (lane == 0) {
      // Would the reference, given the stream's last block with MORE input expected (a PROCESS call that ends on a block
      // boundary, the FINISH coming empty), have closed the meta-block behind it?  The rule of the cuts above, with the
      // counts as they stand before the trailing insert-only command (encode.c:1160-1166 test num_literals_ /
      // num_commands_ before :1169-1173 adds it).  What follows from it: stream_tail_fix (host_plan.h).
      const TileRec& z = R[nt - 1u];
      uint32_t tl = accL, tc = accC;
      if (z.out_ncmds != 0u) {
        const Command lc = (c_tile_slot(ws, D, J, z.buf, nt - 1u) + (nt == 1u ? 0u : 1u))[z.out_ncmds - 1u];
        if ((lc.copy_len & 0x1FFFFFFu) == 0u) { tl -= lc.insert_len; tc -= 1u; }
      }
      const uint64_t nb = (uint64_t)(nt - 1u) - s_tile + 1u;
      counters[TILE_CNT_TAILCLOSED] = (tl >= J.max_literals || tc >= J.max_commands || (nb + 1u) * block > (uint64_t)J.max_metablock_size) ? 1u : 0u;
    }
    ++m;
    if (lane == 0) {
      counters[TILE_CNT_NMB] = m;
      if (overflow) { R[0].flags |= TILE_BAD | TILE_WHY_ERROR; glb_atomic_add(&counters[TILE_CNT_BAD], 1u); }
    }
  }
  wave_sync();
}

// grid = ceil(ntiles / 64), block = 64: lane = tile t; the join between t and t + 1.
DEV void stream_verify(const JobParams& J, const ShardDesc& D, TileRec* R, uint32_t t, uint32_t* counters) {
  if (t >= D.ntiles || (R[0].flags & TILE_BAD)) return;
  uint32_t why = 0;
  TileRec& c = R[t];
  if (c.flags & TILE_BAD) why |= TILE_WHY_TILE;
  if (t == 0) {
    // (a stream's other way is one wave on the whole stream: searching half of the positions twice is still
    //  far better — the bound only keeps literal-spree data, of which most positions are unstored, off the sweeps)
    if (c.nflips > D.len / 2u + 64u) why |= TILE_WHY_EVENTS;
    c.nflips = 0;
    c.pad = 0;                   // (tile_walk_over: the walks of the next pass)
  }
  // (tiles sent back by gate_walk are parsed at the top of the next pass: their joins are checked then)
  const bool pending = !(c.flags & TILE_RAN) || (t + 1u < D.ntiles && !(R[t + 1u].flags & TILE_RAN));
  if (t + 1u < D.ntiles && !pending) {
    TileRec& n = R[t + 1u];
    const bool cut = n.cut != 0;
    // (a tile without a command: a block of noise ends with literals pending — the next block's ExtendLastCommand does
    //  nothing then, encode.c:1103, and nobody asks for the last command — and a block that ExtendLastCommand consumed
    //  whole, the middle of a copy longer than a block, hands on the command it was given, longer by what it added)
    const bool hollow = c.out_ncmds == 0u && c.out_insert == 0u;
    const uint32_t eff_copy_len = hollow ? c.in_copy_len + c.in_ext : c.out_copy_len;
    const uint32_t eff_code = hollow ? c.in_code : c.out_code;
    if (!(R[0].flags & TILE_GATE_OPEN) && c.out_gate == 0u) why |= TILE_WHY_GATE;      // (cannot happen: see tile_verify)
    const uint32_t req_insert = cut ? 0u : c.out_insert;
    const bool same_cmd = cut || c.out_insert != 0u || (n.in_copy_len == eff_copy_len && n.in_code == eff_code);
    // (behind a raw meta-block the distance cache is the one that meta-block began with: k_stream_rollback)
    const int32_t* rdc = (n.rb != 0u && cut) ? n.rb_dc : c.out_dc;
    const bool same = n.in_dc[0] == rdc[0] && n.in_dc[1] == rdc[1] && n.in_dc[2] == rdc[2] &&
                      n.in_dc[3] == rdc[3] && n.in_insert == req_insert && (n.used_cut != 0) == cut && same_cmd;
    if (!same && why == 0) {
      for (int i = 0; i < 4; ++i) n.in_dc[i] = rdc[i];
      n.in_insert = req_insert;
      n.in_copy_len = cut ? 0u : eff_copy_len;
      n.in_code = eff_code;
      n.flags |= TILE_START_EVENT;
      glb_atomic_add(&counters[TILE_CNT_START], 1u);
    }
  }
  if (why != 0) {
    glb_atomic_or(&R[0].flags, TILE_BAD | why);
    glb_atomic_add(&counters[TILE_CNT_BAD], 1u);
  }
}

// grid = nchunks * ix_slices, block = 64, then one thread: the store bits that changed since the last pass, counted
// before anything is walked; more than half of the stream's positions = the rule of stream_verify, applied while it
// still saves the work (k_stream_events returns at once for a stream that is TILE_BAD).
DEV void stream_flipcount(const JobParams& J, const ShardDesc& D, uint8_t* ws, const TileRec* trecs, uint32_t cj, uint32_t w, uint32_t* counters) {
  if (trecs[0].flags & TILE_BAD) return;
  const uint32_t lane = (uint32_t)wave_lane();
  const uint32_t* skip = (const uint32_t*)(ws + J.sbm_off);
  const uint32_t* prev = (const uint32_t*)(ws + J.sbm_off + J.sbm_stride);
  const uint32_t own_lo = cj << J.chunk_log2;
  const uint32_t own_hi = umin(D.len, (cj + 1u) << J.chunk_log2);
  const uint32_t per = ix_slice_len(own_hi - own_lo, J.ix_slices);
  const uint32_t w_lo = (own_lo + w * per) / 32u, w_hi = umin((own_lo + (w + 1u) * per) / 32u, (own_hi + 31u) / 32u);
  uint32_t flips = 0;
  for (uint32_t i = w_lo + lane; i < w_hi; i += 64u) flips += (uint32_t)__builtin_popcount(skip[i] ^ prev[i]);
  if (flips != 0) glb_atomic_add(&counters[TILE_CNT_PREFLIPS], flips);
}
DEV void stream_flipcheck(const ShardDesc& D, TileRec* trecs, uint32_t* counters) {
  if (counters[TILE_CNT_PREFLIPS] > D.len / 2u + 64u && !(trecs[0].flags & TILE_BAD)) {
    trecs[0].flags |= TILE_BAD | TILE_WHY_EVENTS;
    glb_atomic_add(&counters[TILE_CNT_BAD], 1u);
  }
  counters[TILE_CNT_PREFLIPS] = 0;
}

// grid = nchunks * ix_slices, block = 64: the words of the stream's bitmap that lie in the chunk's own part.  A
// position whose bit changed is looked up in its own chunk and in the next one (where it is look-back): its
// successors in both key runs are the searches its store was, or now is, a candidate of.
DEV void stream_events(const JobParams& J, const ShardDesc& D, const ShardDesc* chunks, uint32_t cj, const uint8_t* input,
                       uint8_t* ws, TileRec* trecs, uint32_t w, uint32_t* counters) {
  if (trecs[0].flags & TILE_BAD) return;
  const uint32_t lane = (uint32_t)wave_lane();
  const uint32_t* skip = (const uint32_t*)(ws + J.sbm_off);
  const uint32_t* prev = (const uint32_t*)(ws + J.sbm_off + J.sbm_stride);
  uint32_t* ev = (uint32_t*)(ws + J.sbm_off + 2u * J.sbm_stride);
  const uint8_t* data = input + D.in_off;
  const uint32_t own_lo = cj << J.chunk_log2;
  const uint32_t own_hi = umin(D.len, (cj + 1u) << J.chunk_log2);
  const uint32_t per = ix_slice_len(own_hi - own_lo, J.ix_slices);
  const uint32_t w_lo = (own_lo + w * per) / 32u, w_hi = umin((own_lo + (w + 1u) * per) / 32u, (own_hi + 31u) / 32u);
  uint32_t flips = 0, walked = 0;
  for (uint32_t i = w_lo + lane; i < w_hi; i += 64u) {
    const uint32_t cur = skip[i];
    uint32_t diff = cur ^ prev[i];
    if (diff == 0) continue;
    // (prev[] is brought up to date by k_stream_skcount, the next kernel: the walks below ask it about other positions)
    flips += (uint32_t)__builtin_popcount(diff);
    for (; diff != 0; diff &= diff - 1u) {
      const uint32_t x = i * 32u + (uint32_t)dev_ctz32(diff);
      const bool x_unstored = ((cur >> (x & 31u)) & 1u) != 0u;
      const uint32_t key = hash_pos(ld64(data + x), J.hasher_type, J.bucket_bits).key;
      const uint32_t xt = (J.flags & JOB_FLAG_SWEEP) ? 0xFFFFFFFFu : x >> J.tile_log2;
      uint32_t stored = 0, xlast = x;
      const bool halfc = J.chunk_log2 < (uint32_t)J.lgwin;
      for (uint32_t c = cj; c <= cj + 1u && c < J.nchunks; ++c) {
        const ShardDesc& K = chunks[c];
        // (the store counter's zones of this key in this chunk are to be looked at again: stream_zones)
        ((uint32_t*)(ws + J.skt_off + (uint64_t)c * skt_chunk_bytes((uint32_t)J.bucket_bits)))[(SKT_DIRTY << J.bucket_bits) + key] = 1u;
        IxLayout L;
        ix_layout(K.len, J.ix_slices, J.ix_nb_log2, &L);
        const uint8_t* kb = ws + K.ix_off;
        const uint32_t* srt = (const uint32_t*)(kb + L.srt);
        const uint64_t* res = (const uint64_t*)(kb + L.res);
        const uint32_t total = ((const uint32_t*)(kb + L.cnt))[J.ix_slices << J.ix_nb_log2];
        const uint32_t s = (uint32_t)(res[x - K.ix_base] >> 32) & 0xFFFFFFu;
        stored = 0;
        if (x_unstored && s != 0u) {
          // the entry before x in this chunk's key run changed as well: its walk (this pass, some other lane — it is
          // looked up in this chunk too) visits x, everything x's own walk would visit, and ends where that would end,
          // x not being a store; what it leaves out by the first pass's tile rule lies in ITS tile, in front of x's
          const uint32_t qp = (srt[s - 1u] & 0xFFFFFFu) + K.ix_base;
          if (hash_pos(ld64(data + qp), J.hasher_type, J.bucket_bits).key == key &&
              (((skip[qp >> 5] ^ prev[qp >> 5]) >> (qp & 31u)) & 1u) != 0u) { stored = 16u; continue; }
        }
        xlast = x;
        bool stretch = halfc;
        for (uint32_t j = s + 1u; j < total && stored < 16u; ++j) {
          if ((++walked & 4095u) == 0u && tile_walk_over(&trecs[0], D.len, counters)) return;
          const uint32_t q = (srt[j] & 0xFFFFFFu) + K.ix_base;
          if (hash_pos(ld64(data + q), J.hasher_type, J.bucket_bits).key != key) break;
          if ((q >> J.tile_log2) != xt) glb_atomic_or(&ev[q >> 5], 1u << (q & 31u));
          const uint32_t sq = skip[q >> 5];
          if (!((sq >> (q & 31u)) & 1u)) { ++stored; stretch = false; }
          else if (stretch) {
            // (the entries right behind x that do not walk because of x: changed and unstored, one after the other —
            //  the walk into chunk cj + 2 below has to reach as far as the LAST of them sees)
            if (((sq ^ prev[q >> 5]) >> (q & 31u)) & 1u) xlast = q; else stretch = false;
          }
        }
      }
      if (J.chunk_log2 < (uint32_t)J.lgwin && cj + 2u < J.nchunks && stored < 16u) {
        // chunks of half a window: the key run of chunk cj + 1 ended before 16 stored successors were seen — the
        // searches of chunk cj + 2's own part that still have x in their ring do not find x in their chunk (they are
        // exact searches that go on in the chunk before theirs: k_index.h IxGeom::older, k_chain.h c_search_exact);
        // the walk goes on through the own part of the key's run there, as far as the window reaches
        const ShardDesc& K = chunks[cj + 2u];
        IxLayout L;
        ix_layout(K.len, J.ix_slices, J.ix_nb_log2, &L);
        const uint32_t* srt = (const uint32_t*)(ws + K.ix_off + L.srt);
        const uint32_t* kt = (const uint32_t*)(ws + J.skt_off + (uint64_t)(cj + 2u) * skt_chunk_bytes((uint32_t)J.bucket_bits));
        const uint32_t nk = 1u << J.bucket_bits;
        const uint32_t rl = kt[SKT_RL * nk + key], own = kt[SKT_OWN * nk + key];
        const uint32_t j0 = kt[SKT_RS * nk + key] + rl - own;
        for (uint32_t j = j0; j < j0 + own && stored < 16u; ++j) {
          if ((++walked & 4095u) == 0u && tile_walk_over(&trecs[0], D.len, counters)) return;
          const uint32_t q = (srt[j] & 0xFFFFFFu) + K.ix_base;
          if (q - xlast > J.max_backward_limit) break;
          if ((q >> J.tile_log2) != xt) glb_atomic_or(&ev[q >> 5], 1u << (q & 31u));
          if (!((skip[q >> 5] >> (q & 31u)) & 1u)) ++stored;
        }
      }
    }
  }
  if (flips != 0) {
    glb_atomic_add(&counters[TILE_CNT_FLIPS], flips);
    glb_atomic_add((uint32_t*)&trecs[0].nflips, flips);
  }
}

// grid = ntiles, block = 64: the tile's commands to their place in the stream's array, encoded; the insert-only
// command of a cut behind the tile.
DEV void stream_finish(const JobParams& J, const ShardDesc& D, uint8_t* ws, const TileRec* R, uint32_t tt) {
  const uint32_t lane = (uint32_t)wave_lane();
  if (R[0].flags & TILE_BAD) return;
  Command* dst = (Command*)(ws + D.cmds_off);
  const TileRec& r = R[tt];
  const Command* src = c_tile_slot(ws, D, J, r.buf, tt) + (tt == 0 ? 0u : 1u);
  const uint32_t n = r.out_ncmds;
  const bool more = tt + 1u < D.ntiles;
  // what ExtendLastCommand added to the tile's last command at the next block — and at the blocks behind that one
  // as long as it consumed them whole
  uint32_t ext = 0;
  for (uint32_t u = tt + 1u; u < D.ntiles; ++u) {
    ext += R[u].in_ext;
    if (!(R[u].out_ncmds == 0u && R[u].out_insert == 0u)) break;
  }
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
  if (lane == 0 && more && R[tt + 1u].cut != 0 && r.out_insert != 0) dst[r.cmd_off + n] = make_insert_command(r.out_insert);
}

// ---- the 16-bit store counter of a tiled stream ---------------------------------------------------------------------
// FindLongestMatch sees min(16, stores of the key so far mod 65536) ring slots (..64_simd_inc.h:250-257): the first
// 16 searches of a key after every 65536th store of it see fewer.  Which positions those are depends on every store
// of the key since the stream's start: per chunk and key, the stores before the chunk's look-back (SKT_B) come from
// the run lengths (k_ix_bucket) less the unstored positions (k_stream_skcount counts the bitmap's bits per chunk and
// key), summed over the chunks before (k_stream_kprefix); k_stream_zones then walks the runs that contain a wrap,
// marks the searches behind it in res[] (IX_DANGER, the visible slots in the low bits, IX_KIND_SLOW: the chain
// searches them itself) and raises an event where a mark changed.  Part of every pass of the sweep loop: the marks
// follow the bitmap as it settles.
DEV uint32_t* stream_kt(const JobParams& J, uint8_t* ws, uint32_t c) {
  return (uint32_t*)(ws + J.skt_off + (uint64_t)c * skt_chunk_bytes((uint32_t)J.bucket_bits));
}
// grid = nchunks * ix_slices, block = 64: (first clears SKT_SK of the chunk — a separate launch, k_stream_skclear)
DEV void stream_skclear(const JobParams& J, uint8_t* ws, uint32_t c, uint32_t w) {
  uint32_t* kt = stream_kt(J, ws, c) + ((uint64_t)SKT_SK << J.bucket_bits);
  const uint32_t nk = 1u << J.bucket_bits, share = (nk + J.ix_slices - 1u) / J.ix_slices;
  for (uint32_t i = w * share + (uint32_t)wave_lane(); i < umin((w + 1u) * share, nk); i += 64u) kt[i] = 0;
}
DEV void stream_skcount(const JobParams& J, const ShardDesc& D, const uint8_t* input, uint8_t* ws, uint32_t c, uint32_t w) {
  const uint32_t lane = (uint32_t)wave_lane();
  const uint32_t* skip = (const uint32_t*)(ws + J.sbm_off);
  const uint8_t* data = input + D.in_off;
  uint32_t* sk = stream_kt(J, ws, c) + ((uint64_t)SKT_SK << J.bucket_bits);
  const uint32_t own_lo = c << J.chunk_log2, own_hi = umin(D.len, (c + 1u) << J.chunk_log2);
  const uint32_t per = ix_slice_len(own_hi - own_lo, J.ix_slices);
  const uint32_t w_lo = (own_lo + w * per) / 32u, w_hi = umin((own_lo + (w + 1u) * per) / 32u, (own_hi + 31u) / 32u);
  uint32_t* prev = (uint32_t*)(ws + J.sbm_off + J.sbm_stride);
  for (uint32_t i = w_lo + lane; i < w_hi; i += 64u) {
    const uint32_t cur = skip[i];
    if (prev[i] != cur) prev[i] = cur;           // (what k_stream_events of the next pass compares with)
    for (uint32_t v = cur; v != 0; v &= v - 1u) {
      const uint32_t x = i * 32u + (uint32_t)dev_ctz32(v);
      glb_atomic_add(&sk[hash_pos(ld64(data + x), J.hasher_type, J.bucket_bits).key], 1u);
    }
  }
}
// grid = keys / 64, block = 64: lane = key; SKT_B of chunk c = stores in the own parts of the chunks 0 .. c - 2.
DEV void stream_kprefix(const JobParams& J, uint8_t* ws, uint32_t key) {
  const uint32_t nk = 1u << J.bucket_bits;
  if (key >= nk) return;
  uint32_t run = 0;
  for (uint32_t c = 0; c < J.nchunks; ++c) {
    uint32_t* kt = stream_kt(J, ws, c);
    if (kt[SKT_B * nk + key] != run) { kt[SKT_B * nk + key] = run; kt[SKT_DIRTY * nk + key] = 1u; }
    if (c >= 1u) { const uint32_t* kp = stream_kt(J, ws, c - 1u); run += kp[SKT_OWN * nk + key] - kp[SKT_SK * nk + key]; }
  }
}
// grid = nchunks * keys / 64, block = 64: the wave's 64 keys of chunk c, one after the other.
// Only the runs that changed since they were walked last are walked again (SKT_DIRTY; `all`: the first launch): a
// store bit of the run flipped, or the number of stores in front of it did.  The walk is bound by its loads — an
// entry and its word of the bitmap per position of the chunk and its look-back: 6 ms a launch, four launches a stream,
// a quarter of the time of 256 MiB at lgwin 24 when every launch walked everything (profiles/r06_x).
DEV void stream_zones(const JobParams& J, const ShardDesc& D, const ShardDesc* chunks, uint8_t* ws, uint32_t c, uint32_t kg, uint32_t* counters, bool all) {
  const uint32_t lane = (uint32_t)wave_lane();
  const uint32_t nk = 1u << J.bucket_bits;
  uint32_t* kt = stream_kt(J, ws, c);
  const ShardDesc& K = chunks[c];
  IxLayout L;
  ix_layout(K.len, J.ix_slices, J.ix_nb_log2, &L);
  const uint32_t* srt = (const uint32_t*)(ws + K.ix_off + L.srt);
  uint64_t* res = (uint64_t*)(ws + K.ix_off + L.res);
  const uint32_t* skip = (const uint32_t*)(ws + J.sbm_off);
  uint32_t* ev = (uint32_t*)(ws + J.sbm_off + 2u * J.sbm_stride);
  const uint32_t key = kg * 64u + lane;
  const uint32_t rl_l = kt[SKT_RL * nk + key], b_l = kt[SKT_B * nk + key], zlo_l = kt[SKT_ZLO * nk + key], zhi_l = kt[SKT_ZHI * nk + key];
  // a wrap inside the run (however few of its entries are stored, the run cannot wrap if this says no), or marks
  // of an earlier pass to look after
  const bool dirty_l = all || kt[SKT_DIRTY * nk + key] != 0u;
  if (dirty_l && !all) kt[SKT_DIRTY * nk + key] = 0u;
  uint64_t todo = wave_ballot(dirty_l && ((rl_l != 0u && (b_l & 0xFFFFu) + rl_l >= 65536u) || zhi_l > zlo_l));
  uint32_t changes = 0;
  for (; todo != 0; todo &= todo - 1ull) {
    const int src = dev_ctz64(todo);
    const uint32_t k = kg * 64u + (uint32_t)src;
    const uint32_t rs = kt[SKT_RS * nk + k], rl = wave_bcast(rl_l, src), B = wave_bcast(b_l, src);
    const uint32_t zlo = wave_bcast(zlo_l, src), zhi = wave_bcast(zhi_l, src);
    // the run, 64 entries per step: `stored` = stores of the key before the entry.  Four steps' loads go out together —
    // sorted entry, then its word of the bitmap: two dependent round trips per step were 6.5 ms a launch for the most
    // common key of a 16 MiB chunk (a run of several hundred thousand entries on ONE wave), four launches a 256 MiB
    // stream: a quarter of its time at lgwin 24 (profiles/r06_x) — the steps themselves only pass `stored` on.
    uint32_t stored = B, nlo = 0xFFFFFFFFu, nhi = 0;
    for (uint32_t i00 = 0; i00 < rl; i00 += 256u) {
      if (all) {
        // the first launch: nothing is unstored yet (k_ix_count cleared the bitmap), the counter at entry i is B + i —
        // straight to the next 256 entries that hold a zone, without loading a bitmap word
        const uint32_t v = B + i00, r = v & 0xFFFFu;
        uint32_t i = r < 16u ? i00 : i00 + (65536u - r);
        if (B + i < 65536u) i = 65536u - B;
        if (i >= rl) break;
        i00 = i & ~63u;
        stored = B + i00;
      }
      uint32_t w4[4], s4[4];
#pragma unroll
      for (uint32_t u = 0; u < 4u; ++u) { const uint32_t i = i00 + 64u * u + lane; w4[u] = i < rl ? srt[rs + i] : 0u; }
#pragma unroll
      for (uint32_t u = 0; u < 4u; ++u) { const uint32_t P = (w4[u] & 0xFFFFFFu) + K.ix_base; s4[u] = (!all && i00 + 64u * u + lane < rl) ? skip[P >> 5] : 0u; }
#pragma unroll
      for (uint32_t u = 0; u < 4u; ++u) {
        const uint32_t i0 = i00 + 64u * u;
        if (i0 >= rl) break;
        const uint32_t i = i0 + lane;
        const bool in = i < rl;
        const uint32_t w0 = w4[u];
        const uint32_t p = w0 & 0xFFFFFFu, P = p + K.ix_base;
        const bool st = in && !((s4[u] >> (P & 31u)) & 1u);
        const uint64_t sm = wave_ballot(st);
        const uint32_t before = stored + (uint32_t)dev_popc64(sm & ((1ull << lane) - 1ull));     // the counter when this entry is searched
        const uint32_t vis = before & 0xFFFFu;
        const bool zone = in && before >= 65536u && vis < 16u && p >= K.ix_ownc;
        const bool old = in && rs + i >= zlo && rs + i < zhi;
        if (zone || old) {
          const uint64_t r = res[p];
          const uint32_t lo = (uint32_t)r, hi = (uint32_t)(r >> 32);
          uint32_t nlo2 = lo, nhi2 = hi;
          if (zone) { nlo2 = (IX_KIND_SLOW << 30) | vis; nhi2 = hi | IX_DANGER; }
          else if (hi & IX_DANGER) { nlo2 = IX_KIND_SLOW << 30; nhi2 = hi & ~IX_DANGER; }
          if (nlo2 != lo || nhi2 != hi) {
            res[p] = (uint64_t)nlo2 | ((uint64_t)nhi2 << 32);
            glb_atomic_or(&ev[P >> 5], 1u << (P & 31u));
            ++changes;
          }
        }
        const uint64_t zm = wave_ballot(zone);
        if (zm != 0) { nlo = umin(nlo, rs + i0 + (uint32_t)dev_ctz64(zm)); nhi = umax(nhi, rs + i0 + 64u - (uint32_t)__builtin_clzll(zm)); }
        stored += (uint32_t)dev_popc64(sm);
      }
    }
    if (lane == 0) { kt[SKT_ZLO * nk + k] = nhi > nlo ? nlo : 0u; kt[SKT_ZHI * nk + k] = nhi > nlo ? nhi : 0u; }
  }
  changes = wave_incl_scan(changes);
  if (lane == 63 && changes != 0) glb_atomic_add(&counters[TILE_CNT_FLIPS], changes);
}

// Bits of BrotliStoreUncompressedMetaBlock's header for `len` bytes (brotli_bit_stream.c:1321-1352; k_round.h:
// emit_raw_metablock): ISLAST 0, MNIBBLES, MLEN - 1, ISUNCOMPRESSED.
DEV uint32_t stream_raw_header(uint32_t len, uint64_t* value) {
  const uint32_t lg = (len == 1u) ? 1u : log2floor(len - 1u) + 1u;
  const uint32_t mnibbles = (lg < 16u ? 16u : (lg + 3u)) / 4u;
  *value = ((uint64_t)(mnibbles - 4u) << 1) | ((uint64_t)(len - 1u) << 3) | (1ull << (3u + 4u * mnibbles));
  return 4u + 4u * mnibbles;
}
// grid = 1, block = 64: the meta-blocks' bit offsets in the stream, one after the other (lane 0: where a raw
// meta-block's payload starts depends on everything in front of it) — and which meta-blocks are raw: the ones
// k_build / k_store said (mb_was_raw 1), and of the ones that depend on the starting bit (2) those for which the
// reference's comparison (encode.c:604: the bytes counted from the byte the meta-block's first bit is in, the last
// one's padding included) says so.  moff[nmb] = bits of the stream.
DEV void stream_scan(const JobParams& J, const ShardDesc& D, ShardState* ms, uint32_t nmb, uint64_t* moff, uint32_t* counters) {
  if (wave_lane() != 0) return;
  ShardState init;
  init_shard_state(J, D, &init);
  const uint32_t hb = init.last_bytes_bits;             // the stream header in front of meta-block 0
  uint64_t o = 0;
  uint32_t fault = 0;
  for (uint32_t m = 0; m < nmb; ++m) {
    ShardState& S = ms[m];
    if (S.error != 0 || S.mb_valid != 0) fault = 1;
    const uint32_t bytes = S.mb_bytes;
    const bool last = S.mb_is_last != 0;
    const uint64_t T = S.out_bytes * 8u + S.last_bytes_bits;       // compressed form (meta-block 0: the header bits in front)
    bool raw = S.mb_was_raw == 1u;
    if (S.mb_was_raw == 2u) {
      uint64_t sx = (m == 0 ? T : (o & 7u) + T);
      // (JOB_FLAG_TAILFIN and the rule closed the last meta-block: the reference wrote it with is_last = 0 — no padding
      //  in its comparison — and the empty last one behind it; stream_tail_fix, host_plan.h)
      if (last && !((J.flags & JOB_FLAG_TAILFIN) != 0u && counters[TILE_CNT_TAILCLOSED] != 0u)) sx = (sx + 7u) & ~(uint64_t)7u;
      raw = (uint64_t)bytes + 4u < (sx >> 3);
    }
    moff[m] = o;
    if (raw) {
      uint64_t hv;
      uint64_t p = o + (m == 0 ? hb : 0u) + stream_raw_header(bytes, &hv);
      p = ((p + 7u) & ~(uint64_t)7u) + 8ull * bytes;
      if (last) p = (p + 2u + 7u) & ~(uint64_t)7u;
      o = p;
    } else {
      o += T;
    }
    S.mb_was_raw = raw ? 1u : 0u;
  }
  moff[nmb] = o;
  // for stream_tail_fix: does the last meta-block carry ISLAST itself (a raw one never does) and has the rule of the
  // cuts closed it; the bit its ISLAST sits at
  moff[nmb + 1u] = (nmb != 0u && ms[nmb - 1u].mb_was_raw == 0u && counters[TILE_CNT_TAILCLOSED] != 0u) ? 1u : 0u;
  moff[nmb + 2u] = nmb != 0u ? moff[nmb - 1u] + (nmb == 1u ? hb : 0u) : 0u;
  if (fault) glb_atomic_add(&counters[TILE_CNT_RAW], 1u);
}

// grid = 1, block = 64: what the raw meta-blocks mean for the parse — the tile behind a raw meta-block starts from
// the distance cache that meta-block started from.  Sets TileRec::rb / rb_dc of the tiles that begin a meta-block
// and counts the tiles for which that is news (counters[TILE_CNT_RBCHG]): the sweep loop runs again then.
#define TILE_CNT_RBCHG 8
DEV void stream_rollback(const JobParams& J, const ShardDesc& D, const ShardState* ms, uint32_t nmb, TileRec* R, uint32_t* counters) {
  const uint32_t lane = (uint32_t)wave_lane();
  for (uint32_t t = lane; t < D.ntiles; t += 64u) R[t].rb_new = 0;
  wave_sync();
  if (lane == 0) {
    ShardState init;
    init_shard_state(J, D, &init);
    int32_t dc[4];                                       // the cache meta-block m starts from
    for (int i = 0; i < 4; ++i) dc[i] = init.dist_cache[i];
    for (uint32_t m = 0; m + 1u < nmb; ++m) {
      const uint32_t tn = ms[m + 1u].mb_start >> J.tile_log2;   // the tile the next meta-block begins with
      TileRec& n = R[tn];
      if (ms[m].mb_was_raw != 0u) {
        n.rb_new = 1;
        for (int i = 0; i < 4; ++i) n.rb_dc_new[i] = dc[i];    // (and the next one starts from the same cache)
      } else {
        for (int i = 0; i < 4; ++i) dc[i] = n.used_dc[i];       // what the tile was parsed from = the cache behind m
      }
    }
  }
  wave_sync();
  uint32_t changed = 0;
  for (uint32_t t = lane; t < D.ntiles; t += 64u) {
    TileRec& r = R[t];
    bool diff = r.rb != r.rb_new;
    if (r.rb_new != 0u) for (int i = 0; i < 4; ++i) diff = diff || r.rb_dc[i] != r.rb_dc_new[i];
    if (diff) {
      r.rb = r.rb_new;
      for (int i = 0; i < 4; ++i) r.rb_dc[i] = r.rb_dc_new[i];
      ++changed;
    }
  }
  changed = wave_incl_scan(changed);
  if (lane == 63 && changed != 0) glb_atomic_add(&counters[TILE_CNT_RBCHG], changed);
  wave_sync();
}

// grid = nmb * parts, block = 256: the bits of meta-block m (written from bit 0 of its own buffer) to bit moff[m] of
// the stream, 32 bits per lane and step; the first and last word of a meta-block are shared with its neighbours.
DEV void stream_place_bits(uint32_t* dst, uint64_t at, uint64_t value, uint32_t nbits) {     // nbits <= 40, one thread
  const uint64_t w = at >> 5;
  const uint32_t sh = (uint32_t)(at & 31u);
  glb_atomic_or(&dst[w], (uint32_t)(value << sh));
  if (sh + nbits > 32u) glb_atomic_or(&dst[w + 1u], (uint32_t)(value >> (32u - sh)));
  if (sh + nbits > 64u) glb_atomic_or(&dst[w + 2u], (uint32_t)(value >> (64u - sh)));
}
DEV void stream_place(const JobParams& J, const ShardDesc& D, const ShardDesc* md, const ShardState* ms, const uint64_t* moff,
                      const uint8_t* input, const uint8_t* ws, uint8_t* out,
                      uint32_t m, uint32_t part, uint32_t parts, uint32_t tid, uint32_t nthreads) {
  if (ms[m].mb_was_raw != 0u) {
    // BrotliStoreUncompressedMetaBlock at its place in the stream: header bits, the payload from the next byte on
    ShardState init;
    init_shard_state(J, D, &init);
    const uint32_t bytes = ms[m].mb_bytes;
    uint64_t hv;
    const uint32_t hbits = stream_raw_header(bytes, &hv);
    uint64_t p = moff[m];
    if (part == 0 && tid == 0) {
      if (m == 0 && init.last_bytes_bits != 0u) stream_place_bits((uint32_t*)out, p, init.last_bytes, init.last_bytes_bits);
      stream_place_bits((uint32_t*)out, p + (m == 0 ? init.last_bytes_bits : 0u), hv, hbits);
    }
    p += (m == 0 ? init.last_bytes_bits : 0u) + hbits;
    const uint64_t b0 = (p + 7u) >> 3;
    const uint8_t* src = input + md[m].in_off + ms[m].mb_start;
    const uint64_t per = ((uint64_t)bytes + parts - 1u) / parts;
    const uint64_t lo = (uint64_t)part * per, hi = lo + per < bytes ? lo + per : bytes;
    for (uint64_t i = lo + tid; i < hi; i += nthreads) out[b0 + i] = src[i];
    if (ms[m].mb_is_last != 0u && part == 0 && tid == 0) stream_place_bits((uint32_t*)out, (b0 + bytes) * 8u, 3u, 2u);   // ISLAST, ISEMPTY
    return;
  }
  const uint64_t nbits = ms[m].out_bytes * 8u + ms[m].last_bytes_bits;
  if (nbits == 0) return;
  const uint64_t o = moff[m];
  const uint32_t* src = (const uint32_t*)(ws + md[m].out_off);
  uint32_t* dst = (uint32_t*)out;
  const uint64_t k0 = o >> 5, k1 = (o + nbits + 31u) >> 5;      // destination words [k0, k1)
  const uint64_t per = (k1 - k0 + parts - 1u) / parts;
  const uint64_t lo = k0 + (uint64_t)part * per, hi = lo + per < k1 ? lo + per : k1;
  for (uint64_t k = lo + tid; k < hi; k += nthreads) {
    const int64_t rel = (int64_t)(k * 32u) - (int64_t)o;         // first source bit of this word
    const uint64_t a = rel < 0 ? 0u : (uint64_t)rel;
    const uint64_t b = (uint64_t)(rel + 32) < nbits ? (uint64_t)(rel + 32) : nbits;
    const uint32_t cnt = (uint32_t)(b - a);
    const uint64_t wd = (uint64_t)src[a >> 5] | ((uint64_t)src[(a >> 5) + 1u] << 32);
    uint32_t bits = (uint32_t)(wd >> (a & 31u));
    if (cnt < 32u) bits &= (1u << cnt) - 1u;
    const uint32_t val = bits << (uint32_t)((int64_t)a - rel);
    if (k == k0 || k + 1u == k1) glb_atomic_or(&dst[k], val);
    else dst[k] = val;
  }
}

#endif  // BROTLI_AMD_CSRC_K_TILE_H_
