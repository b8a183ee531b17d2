// brotli_amd/csrc/k_parse_quick.h — K1 for qualities 2 - 4: one encoder shard per wavefront
// over the HashLongestMatchQuickly family.
//
// Semantics: CreateBackwardReferences (c/enc/backward_references_inc.h:10-242) below
// MIN_QUALITY_FOR_EXTENSIVE_REFERENCE_SEARCH — the lazy probe starts from the length it has to
// beat (:127-128) — over H2 / H3 / H4 / H54 (c/enc/hash_longest_match_quickly_inc.h, template
// parameters c/enc/hash.h:251-279, 329-338, choice c/enc/quality.h:176-179): a table of
// 1 << bucket_bits positions, no tags, no counters; a position is stored in one of
// 1 << sweep_bits slots (key + 8 i, i picked by bits 3.. of the position), a search reads the
// last distance and then all of them in slot order.
//
// Design: the driver, the static dictionary, command construction and the EncodeData glue are
// those of k_parse4.h / k_parse_deep.h (wave-uniform state, one shard per wave).  A search puts
// its candidates on lanes 0 .. sweep (lane 0: the last distance): every lane loads 32 bytes of
// its candidate, keeps which of them equal the input's (a 32-bit mask) and extends past 32 on
// its own.  The reference walks the candidates in order and looks at one only if it agrees with
// the input at offset best_len — a byte *behind* a shorter candidate's first mismatch can
// decide, so the outcome is not an arg-max: the walk is replayed here step by step, uniformly,
// reading the deciding byte from the candidate's mask (or, beyond 32 bytes, from memory).
// Insertions are positions in increasing order over the whole shard, so a StoreRange is one
// atomic max per position, 64 positions at a time; the searches read the table through the L2
// (where the atomics land), which keeps a wave's accesses to a slot in order without fences.
//
// The same kernel serves qualities 5 - 9 at windows of 10 - 16 bits: the forgetful-chain hashers
// H40 / H41 / H42 (c/enc/hash_forgetful_chain_inc.h, parameters c/enc/hash.h:296-326,
// c/enc/quality.h:180-181) — see the fc_* functions below.
#ifndef BROTLI_AMD_CSRC_K_PARSE_QUICK_H_
#define BROTLI_AMD_CSRC_K_PARSE_QUICK_H_

#include "k_parse_deep.h"

struct QuickGeom {
  uint32_t mask;        // (1 << bucket_bits) - 1
  uint32_t sweep;       // slots a key can land in
  uint32_t hash_shift;  // 64 - 8 * HASH_LEN
  uint32_t bucket_bits;
  bool use_dictionary;
};
DEV QuickGeom quick_geom(const JobParams& J) {
  QuickGeom G;
  G.bucket_bits = (uint32_t)J.bucket_bits;
  G.mask = (1u << J.bucket_bits) - 1u;
  G.sweep = 1u << J.block_bits;                       // (block_bits carries BUCKET_SWEEP_BITS here)
  G.hash_shift = J.hasher_type == 54 ? 8u : 24u;      // HASH_LEN 7 / 5
  G.use_dictionary = J.hasher_type == 2 || J.hasher_type == 4;
  return G;
}
// HashBytes, hash_longest_match_quickly_inc.h:23-29
DEV uint32_t quick_key(const QuickGeom& G, uint64_t x) {
  return (uint32_t)(((x << G.hash_shift) * 0x1FE35A7BD3579BD3ull) >> (64u - G.bucket_bits));
}
DEV uint32_t quick_slot(const QuickGeom& G, uint32_t key, uint32_t pos) {
  return (key + (pos & ((G.sweep - 1u) << 3))) & G.mask;
}

// ---- ordered insertion (Store / StoreRange, :93-118) --------------------------------------
DEV void k_drain_stores(const JobParams& J, const QuickGeom& G, QShard& g) {
  (void)J;
  if (g.st_count == 0) return;
  uint32_t* table = (uint32_t*)g.table;
  const int lane = wave_lane();
  for (uint32_t i0 = 0; i0 < g.st_count; i0 += 64) {
    const uint32_t i = i0 + (uint32_t)lane;
    if (i < g.st_count) {
      const uint32_t pos = g.st_first + i * g.st_stride;
      glb_atomic_max(&table[quick_slot(G, quick_key(G, ld64(g.data + pos)), pos)], pos);
    }
  }
  wave_sync();     // (the table is read through the L2 as well: no fence, the wave's accesses to a word stay in order)
  g.st_count = 0;
}

// bit k set = byte k of a and b differ (k < 8)
DEV uint32_t quick_neq8(uint64_t a, uint64_t b) {
  uint64_t y = a ^ b;
  y |= y >> 4; y |= y >> 2; y |= y >> 1;
  y &= 0x0101010101010101ull;
  return (uint32_t)((y * 0x0102040810204080ull) >> 56);
}

// ---- FindLongestMatch (:141-262) ------------------------------------------------------------
// `len_in`: out->len as the caller hands it in.
DEV QResult k_search(const JobParams& J, const QuickGeom& G, const DeviceTables* T, QShard& g, uint32_t P,
                     uint32_t len_in) {
  const int lane = wave_lane();
  uint32_t* table = (uint32_t*)g.table;
  const uint32_t max_length = g.pos_end - P;
  const uint32_t max_backward = umin(P, J.max_backward_limit);
  const uint32_t key = quick_key(G, ld64(g.data + P));
  // candidates: lane 0 the last distance, lanes 1 .. sweep the slots in order
  uint32_t prev = 0, backward = 0;
  bool valid = false;
  if (lane == 0) {
    backward = (uint32_t)g.dc[0];
    valid = g.dc[0] > 0 && backward <= P && backward <= max_backward;     // prev_ix < cur_ix (:159)
    prev = P - backward;
  } else if ((uint32_t)lane <= G.sweep) {
    prev = glb_load_l2(&table[(key + (((uint32_t)lane - 1u) << 3)) & G.mask]);
    backward = P - prev;
    valid = backward != 0u && backward <= max_backward;
  }
  if (G.sweep == 1u) {                                 // :187-188: stored before it is looked at
    wave_sync();
    if (lane == 0) table[key] = P;
  }
  uint32_t neq = 0xFFFFFFFFu, len = 0;
  if (valid) {
    const B32 a = load_b32(g.data + P), b = load_b32(g.data + prev);
    neq = quick_neq8(a.q[0], b.q[0]) | (quick_neq8(a.q[1], b.q[1]) << 8) |
          (quick_neq8(a.q[2], b.q[2]) << 16) | (quick_neq8(a.q[3], b.q[3]) << 24);
    len = neq ? (uint32_t)dev_ctz32(neq) : 32u;
    if (len == 32u && max_length > 32u) len = q_extend(g.data, P, prev, max_length);
    len = umin(len, max_length);
  }
  // the walk, uniform over the wave
  QResult out;
  out.len = len_in; out.distance = 0; out.score = K_MIN_SCORE; out.delta = 0;
  uint32_t best_len = len_in;
  bool ask_dictionary = true;
  const uint32_t ncand = G.sweep + 1u;
  for (uint32_t c = 0; c < ncand && ask_dictionary; ++c) {
    const uint32_t c_valid = wave_shfl((uint32_t)valid, (int)c);
    const uint32_t c_len = wave_shfl(len, (int)c), c_back = wave_shfl(backward, (int)c);
    const uint32_t c_prev = wave_shfl(prev, (int)c), c_neq = wave_shfl(neq, (int)c);
    // compare_char == data[prev_ix + best_len] (:161, :190, :221): read before the validity test
    // in the bucket phase; an invalid candidate goes away either way
    // (at best_len == max_length the input's byte is the one behind the block: d_ring_byte)
    bool agrees;
    if (c_len > best_len) agrees = true;
    else if (best_len < 32u && best_len < max_length) agrees = c_valid != 0u && ((c_neq >> best_len) & 1u) == 0u;
    else agrees = c_valid != 0u && g.data[c_prev + best_len] == d_ring_byte(J, g, P + best_len);
    if (c == 0) {
      if (!c_valid || !agrees || c_len < 4u) continue;
      const uint32_t score = 135u * c_len + 1935u;     // BackwardReferenceScoreUsingLastDistance
      if (out.score < score) {
        out.len = c_len; out.distance = c_back; out.score = score;
        if (G.sweep == 1u) ask_dictionary = false;     // :170-172: done (the position is stored already)
        best_len = c_len;
      }
      continue;
    }
    if (G.sweep == 1u) {
      // :189-207: a candidate that disagrees or is out of reach ends the search without the dictionary
      if (!c_valid || !agrees) { ask_dictionary = false; break; }
      if (c_len >= 4u) {
        const uint32_t score = 1920u + 135u * c_len - 30u * log2floor(c_back);
        if (out.score < score) { out.len = c_len; out.distance = c_back; out.score = score; ask_dictionary = false; }
      }
      break;
    }
    if (!c_valid || !agrees || c_len < 4u) continue;
    const uint32_t score = 1920u + 135u * c_len - 30u * log2floor(c_back);
    if (out.score < score) {
      best_len = c_len;
      out.len = c_len; out.distance = c_back; out.score = score;
    }
  }
  if (G.use_dictionary && ask_dictionary && out.score == K_MIN_SCORE)
    q_dict_search(J, T, g, true, P, max_length, out);
  if (G.sweep != 1u) {
    wave_sync();
    if (lane == 0) table[quick_slot(G, key, P)] = P;
  }
  wave_sync();
  return out;
}

// ---- the forgetful-chain family (H40 / H41 / H42) ------------------------------------------
// Per shard, in the table region: addr u32[32768] (position of the newest node of a bucket,
// 0xCCCCCCCC = none), head u16[32768] (its slot), tiny u8[65536] (low byte of the key of the
// position with these low 16 bits), free u16[banks], then the banks: {delta u16, next u16} per
// slot.  A node is stored into the next slot of its key's bank, whatever lived there is forgotten
// — chains may run into each other's tails, and the walk follows whatever it finds, exactly as the
// reference's does (:133-149, 249-287).  Everything here is a chain of dependent reads and
// writes: one lane's worth of work done uniformly by the wave.
struct FcGeom {
  uint32_t* addr;
  uint16_t* head;
  uint8_t* tiny;
  uint16_t* free_idx;
  uint32_t* slots;       // delta | next << 16
  uint32_t bank_mask, bank_bits, max_hops;
};
DEV uint32_t fc_table_bytes_before_slots() { return 32768u * 4u + 32768u * 2u + 65536u + 1024u; }
DEV FcGeom fc_geom(const JobParams& J, const QShard& g) {
  FcGeom F;
  F.addr = (uint32_t*)g.table;
  F.head = (uint16_t*)(g.table + 32768u * 4u);
  F.tiny = g.table + 32768u * 6u;
  F.free_idx = (uint16_t*)(g.table + 32768u * 6u + 65536u);
  F.slots = (uint32_t*)(g.table + fc_table_bytes_before_slots());
  F.bank_bits = J.hasher_type == 42 ? 9u : 16u;
  F.bank_mask = J.hasher_type == 42 ? 511u : 0u;
  F.max_hops = (J.quality > 6 ? 7u : 8u) << (J.quality - 4);      // :87
  return F;
}
DEV uint32_t fc_key(const uint8_t* p) { return (ld32(p) * 0x1E35A7BDu) >> (32 - 15); }

// Store (:133-149) of position ix, every lane the same (plain loads and stores of one wave to
// one address stay in order).
DEV void fc_store(const FcGeom& F, const QShard& g, uint32_t ix) {
  const int lane = wave_lane();
  const uint32_t key = fc_key(g.data + ix);
  const uint32_t bank = key & F.bank_mask;
  const uint32_t idx = (uint32_t)F.free_idx[bank] & ((1u << F.bank_bits) - 1u);
  uint32_t delta = ix - F.addr[key];
  if (delta > 0xFFFFu) delta = 0xFFFFu;
  const uint32_t node = delta | ((uint32_t)F.head[key] << 16);
  wave_sync();
  if (lane == 0) {
    F.free_idx[bank] = (uint16_t)(F.free_idx[bank] + 1u);
    F.tiny[ix & 0xFFFFu] = (uint8_t)key;
    F.slots[(bank << F.bank_bits) + idx] = node;
    F.addr[key] = ix;
    F.head[key] = (uint16_t)idx;
  }
  wave_sync();
}

DEV void fc_drain_stores(const FcGeom& F, QShard& g) {
  for (uint32_t i = 0; i < g.st_count; ++i) fc_store(F, g, g.st_first + i * g.st_stride);
  g.st_count = 0;
}

// FindMatchLengthWithLimit, uniform
DEV uint32_t fc_match_len(const uint8_t* data, uint32_t a, uint32_t b, uint32_t limit) {
  uint32_t off = 0;
  while (off + 8 <= limit) {
    const uint64_t x = ld64(data + a + off) ^ ld64(data + b + off);
    if (x) return off + ((uint32_t)dev_ctz64(x) >> 3);
    off += 8;
  }
  while (off < limit && data[a + off] == data[b + off]) ++off;
  return off;
}

// FindLongestMatch (:190-298)
DEV QResult fc_search(const JobParams& J, const FcGeom& F, const DeviceTables* T, QShard& g, uint32_t P) {
  const int lane = wave_lane();
  const uint32_t max_length = g.pos_end - P;
  const uint32_t max_backward = umin(P, J.max_backward_limit);
  const uint32_t key = fc_key(g.data + P);
  const uint32_t tiny_hash = key & 0xFFu;
  QResult out;
  out.len = 0; out.distance = 0; out.score = K_MIN_SCORE; out.delta = 0;
  uint32_t best_len = 0;
  {
    // the distance cache: one candidate per lane, the best adjusted score wins, the earlier entry
    // on a tie (each is accepted only if it beats what came before, :232-243)
    uint32_t k = 0, len = 0, backward = 0;
    if (lane < J.ndist) {
      backward = d_dc_entry(g, lane);
      const uint32_t prev = P - backward;
      bool ok = (int32_t)backward > 0 && backward <= P && backward <= max_backward;
      if (lane > 0 && F.tiny[prev & 0xFFFFu] != tiny_hash) ok = false;      // (:219, before the range test: same outcome)
      if (ok) {
        len = fc_match_len(g.data, prev, P, max_length);
        if (len >= 2u) {
          uint32_t score = 135u * len + 1935u;
          if (lane != 0) score -= 39u + ((0x1CA10u >> ((uint32_t)lane & 0xEu)) & 0xEu);
          if (score > K_MIN_SCORE) k = (score << 5) | (31u - (uint32_t)lane);
        }
      }
    }
    const uint32_t best = d_max(k);
    if (best != 0) {
      const int src = 31 - (int)(best & 31u);
      out.score = best >> 5;
      out.len = wave_shfl(len, src);
      out.distance = wave_shfl(backward, src);
      best_len = out.len;
    }
  }
  if (best_len < 3u) best_len = 3u;
  {
    const uint32_t ring_mask = J.ring_mask;
    const uint32_t cur_masked = P & ring_mask;
    const uint32_t bank = key & F.bank_mask;
    uint32_t backward = 0, hops = F.max_hops;
    uint32_t delta = P - F.addr[key];
    uint32_t slot = F.head[key];
    while (hops--) {
      backward += delta;
      if (backward > max_backward || backward < delta) break;       // (second test: the 64-bit sum of the reference cannot wrap)
      const uint32_t prev = P - backward;
      const uint32_t node = F.slots[(bank << F.bank_bits) + slot];
      slot = node >> 16;
      delta = node & 0xFFFFu;
      if (cur_masked + best_len > ring_mask || (prev & ring_mask) + best_len > ring_mask) continue;
      // the four bytes ending at offset best_len (:262-266); the input's last one may be the byte
      // behind the block
      uint32_t a = ld32(g.data + P + best_len - 3u), b = ld32(g.data + prev + best_len - 3u);
      if (best_len == max_length) a = (a & 0x00FFFFFFu) | (d_ring_byte(J, g, P + best_len) << 24);
      if (a != b) continue;
      const uint32_t len = fc_match_len(g.data, prev, P, max_length);
      if (len >= 4u) {
        const uint32_t score = 1920u + 135u * len - 30u * log2floor(backward);
        if (out.score < score) { out.score = score; out.len = len; out.distance = backward; best_len = len; }
      }
    }
    fc_store(F, g, P);
  }
  if (out.score == K_MIN_SCORE) q_dict_search(J, T, g, true, P, max_length, out);
  return out;
}

DEV void k_setup_block(const JobParams& J, const QuickGeom& G, QShard& g) {
  const int lane = wave_lane();
  const uint32_t htl = hasher_htl(J.hasher_type);
  if (g.blk_flags & QBLK_STITCH) {                     // StitchToPreviousBlock, :120-133
    g.st_first = g.blk_pos - 3u;
    g.st_count = 3;
    g.st_stride = 1;
  }
  if (J.hasher_type >= 40 && J.hasher_type <= 42) fc_drain_stores(fc_geom(J, g), g); else k_drain_stores(J, G, g);
  uint32_t bytes = g.blk_bytes, pos = g.blk_pos;
  if (g.blk_flags & QBLK_EXTEND) {                     // ExtendLastCommand, encode.c:905-971
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
DEV void parse_quick_round(const JobParams& J, const ShardDesc& D, ShardState* S, const DeviceTables* T,
                           const uint8_t* input, uint8_t* ws, const CompoundDict* cd = nullptr) {
  const int lane = wave_lane();
  const bool writer = lane == 0;
  const uint32_t htl = hasher_htl(J.hasher_type);
  const QuickGeom G = quick_geom(J);
  if (S->done || S->mb_valid || S->error) return;

  QShard g;
  g.data = input + D.in_off;
  g.table = ws + D.table_off;
  g.nums = nullptr;
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
  const bool chain = J.hasher_type >= 40 && J.hasher_type <= 42;             // H40 / H41 / H42
  const FcGeom F = fc_geom(J, g);

  while (g.state != Q_DONE) {       // all state is wave-uniform here
    if (g.state == Q_PRE) q_driver_pre(J, g);
    if (g.state == Q_SETUP) k_setup_block(J, G, g);
    if (g.state == Q_SEARCH && !(g.position + htl < g.pos_end)) {
      g.insert_length += g.pos_end - g.position;
      g.r.last_insert_len = g.insert_length;
      g.state = Q_POST;
    }
    if (g.state == Q_SEARCH || g.state == Q_LAZY) {
      const bool lazy = g.state == Q_LAZY;
      const uint32_t P = g.position + (lazy ? 1u : 0u);
      // the lazy probe only looks for something longer than what it has (:127-128)
      const uint32_t len_in = lazy ? umin(g.sr_len - 1u, g.pos_end - P) : 0u;
      QResult cur = chain ? fc_search(J, F, T, g, P) : k_search(J, G, T, g, P, len_in);
      // attached dictionaries: H2 and H54 have no dictionary variant (backward_references.c:194-243)
      if (g.cd && J.hasher_type != 2 && J.hasher_type != 54) q_compound_lookup(J, g, P, g.pos_end - P, cur);
      g.stat_searches++;
      bool commit = false;
      if (!lazy) {
        if (cur.score > K_MIN_SCORE) {
          g.sr_len = cur.len; g.sr_dist = cur.distance; g.sr_score = cur.score; g.sr_delta = cur.delta;
          g.delayed = 0;
          g.state = Q_LAZY;
        } else {
          ++g.insert_length;
          ++g.position;
          if (g.position > g.apply_random_heuristics) {     // literal spree, :207-237
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
      if (chain) fc_drain_stores(F, g); else k_drain_stores(J, G, g);
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
  }
  wave_sync();
}

#endif  // BROTLI_AMD_CSRC_K_PARSE_QUICK_H_
