// brotli_amd/csrc/k_parse4.h — K1, second generation: the LZ77 parse with FOUR
// encoder shards per wavefront (16 lanes each).
//
// Semantics: identical to k_parse.h / k_round.h (CreateBackwardReferences,
// c/enc/backward_references_inc.h:10-242; H68 / H58,
// c/enc/hash_longest_match64_simd_inc.h:114-302; static dictionary,
// c/enc/hash.h:140-202; EncodeData glue, c/enc/encode.c:905-1173).
//
// Why: with one shard per wave every decision is wave-uniform and lands on
// the scalar unit (one per CU, shared by 32 waves); rocprofv3 showed the first
// kernel bound by SALU issue (131 SALU + 53 VALU per input byte,
// profiles/r01_a_*).  Here a 16-lane group owns a shard: lane t of a group is
// slot t of the 16-slot bucket row, lanes 0..3 double as the distance-cache
// probes, and all encoder state is kept per lane (replicated inside a group),
// so one VALU instruction advances four encoders and the scalar unit only
// runs the loop skeleton.  A 128-byte bucket record is one coalesced access of
// a group; four groups keep four independent memory round trips in flight.
//
// Preconditions (checked by the host, hip_layer.hip): every shard is at most
// (1 << lgwin) - 16 bytes, so positions never wrap the ring and no candidate
// is ever farther than the window; longer shards use k_parse.h.
//
// The order-dependent candidate resolve of FindLongestMatch is evaluated as an
// arg-max over the group (exact whenever no candidate needs the reference's
// byte "gate" to decide, which is detected) with a step-by-step emulation as
// the fallback (resolve_slow), also reachable with JOB_FLAG_FORCE_SLOW so the
// tests can pin one against the other.
#ifndef BROTLI_AMD_CSRC_K_PARSE4_H_
#define BROTLI_AMD_CSRC_K_PARSE4_H_

#include "k_round.h"

#define Q_GROUPS 4
#if defined(BROTLI_AMD_SIMT_SIM)
extern unsigned long long g_sim_counts[16];
#define SIM_COUNT(i, n) do { if (wave_lane() == 0) g_sim_counts[i] += (n); } while (0)
#else
#define SIM_COUNT(i, n) do { } while (0)
#endif

// Optional phase timing (build with -DQ_PROFILE): shader-clock cycles per phase
// accumulated per shard into ShardState::prof[].
#if defined(Q_PROFILE) && !defined(BROTLI_AMD_SIMT_SIM)
#define QP_NOW() __builtin_amdgcn_s_memtime()
#define QP_ADD(g, i, t0) do { const uint64_t qp_n = QP_NOW(); (g).prof[i] += qp_n - (t0); (t0) = qp_n; } while (0)
#else
#define QP_NOW() 0ull
#define QP_ADD(g, i, t0) do { (void)(t0); } while (0)
#endif
#define Q_DUP_SLOTS 1024u
#define QREC_ENTRY(i) ((uint32_t)(i) * 8u)   // byte offset of entry i in a record (see q_entry)

#define QST_DONE 1u
#define QST_ERROR 2u
#define QST_HAVE_MB 4u
#define QBLK_LAST 1u
#define QBLK_FLUSH 2u
#define QBLK_STITCH 4u
#define QBLK_EXTEND 8u
#define QBLK_NOSEAL 16u     // flushed block without the padding block (final_op 3, see k_round.h)
enum QState { Q_PRE = 0, Q_SETUP = 1, Q_SEARCH = 2, Q_LAZY = 3, Q_POST = 4, Q_DONE = 5 };

// All fields are identical in the 16 lanes of a group.
struct QShard {
  // shard constants
  const uint8_t* data;
  uint8_t* table;
  uint16_t* nums;          // k_parse_deep.h: bucket counters (separate array)
  Command* cmds;
  uint32_t stream_offset;
  // cold shard facts (length, final operation, command capacity, output buffer) are
  // re-read from the descriptor when a block starts or ends: `descs` / `wsb` are
  // wave-uniform kernel arguments, `shard` is the only per-lane register they cost
  const ShardDesc* descs;
  uint8_t* wsb;
  uint32_t shard;
  // stream state (RoundRegs)
  RoundRegs r;
  int32_t dc[4];
  uint32_t dict_lookups, dict_matches;
  // current block
  uint32_t blk_flags;      // QBLK_*: cold per-block facts in one register
  uint32_t blk_bytes, blk_pos;
  uint32_t position, pos_end, store_end, insert_length, apply_random_heuristics;
  // lazy matching
  uint32_t sr_len, sr_dist, sr_score;
  int32_t sr_delta;
  uint32_t delayed;
  // pending ordered insertions: first + i * stride, i < count
  uint32_t st_first, st_count, st_stride;
  uint64_t st_x;          // bytes at st_first + t * st_stride, fetched ahead of the first step
  uint32_t st_x_valid;
  // the 32 bytes at the next search position, requested at the end of a step so
  // that they travel during the insertions (-DQ_NEXT32)
  B32 n32;
  uint32_t n32_pos;
  uint32_t state;
  uint32_t status;         // QST_*
  uint32_t stat_searches;
  // JOB_FLAG_DUO: two groups serve one shard with identical state; role 0 (primary)
  // searches the position the state machine asks for and owns every memory write,
  // role 1 (scout) searches the position after it in the same step
  uint32_t role;
  // software prefetch of bucket records: value of the load issued last step
  // (consumed one step later so nothing waits on it) and a sink that keeps the
  // loads alive
  uint32_t pf_val, pf_acc;
  uint64_t prof[12];
  // attached dictionaries (k_dict.h): only the one-shard-per-wave stream kernels set them
  const CompoundDict* cd = nullptr;
  uint32_t gap = 0;        // their total size (backward_references_inc.h:31)
  // k_chain.h, tiled jobs: the part of the shard this group parses (0xFFFFFFFF: what the descriptor says)
  uint32_t lim_len = 0xFFFFFFFFu, lim_op = 0xFFFFFFFFu, lim_cmd_cap = 0xFFFFFFFFu;
  uint32_t ring_mask = 0xFFFFFFFFu;   // a stream longer than the ring buffer: its mask (the rules of the ring's physical end, q_resolve_slow)
  uint32_t ring_mask_stream = 0;      // 1: the group parses a tile of a stream (JOB_FLAG_STREAMT)
  uint32_t no_cut = 0;     // a stream's tile (JOB_FLAG_STREAMT): the block always merges — where meta-blocks end is decided elsewhere
  // k_chain.h: commands are left raw (CMD_RAW, enc_types.h) with CMDF_* in dist_prefix
  uint32_t raw_cmds = 0, cmd_flags = 0;
  // ... and with the static-dictionary lookups / matches (hash.h:49-50) counted since the last command or block
  // start, so that a replay of the command (k_chain.h sweeps) keeps the two counters without searching
  uint32_t dict_mark_l = 0, dict_mark_m = 0;
};
#define CMDF_NOPROBE 1u    // the position behind the copy's start was not searched (the lazy chain ended on an advance)
#define CMDF_SPREE 2u      // the literal spree (backward_references_inc.h:208-236) skipped searches before this command
                           //   (or the counts below did not fit): a sweep parses this command again
#define CMDF_LOOKUPS_SHIFT 2u   // 8 bits: dictionary lookups, 4 bits from bit 10: matches, while the command was decided
#define CMDF_DELAYED_SHIFT 14u  // 2 bits: by how many positions the lazy matching moved the copy's start (3 = three or more)
DEV uint32_t q_dict_flags(QShard& g) {
  const uint32_t dl = g.dict_lookups - g.dict_mark_l, dm = g.dict_matches - g.dict_mark_m;
  g.dict_mark_l = g.dict_lookups;
  g.dict_mark_m = g.dict_matches;
  return (dl > 255u || dm > 15u) ? CMDF_SPREE : (dl << CMDF_LOOKUPS_SHIFT) | (dm << 10);
}

DEV const ShardDesc& q_desc(const QShard& g) { return g.descs[g.shard]; }
DEV uint32_t q_len(const QShard& g) { return g.lim_len != 0xFFFFFFFFu ? g.lim_len : q_desc(g).len; }
DEV uint32_t q_final_op(const QShard& g) { return g.lim_op != 0xFFFFFFFFu ? g.lim_op : q_desc(g).final_op; }
DEV uint32_t q_cmd_cap(const QShard& g) { return g.lim_cmd_cap != 0xFFFFFFFFu ? g.lim_cmd_cap : q_desc(g).cmd_cap; }
DEV uint8_t* q_out(const QShard& g) { return g.wsb + q_desc(g).out_off; }
DEV int q_t() { return wave_lane() & 15; }
DEV int q_base() { return wave_lane() & 48; }
DEV uint32_t q_mask16(uint64_t ballot) { return (uint32_t)(ballot >> q_base()) & 0xFFFFu; }
DEV bool wave_any(bool p) { return wave_ballot(p) != 0; }
DEV uint32_t q_bcast(uint32_t v, int t) { return wave_shfl(v, q_base() | t); }
// Maximum over the 16 lanes of a group, delivered to every lane of the group:
// four DPP row rotations (VALU only).
DEV uint32_t q_max(uint32_t v) {
  uint32_t o;
  o = wave_row_ror(v, 8); v = o > v ? o : v;
  o = wave_row_ror(v, 4); v = o > v ? o : v;
  o = wave_row_ror(v, 2); v = o > v ? o : v;
  o = wave_row_ror(v, 1); v = o > v ? o : v;
  return v;
}
DEV uint32_t q_or(uint32_t v) {
  v |= wave_row_ror(v, 8);
  v |= wave_row_ror(v, 4);
  v |= wave_row_ror(v, 2);
  v |= wave_row_ror(v, 1);
  return v;
}
// Value held by lane i of the group (v is an unsigned payload).
DEV uint32_t q_from(uint32_t v, int i) { return q_max(q_t() == i ? v : 0u); }
DEV uint16_t ld16(const uint8_t* p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }

// Byte the reference reads at ring index x <= pos_end: real data below pos_end; at pos_end the zero bytes written
// behind the block on the first lap (encode.c:879-893) or, once the ring has been lapped (a stream longer than the
// ring, k_chain.h), the stale byte of the lap before (ringbuffer.h:103-159).
DEV uint32_t q_ring_byte(const QShard& g, uint32_t x) {
  if (x < g.pos_end) return g.data[x];
  return g.pos_end > g.ring_mask ? g.data[x - g.ring_mask - 1u] : 0u;
}

// (Written as a conditional over the four members — not g.dc[i], and nothing may index g.dc through its address, k_dict.h's
//  compound_lookup included: one such index keeps the whole QShard in scratch memory, where every read of g.position is a
//  round trip.  k_parse_deep and k_parse_quick ran that way until round 6: .private_segment_fixed_size 464, half a
//  terabyte of scratch writes per GiB at quality 9, profiles/r06_pmc_q9_before.txt.  Reading all four entries first and
//  selecting between the values costs k_chain 2.8 ms per GiB — profiles/r06_g_ab_q_dc_entry.txt — and is not needed once
//  nothing pins the struct.)
DEV uint32_t q_dc_entry(const QShard& g, int i) {
  return (uint32_t)(i == 0 ? g.dc[0] : i == 1 ? g.dc[1] : i == 2 ? g.dc[2] : g.dc[3]);
}

// Match length of data[a..] and data[b..] beyond the first 32 bytes
// (find_match_length.h:19-40), limit = bytes available at a.
DEV uint32_t q_extend(const uint8_t* data, uint32_t a, uint32_t b, uint32_t limit) {
  uint32_t off = 32;
  while (off + 8 <= limit) {
    const uint64_t x = ld64(data + a + off) ^ ld64(data + b + off);
    if (x) return off + ((uint32_t)dev_ctz64(x) >> 3);
    off += 8;
  }
  while (off < limit && data[a + off] == data[b + off]) ++off;
  return off;
}

// Touches the 128-byte record of `key` (8 bytes per lane of the group) so that
// the search that will need it finds it in the L2 / L1.  The loaded value is
// folded into a sink one step later.  Measured on MI355X (profiles/
// r01_e_variants.log): with the memory system already saturated by request
// count, these prefetches cost more than they hide, so they are compiled in
// only with -DQ_NEXT_PREFETCH / -DQ_COMMIT_PREFETCH.
DEV void q_prefetch_record(QShard& g, bool act, uint32_t key) {
  g.pf_acc ^= g.pf_val;
  g.pf_val = 0;
  if (act) g.pf_val = ld32(g.table + (size_t)key * REC_BYTES + QREC_ENTRY(q_t()));
}

// ---- bucket records of this kernel ---------------------------------------------------
// 128 bytes per key, sixteen 8-byte entries {u32 position, u16 tag2, u8 tag,
// u8 aux}: lane t of a group owns entry t, so a probe is ONE dwordx2 load per
// lane (one coalesced 128-byte access per group) and an insertion is one
// 8-byte store.  The reference's 16-bit counter num_ lives in the aux bytes of
// entries 0 (low byte) and 1 (high byte).  Two alternatives were measured on
// MI355X and lost at the shard counts that matter (profiles/r01_g_*): a
// generation stamp in entries 2 / 3 that makes clearing lazy (first touches
// cost more than k_init saves above 64 KiB shards), and a separate dense
// counter array (a second HBM line per bucket update).  (k_parse.h keeps its own
// array-of-fields layout; k_init writes either, see init_shard_table.)
DEV uint64_t q_entry(uint32_t pos, uint32_t tag2, uint32_t tag, uint32_t aux) {
  return (uint64_t)pos | ((uint64_t)(tag2 & 0xFFFFu) << 32) | ((uint64_t)(tag & 0xFFu) << 48) |
         ((uint64_t)(aux & 0xFFu) << 56);
}

// ---- ordered insertion of up to 16 positions per group ---------------------------
// Store / StoreRange (..64_simd_inc.h:114-137).  `act`: this lane inserts `pos`.
DEV void q_store16(const JobParams& J, const QShard& g, bool act, uint32_t pos, bool have_x,
                   uint64_t x, uint8_t* lds_dup) {
  const int t = q_t();
  KeyTag kt;
  kt.key = 0; kt.tag = 0; kt.tag2 = 0;
  act = act && g.role == 0;
  if (act) kt = hash_pos(have_x ? x : ld64(g.data + pos), J.hasher_type, J.bucket_bits);
  // Two lanes of a group with one key must be ranked; detect that case through
  // an LDS scoreboard (false positives only cost time).
  uint8_t* sb = lds_dup + (size_t)(q_base() >> 4) * Q_DUP_SLOTS + (kt.key & (Q_DUP_SLOTS - 1u));
  if (act) *sb = (uint8_t)t;
  wave_sync();
  const bool dup = act && *sb != (uint8_t)t;
  const bool any_dup = wave_any(dup);
  if (any_dup) SIM_COUNT(6, 1);                        // insertion steps with a key collision
  uint8_t* rec = g.table + (size_t)kt.key * REC_BYTES;
  // entries 0 and 1 carry the counter
  uint64_t e0 = 0, e1 = 0;
  if (act) { e0 = ld64(rec); e1 = ld64(rec + 8); }
  const uint32_t num = (uint32_t)(e0 >> 56) | ((uint32_t)(e1 >> 56) << 8);
  wave_sync();
  uint32_t below = 0, total = 1;
  if (any_dup) {
    // same = lanes of my group that insert the same key
    uint64_t same = ~0ull;
    for (int b = 0; b < J.bucket_bits; ++b) {
      const bool bit = (kt.key >> b) & 1;
      const uint64_t m = wave_ballot(act && bit);
      same &= bit ? m : ~m;
    }
    same &= wave_ballot(act);
    const uint32_t same16 = q_mask16(same);
    below = (uint32_t)__builtin_popcount(same16 & ((1u << t) - 1u));
    total = (uint32_t)__builtin_popcount(same16);
  }
  if (act) {
    const uint32_t s = (num - below) & 15u;
    const uint32_t aux = s == 0 ? (uint32_t)(e0 >> 56) : s == 1 ? (uint32_t)(e1 >> 56) : 0u;
    const uint64_t e = q_entry(pos, kt.tag2, kt.tag, aux);
    __builtin_memcpy(rec + QREC_ENTRY(s), &e, 8);
  }
  wave_sync();   // the counter bytes go last (another lane may have rewritten entry 0 / 1)
  if (act && below + 1 == total) {
    const uint32_t nn = (num - total) & 0xFFFFu;
    rec[7] = (uint8_t)nn;
    if ((nn >> 8) != (num >> 8)) rec[15] = (uint8_t)(nn >> 8);
  }
  wave_sync();
}

// Drains every group's pending insertions.
DEV void q_drain_stores(const JobParams& J, QShard& g, uint8_t* lds_dup) {
  while (wave_any(g.st_count != 0)) {
    SIM_COUNT(3, 1);                                   // insertion steps (wave level)
    const uint32_t n = umin(g.st_count, 16u);
    const bool act = (uint32_t)q_t() < n;
    q_store16(J, g, act, g.st_first + (uint32_t)q_t() * g.st_stride, g.st_x_valid != 0, g.st_x, lds_dup);
    g.st_x_valid = 0;
    g.st_first += n * g.st_stride;
    g.st_count -= n;
  }
}

// ---- one FindLongestMatch per group ------------------------------------------------
struct QResult { uint32_t len, distance, score; int32_t delta; };

// Exact step-by-step emulation (..64_simd_inc.h:201-292) over the candidates
// the lanes hold: distance-cache entries in lanes 0..ndist-1, bucket slots in
// ring order starting at `head`.
DEV QResult q_resolve_slow(const QShard& g, bool want, uint32_t P, uint32_t max_length,
                           uint32_t head, int ndist, bool d_ok, uint32_t d_len, uint32_t d_prev,
                           uint32_t d_score, bool b_ok, uint32_t b_len, uint32_t b_prev,
                           uint32_t b_score) {
  QResult r;
  r.len = 0; r.distance = 0; r.score = K_MIN_SCORE; r.delta = 0;
  uint32_t best_len = 0;
  // the ring's physical end (:187-195, 243-249): nothing more is looked at once the current position is within
  // best_len of it, and a candidate that close to it is passed over
  const uint32_t rm = g.ring_mask, cur_masked = P & rm;
  bool stopped = false;
  for (int i = 0; i < ndist; ++i) {
    const bool ok = q_bcast(d_ok ? 1u : 0u, i) != 0;
    const uint32_t len_i = q_bcast(d_len, i), prev_i = q_bcast(d_prev, i), score_i = q_bcast(d_score, i);
    if (!want || !ok || stopped) continue;
    if (cur_masked + best_len > rm) { stopped = true; continue; }
    if ((prev_i & rm) + best_len > rm) continue;
    // The byte gate from the lengths already measured (len_i = the whole match, up to max_length): a match longer than
    // best_len agrees at best_len; one that ends AT best_len, inside the block, differs there.  Only a shorter one —
    // its first mismatch lies before the byte asked about — or one that ends with the block has to look.
    if (len_i <= best_len) {
      if (len_i == best_len && len_i < max_length) continue;
      if (q_ring_byte(g, P + best_len) != q_ring_byte(g, prev_i + best_len)) continue;
    }
    if (!(len_i >= 3 || (len_i == 2 && i < 2))) continue;
    if (!(r.score < score_i)) continue;
    best_len = len_i;
    r.len = len_i; r.distance = P - prev_i; r.score = score_i;
  }
  if (best_len < 3) best_len = 3;
  // The bucket slots in ring order starting at `head` (:246-292), replayed a winner at a time: given the best match
  // so far, every lane decides for ITS slot whether the walk would take it — the two rules of the ring's end, the
  // four gate bytes best_len - 3 .. best_len (:265-270), length, score — and the first such slot in visiting order is
  // the walk's next winner: the slots before it fail against the very state they would meet.  A round per winner
  // (two or three) instead of 16 dependent steps.  The gate from the lengths already measured (len = the whole match,
  // up to max_length): a match longer than best_len agrees on all four bytes; one that ends in best_len - 3 ..
  // best_len, inside the block, differs there; only a shorter one — its mismatch lies before the bytes asked about —
  // or one that ends with the block has to look: two 4-byte loads for all such slots of the group at once.  (With 16
  // equally long candidates — runs of zeros — the walk was 20 candidates x 8 dependent loads, 62 % of such a shard's
  // cycles: profiles/r04_h, r05.)
  {
    const int t = q_t();
    const uint32_t myj = ((uint32_t)t - head) & 15u;                  // this lane's slot is the myj-th the walk visits
    uint32_t next_j = 0;
    bool live = want;
    while (wave_any(live)) {
      if (live && cur_masked + best_len > rm) live = false;          // (:243-245: nothing more is looked at)
      bool elig = live && b_ok && myj >= next_j && b_len >= 4u && r.score < b_score && !((b_prev & rm) + best_len > rm);
      if (elig && b_len <= best_len) {
        if (b_len + 3u >= best_len && b_len < max_length) elig = false;
        else if (P + best_len < g.pos_end) elig = ld32(g.data + P + best_len - 3u) == ld32(g.data + b_prev + best_len - 3u);
        else {
          for (uint32_t k = best_len - 3; k <= best_len; ++k)
            if (q_ring_byte(g, P + k) != q_ring_byte(g, b_prev + k)) { elig = false; break; }
        }
      }
      const uint32_t first = q_max(elig ? 16u - myj : 0u);            // 16 - (the lowest visiting index of an eligible slot)
      if (first == 0u) live = false;
      const int src = (int)((head + (16u - first)) & 15u);
      const uint32_t len_w = q_bcast(b_len, src), prev_w = q_bcast(b_prev, src), score_w = q_bcast(b_score, src);
      if (live) {
        best_len = len_w;
        r.len = len_w; r.distance = P - prev_w; r.score = score_w;
        next_j = 16u - first + 1u;
      }
    }
  }
  return r;
}

// Static dictionary probe (hash.h:140-202): lanes 0 and 1 of a group take one
// hash slot each; acceptance is sequential (slot 0 first).  `want`: this
// group's search found nothing.
DEV void q_dict_search(const JobParams& J, const DeviceTables* T, QShard& g, bool want, uint32_t P,
                       uint32_t max_length, QResult& out) {
  const int t = q_t();
  const bool go = want && !(g.dict_matches < (g.dict_lookups >> 7));
  if (!wave_any(go)) return;
  SIM_COUNT(4, 1);                                     // dictionary probes (wave level)
  const int nprobes = (J.hasher_type < 5 || J.hasher_type == 54) ? 1 : 2;   // `shallow`, hash.h:187
  uint32_t matchlen = 0, wlen = 0, widx = 0;
  if (go && t < nprobes) {
    const uint32_t key = (((ld32(g.data + P) * 0x1E35A7BDu) >> (32 - 14)) << 1) + (uint32_t)t;
    wlen = T->dict_hash_lengths[key];
    widx = T->dict_hash_words[key];
    if (wlen != 0 && wlen <= max_length) {
      const uint8_t* w = T->dict + T->dict_offsets_by_length[wlen & 31] + wlen * widx;
      while (matchlen < wlen && g.data[P + matchlen] == w[matchlen]) ++matchlen;
    }
  }
  const uint32_t dictionary_start = umin(P + g.stream_offset, J.max_backward_limit) + g.gap;
  for (int i = 0; i < nprobes; ++i) {
    const uint32_t len = q_from(wlen, i), word_idx = q_from(widx, i), ml = q_from(matchlen, i);
    if (!go) continue;
    g.dict_lookups++;
    if (len == 0 || len > max_length) continue;
    if (ml + 10 <= len || ml == 0) continue;
    const uint32_t cut = len - ml;
    const uint32_t transform_id = (cut << 2) + (uint32_t)((0x071B520ADA2D3200ull >> (cut * 6)) & 0x3F);
    const uint32_t backward = dictionary_start + 1u + word_idx +
        (transform_id << T->dict_size_bits_by_length[len]);
    if (backward > 0x3FFFFFCu) continue;   // params->dist.max_distance
    const uint32_t score = 1920u + 135u * ml - 30u * log2floor(backward);
    if (score < out.score) continue;
    out.len = ml;
    out.delta = (int32_t)len - (int32_t)ml;
    out.distance = backward;
    out.score = score;
    g.dict_matches++;
  }
}

// One shard per wave (k_parse_deep.h, k_parse_quick.h): the attached-dictionary lookup that follows
// FindLongestMatch (backward_references_inc.h:115-119, 147-152) and the dictionary branch of
// ExtendLastCommand (encode.c:930-961).
DEV void q_compound_lookup(const JobParams& J, const QShard& g, uint32_t P, uint32_t max_length, QResult& r) {
  if (!g.cd) return;
  SearchResult sr;
  sr.len = r.len; sr.distance = r.distance; sr.score = r.score; sr.len_code_delta = r.delta;
  compound_lookup(g.cd, g.data + P, P & J.ring_mask, J.ring_mask, g.dc[0], g.dc[1], g.dc[2], g.dc[3], max_length,
                  umin(P + g.stream_offset, J.max_backward_limit), K_DIST_MAX_DISTANCE, sr);
  r.len = sr.len; r.distance = sr.distance; r.score = sr.score; r.delta = sr.len_code_delta;
}
// The same for the 16-lane groups of k_parse4 (four shards per wave; `want` per group).
DEV void q_compound_lookup16(const JobParams& J, const QShard& g, bool want, uint32_t P, uint32_t max_length, QResult& r,
                             const DictAhead& ahead) {
  if (!g.cd) return;                                   // (a kernel argument: the same for the whole wave)
  SearchResult sr;
  sr.len = r.len; sr.distance = r.distance; sr.score = r.score; sr.len_code_delta = r.delta;
  compound_lookup16(g.cd, want, g.data + (want ? P : 0u), P & J.ring_mask, J.ring_mask, g.dc[0], g.dc[1], g.dc[2], g.dc[3],
                    max_length, umin(P + g.stream_offset, J.max_backward_limit), K_DIST_MAX_DISTANCE, sr, ahead);
  if (want) { r.len = sr.len; r.distance = sr.distance; r.score = sr.score; r.delta = sr.len_code_delta; }
}
DEV void q_compound_extend(const QShard& g, Command& last, uint32_t cmd_dist, uint32_t max_distance,
                           uint32_t& bytes, uint32_t& pos) {
  if (!g.cd || cmd_dist <= max_distance) return;
  const uint32_t gained = compound_extend(g.cd, g.data + pos, bytes, cmd_dist, max_distance, last.copy_len & 0x1FFFFFFu);
  last.copy_len += gained;
  bytes -= gained;
  pos += gained;
}

// What the insertion of the searched position needs from its search (per lane).
// Packed (the scout keeps it across the primary's transition and the kernel sits at its
// 128-VGPR budget): key | tag << 16; tag2 | num << 16; the lane's own entry for the
// counter-byte rewrite: position, tag2 | tag << 16.
struct QIns { uint32_t key_tag, tag2_num, slot_e, e_tags; };

// insert P (:293-295): the lane that owns ring slot num & 15 writes the new
// entry; lanes 0 / 1 refresh the counter bytes kept in their entries.
DEV void q_insert(QShard& g, bool act, uint32_t P, const QIns& in) {
  const int t = q_t();
  const uint32_t num = in.tag2_num >> 16;
  const uint32_t ts = num & 15u, nn = (num - 1u) & 0xFFFFu;
  const bool mine = (uint32_t)t == ts;
  const bool hi_changed = (nn >> 8) != (num >> 8);
  if (act && (mine || t == 0 || (t == 1 && hi_changed))) {
    const uint32_t naux = t == 0 ? (nn & 0xFFu) : t == 1 ? (nn >> 8) : 0u;
    const uint64_t e = mine ? q_entry(P, in.tag2_num & 0xFFFFu, in.key_tag >> 16, naux)
                            : q_entry(in.slot_e, in.e_tags & 0xFFFFu, in.e_tags >> 16, naux);
    __builtin_memcpy(g.table + (size_t)(in.key_tag & 0xFFFFu) * REC_BYTES + QREC_ENTRY(t), &e, 8);
  }
}

// Searches position P for every group with want set (hash table and distance
// cache only); the caller inserts P (q_insert) and runs the dictionary probe.
DEV QResult q_search(const JobParams& J, const DeviceTables* T, QShard& g, bool want, uint32_t P, QIns& ins,
                     DictAhead* dict_ahead = nullptr) {
  const int t = q_t();
  const int ndist = J.ndist;
  const uint32_t max_length = g.pos_end - P;
  const uint32_t max_backward = umin(P, J.max_backward_limit);
  uint64_t qt = QP_NOW();
  // No branches around the first-round loads, so that the compiler can count
  // them precisely (s_waitcnt vmcnt(n)): idle lanes read offset 0.
#if !defined(Q_NO_NEXT32)
  B32 cur32 = g.n32;
  if (wave_any(want && g.n32_pos != P)) {
    if (want && g.n32_pos != P) cur32 = load_b32(g.data + P);
  }
#else
  const B32 cur32 = load_b32(g.data + (want ? P : 0u));
#endif
  // The distance-cache probes (:201-240) depend only on P and the cache, not on
  // the hash table: their strings are requested right away, in the same round
  // trip as the bytes at P and overlapping the bucket-record access.
  const uint32_t backward = q_dc_entry(g, t);
  const bool d_cand = want && t < ndist && (int32_t)backward > 0 && backward <= max_backward;
  const uint32_t d_prev = P - backward;
  const B32 pd = load_b32(g.data + (d_cand ? d_prev : (want ? P : 0u)));
  const KeyTag kt = hash_pos(cur32.q[0], J.hasher_type, J.bucket_bits);
  // attached dictionaries: the first chunk's key range and items travel beside the bucket record (k_dict.h)
  if (dict_ahead != nullptr && g.cd != nullptr) *dict_ahead = dict_ahead16(g.cd, want && g.role == 0, cur32.q[0]);
#if defined(Q_PROFILE)
  if (wave_any(want && kt.key == 0x7FFFFFFFu)) g.pf_acc++;   // (profiling fence: bytes at P consumed)
#endif
  QP_ADD(g, 8, qt);
  const uint8_t* rec = g.table + (size_t)kt.key * REC_BYTES;
  const uint64_t ent = ld64((want ? rec : g.table) + QREC_ENTRY(t));
  const uint32_t slot = (uint32_t)ent;
  const uint32_t tag2 = (uint32_t)(ent >> 32) & 0xFFFFu;
  const uint32_t tag = (uint32_t)(ent >> 48) & 0xFFu;
  const uint32_t aux = (uint32_t)(ent >> 56);
  // aux bytes of entries 0 and 1 = the counter, gathered with one reduction
  const uint32_t num = q_or(t < 2 ? aux << (8 * t) : 0u);
  const uint32_t head = (num + 1u) & 15u;
  const uint32_t n = (65535u - num) & 0xFFFFu;
  const uint32_t logical = ((uint32_t)t - head) & 15u;   // 0 = newest
#if defined(Q_PROFILE)
  if (wave_any(want && tag == 0x1234567u)) g.pf_acc++;   // (profiling fence: record values consumed)
#endif
  QP_ADD(g, 0, qt);
  // bucket candidate of this lane (:246-262)
  const bool b_cand = want && (n >= 16u || logical < n) && tag == kt.tag && tag2 == kt.tag2 &&
                      (P - slot) <= max_backward;
  const uint32_t b_prev = slot;
  // Bucket candidates need a second round trip (their positions come from the
  // record); most searches on text have at most one after the tag2 filter.
  uint32_t b_len = 0, d_len = 0;
  bool b_ext = false, d_ext = false;
  {
    B32 pb;
    pb.q[0] = pb.q[1] = pb.q[2] = pb.q[3] = 0;
    if (b_cand) pb = load_b32(g.data + b_prev);
#if defined(Q_NEXT_PREFETCH)
    {
      const uint64_t x1 = (cur32.q[0] >> 8) | (cur32.q[1] << 56);
      const KeyTag k1 = hash_pos(x1, J.hasher_type, J.bucket_bits);
      q_prefetch_record(g, want, k1.key);
    }
#endif
    const uint32_t md = common_prefix32(cur32, pd);
#if defined(Q_PROFILE)
    if (wave_any(want && md == 77u)) g.pf_acc++;   // (profiling fence: distance-cache strings consumed)
#endif
    QP_ADD(g, 9, qt);
    const uint32_t mb = common_prefix32(cur32, pb);
#if defined(Q_PROFILE)
    if (wave_any(want && mb == 77u)) g.pf_acc++;   // (profiling fence: bucket candidate strings consumed)
#endif
    QP_ADD(g, 10, qt);
    if (b_cand) { b_len = umin(mb, max_length); b_ext = mb == 32u && max_length > 32u; }
    if (d_cand) { d_len = umin(md, max_length); d_ext = md == 32u && max_length > 32u; }
  }
  SIM_COUNT(0, 1);                                     // search steps (wave level)
  if (wave_any(b_cand)) SIM_COUNT(1, 1);               // steps with a bucket candidate load
  if (wave_any(b_ext || d_ext)) {
    SIM_COUNT(2, 1);                                   // steps with a > 32 byte extension
    if (b_ext) b_len = q_extend(g.data, P, b_prev, max_length);
    if (d_ext) d_len = q_extend(g.data, P, d_prev, max_length);
  }
#if defined(Q_PROFILE)
  if (wave_any(b_len + d_len == 0xFFFFFFFFu)) g.pf_acc++;   // (profiling fence: candidate bytes consumed)
#endif
  QP_ADD(g, 1, qt);
  // scores (hash.h:123-138)
  const uint32_t b_score = 1920u + 135u * b_len - 30u * log2floor((P - b_prev) | 1u);
  uint32_t d_score = 135u * d_len + 1935u;
  if (t != 0) d_score -= 39u + ((0x1CA10u >> ((uint32_t)t & 0xEu)) & 0xEu);
  const bool b_ok = b_cand && b_len >= 4u;
  const bool d_ok = d_cand && (d_len >= 3u || (d_len == 2u && t < 2));

  // Arg-max with the reference's order as the tie break: distance cache
  // entries 0..3 first, then bucket slots newest to oldest.
  const uint32_t d_key = d_ok ? (d_score << 5) | (31u - (uint32_t)t) : 0u;
  const uint32_t b_key = b_ok ? (b_score << 5) | (27u - logical) : 0u;
  const uint32_t d_best = q_max(d_key);
  const uint32_t dc_score = d_best ? (d_best >> 5) : K_MIN_SCORE;
  // length of the distance-cache winner (needed for the gate test below)
  const uint32_t dc_len = q_max((d_key != 0 && d_key == d_best) ? d_len : 0u);
  const uint32_t dc_len3 = dc_len < 3u ? 3u : dc_len;
  // A bucket candidate that beats the distance-cache winner without being
  // longer depends on the byte gate: take the exact path for that group.
  const bool unsure = b_ok && b_score > dc_score && b_len <= dc_len3;
  const bool force_slow = (J.flags & JOB_FLAG_FORCE_SLOW) != 0;
  const bool slow = q_mask16(wave_ballot(unsure)) != 0 || (force_slow && want);
  const uint32_t best = q_max(b_key > d_key ? b_key : d_key);
  // the winner's payload: keys are unique inside a group, so exactly one lane
  // (or none) contributes a non-zero value to each reduction
  const bool win_is_d = best != 0 && d_key == best;
  const bool win_is_b = best != 0 && b_key == best;
  QResult r;
  r.len = q_max(win_is_d ? d_len : win_is_b ? b_len : 0u);
  r.distance = q_max(win_is_d ? backward : win_is_b ? P - b_prev : 0u);
  r.score = best >> 5;
  r.delta = 0;
  if (best == 0 || r.score <= K_MIN_SCORE) { r.len = 0; r.distance = 0; r.score = K_MIN_SCORE; }
  if (wave_any(slow)) {
    SIM_COUNT(5, 1);                                   // exact resolves
    const QResult s = q_resolve_slow(g, want, P, max_length, head, ndist, d_ok || (d_cand), d_len,
                                     d_prev, d_score, b_cand, b_len, b_prev, b_score);
    if (slow) r = s;
  }
  QP_ADD(g, 2, qt);
  ins.key_tag = kt.key | (kt.tag << 16);
  ins.tag2_num = kt.tag2 | (num << 16);
  ins.slot_e = slot;
  ins.e_tags = tag2 | (tag << 16);
  (void)T;
  return r;
}

// ---- stream driver pieces (k_round.h's parse_round, per lane) ------------------------
DEV void q_flush_padding(QShard& g, bool writer) {
  RoundRegs& r = g.r;
  if (r.last_bytes_bits != 0) {
    const uint32_t seal = r.last_bytes | (0x6u << r.last_bytes_bits);
    const uint32_t seal_bits = r.last_bytes_bits + 6u;
    const uint32_t nb = (seal_bits + 7u) >> 3;
    if (writer) for (uint32_t i = 0; i < nb; ++i) q_out(g)[r.out_bytes + i] = (uint8_t)(seal >> (8u * i));
    r.out_bytes += nb;
    r.last_bytes = 0;
    r.last_bytes_bits = 0;
  }
  if (r.flint == -1) r.flint = -2;
}

// Loop top of BrotliEncoderCompressStream up to the start of EncodeData.
DEV void q_driver_pre(const JobParams& J, QShard& g) {
  RoundRegs& r = g.r;
  const uint32_t block = 1u << J.lgblock;
  const uint32_t htl = hasher_htl(J.hasher_type);
  for (;;) {
    const uint32_t avail = q_len(g) - r.input_pos;
    const uint32_t d = r.input_pos - r.last_processed_pos;
    uint32_t remaining = d >= block ? 0u : block - d;
    if (r.flint >= 0 && remaining > (uint32_t)r.flint) remaining = (uint32_t)r.flint;
    if (remaining != 0 && avail != 0) {
      const uint32_t n = umin(remaining, avail);
      r.input_pos += n;
      if (r.flint > 0) r.flint -= (int32_t)n;
      continue;
    }
    if (remaining != 0 && avail == 0 && q_final_op(g) == 0) { g.status |= QST_DONE; g.state = Q_DONE; return; }
    bool is_last = avail == 0 && q_final_op(g) == 2;
    bool force_flush = avail == 0 && (q_final_op(g) == 1 || q_final_op(g) == 3);
    bool seal = avail == 0 && q_final_op(g) == 1;
    if (!is_last && r.flint == 0) { r.flint = -1; force_flush = true; seal = true; }
    const uint32_t bytes = r.input_pos - r.last_processed_pos;
    const uint32_t pos = r.last_processed_pos;
    if (r.ncmds + bytes / 2u + 2u > q_cmd_cap(g)) { g.status |= QST_ERROR; g.state = Q_DONE; return; }
    g.blk_flags = (is_last ? QBLK_LAST : 0u) | (force_flush ? QBLK_FLUSH : 0u) |
                  (force_flush && !seal ? QBLK_NOSEAL : 0u);
    g.blk_bytes = bytes;
    g.blk_pos = pos;
    g.pos_end = r.input_pos;
    if (bytes >= htl - 1u && pos >= 3u) g.blk_flags |= QBLK_STITCH;
    if (r.ncmds != 0 && r.last_insert_len == 0) g.blk_flags |= QBLK_EXTEND;
    g.state = Q_SETUP;
    return;
  }
}

// After CreateBackwardReferences of a block: merge / cut decision (encode.c:1141-1216).
DEV void q_driver_post(const JobParams& J, QShard& g, bool writer) {
  RoundRegs& r = g.r;
  const uint32_t block = 1u << J.lgblock;
  const bool is_last = (g.blk_flags & QBLK_LAST) != 0, force_flush = (g.blk_flags & QBLK_FLUSH) != 0;
  const uint32_t avail = q_len(g) - r.input_pos;
  {
    const uint32_t processed = r.input_pos - r.last_flush_pos;
    const bool next_fits = processed + block <= J.max_metablock_size;
    // without block splitting a meta-block is cut as soon as enough symbols have gathered (:1150-1153)
    const bool should_flush = J.flush_symbols != 0u && r.nlits + r.ncmds >= J.flush_symbols;
    if (g.no_cut != 0u || (!is_last && !force_flush && !should_flush && next_fits && r.nlits < J.max_literals && r.ncmds < J.max_commands)) {
      // (a block without bytes and without an operation to carry out would come back here
      // forever: an unknown final_op — fail instead of spinning)
      if (g.blk_bytes == 0 && avail == 0) { g.status |= QST_ERROR | QST_DONE; g.state = Q_DONE; return; }
      r.last_processed_pos = r.input_pos;
      g.state = Q_PRE;
      return;
    }
  }
  if (r.last_insert_len > 0) {
    if (writer) g.cmds[r.ncmds] = make_insert_command(r.last_insert_len);
    ++r.ncmds;
    r.nlits += r.last_insert_len;
    r.last_insert_len = 0;
  }
  if (!is_last && r.input_pos == r.last_flush_pos) {
    r.last_processed_pos = r.input_pos;
    if (force_flush && !(g.blk_flags & QBLK_NOSEAL)) q_flush_padding(g, writer);
    if (avail == 0) { g.status |= QST_DONE; g.state = Q_DONE; } else g.state = Q_PRE;
    return;
  }
  const uint32_t mbytes = r.input_pos - r.last_flush_pos;
  if (mbytes <= 2u) {
    // ShouldCompress() is false for <= 2 bytes (encode.c:461): flint blocks
    // and tiny tails are written right here by the group's lane 0.
    BitWriter w;
    bw_init(w, q_out(g) + r.out_bytes, r.last_bytes_bits, r.last_bytes);
    if (mbytes == 0) {
      bw_put(w, 2, 3);
      bw_align_byte(w);
    } else {
      for (int i = 0; i < 4; ++i) g.dc[i] = r.saved_dc[i];
      // BrotliStoreUncompressedMetaBlock (brotli_bit_stream.c:1321-1352)
      bw_put(w, 1, 0);
      bw_put(w, 2, 0);
      bw_put(w, 16, mbytes - 1u);
      bw_put(w, 1, 1);
      bw_align_byte(w);
      for (uint32_t i = 0; i < mbytes; ++i) bw_put(w, 8, g.data[r.last_flush_pos + i]);
      if (is_last) {
        bw_put(w, 1, 1);
        bw_put(w, 1, 1);
        bw_align_byte(w);
      }
    }
    // flush whole bytes; only the group's writer lane touches memory
    while (w.nacc >= 8) {
      if (writer) w.out[w.byte_pos] = (uint8_t)w.acc;
      w.byte_pos += 1;
      w.acc >>= 8;
      w.nacc -= 8;
    }
    r.out_bytes += w.byte_pos;
    r.last_bytes = (uint32_t)(w.acc & 0xFF);
    r.last_bytes_bits = w.nacc;
    r.last_flush_pos = r.input_pos;
    r.last_processed_pos = r.input_pos;
    if (r.last_flush_pos > 0) r.prev_byte = g.data[r.last_flush_pos - 1];
    if (r.last_flush_pos > 1) r.prev_byte2 = g.data[r.last_flush_pos - 2];
    r.ncmds = 0;
    r.nlits = 0;
    for (int i = 0; i < 4; ++i) r.saved_dc[i] = g.dc[i];
    if (force_flush && !(g.blk_flags & QBLK_NOSEAL)) q_flush_padding(g, writer);
    if (is_last || avail == 0) { g.status |= QST_DONE; g.state = Q_DONE; } else g.state = Q_PRE;
    return;
  }
  g.status |= QST_HAVE_MB;
  g.state = Q_DONE;
}

// ExtendLastCommand (encode.c:905-971) for the groups with want set; then the
// CreateBackwardReferences prologue.
DEV void q_setup_extend(const JobParams& J, QShard& g, bool want);
DEV void q_setup_block(const JobParams& J, QShard& g, bool want, uint8_t* lds_dup) {
  // StitchToPreviousBlock (..64_simd_inc.h:139-151)
  if (want && (g.blk_flags & QBLK_STITCH)) {
    g.st_first = g.blk_pos - 3u;
    g.st_count = 3;
    g.st_stride = 1;
  }
  q_drain_stores(J, g, lds_dup);
  q_setup_extend(J, g, want);
}
// ExtendLastCommand + the CreateBackwardReferences prologue (shared with k_chain.h).
DEV void q_setup_extend(const JobParams& J, QShard& g, bool want) {
  const int t = q_t();
  const uint32_t htl = hasher_htl(J.hasher_type);
  uint32_t bytes = g.blk_bytes, pos = g.blk_pos;
  bool ext = false, dext = false;
  uint32_t cmd_dist = 0, dext_max = 0;
  Command last;
  last.insert_len = last.copy_len = last.dist_extra = 0;
  last.cmd_prefix = last.dist_prefix = 0;
  const bool try_ext = want && (g.blk_flags & QBLK_EXTEND) != 0;
  if (try_ext) {
    last = g.cmds[g.r.ncmds - 1];
    const uint32_t last_copy_len = last.copy_len & 0x1FFFFFFu;
    const uint32_t lpp = g.r.last_processed_pos - last_copy_len;
    const uint32_t max_distance = umin(lpp, J.max_backward_limit);
    cmd_dist = (uint32_t)g.dc[0];
    uint32_t distance_code;
    const uint32_t dcode = last.dist_prefix & 0x3FFu;
    if (last.cmd_prefix == CMD_RAW) {
      distance_code = last.dist_extra;
    } else if (dcode < 16) {
      distance_code = dcode;
    } else {
      const uint32_t nbits = last.dist_prefix >> 10;
      const uint32_t hcode = dcode - 16u;
      const uint32_t offset = ((2u + (hcode & 1u)) << nbits) - 4u;
      distance_code = offset + last.dist_extra + 16u;
    }
    const bool code_ok = distance_code < 16u || distance_code - 15u == cmd_dist;
    ext = code_ok && g.dc[0] > 0 && cmd_dist <= max_distance;
    dext = code_ok && !ext && cmd_dist > max_distance;     // (encode.c:930-961: the copy continues inside the attached dictionary)
    dext_max = max_distance;
    if (!code_ok) last.insert_len = 0xFFFFFFFFu;   // marker: leave the command alone
  }
  if (g.cd) {
    const uint32_t gained = compound_extend16(g.cd, dext, g.data + pos, bytes, cmd_dist, dext_max, last.copy_len & 0x1FFFFFFu);
    if (dext) { last.copy_len += gained; bytes -= gained; pos += gained; }
  }
  bool running = ext;
  while (wave_any(running)) {
    const bool ok = running && (uint32_t)t < bytes &&
        g.data[pos + (uint32_t)t] == g.data[pos + (uint32_t)t - cmd_dist];
    const uint32_t m16 = q_mask16(wave_ballot(ok));
    const uint32_t run = (m16 == 0xFFFFu) ? 16u : (uint32_t)dev_ctz32(~m16);
    if (running) {
      last.copy_len += run;
      bytes -= run;
      pos += run;
      if (run < 16u || bytes == 0) running = false;
    }
  }
  if (try_ext && last.insert_len != 0xFFFFFFFFu) {
    if (last.cmd_prefix != CMD_RAW)
    last.cmd_prefix = (uint16_t)combine_length_codes(insert_length_code(last.insert_len),
        copy_length_code((uint32_t)((int)(last.copy_len & 0x1FFFFFFu) + (int)(last.copy_len >> 25))),
        (last.dist_prefix & 0x3FF) == 0);
    if (t == 0 && g.role == 0) g.cmds[g.r.ncmds - 1] = last;
  }
  wave_sync();
  if (want) {
    g.position = pos;
    g.pos_end = pos + bytes;
    g.store_end = bytes >= htl ? g.pos_end - htl + 1u : pos;
    g.insert_length = g.r.last_insert_len;
    g.apply_random_heuristics = pos + J.spree_window;
    g.dict_mark_l = g.dict_lookups;
    g.dict_mark_m = g.dict_matches;
    g.state = Q_SEARCH;
  }
}

// Commits the pending match g.sr_* at g.position (:165-206): StoreRange bounds (left pending in
// st_*), distance cache, command.  Shared with k_chain.h.
DEV void q_commit(const JobParams& J, QShard& g, bool commit, uint32_t htl) {
  const int t = q_t();
  (void)t; (void)htl;
  if (!commit) return;
  g.state = Q_SEARCH;
  // StoreRange bounds first, so that the bytes of the first insertion
  // step (and the next search position's key) travel while the command
  // is being built.
  uint32_t range_start = g.position + 2u;
  const uint32_t range_end = umin(g.position + g.sr_len, g.store_end);
  if (g.sr_dist < (g.sr_len >> 2)) {
    range_start = umin(range_end, umax(range_start, g.position + g.sr_len - (g.sr_dist << 2)));
  }
  if (range_start < range_end) {
    g.st_first = range_start;
    g.st_count = range_end - range_start;
    g.st_stride = 1;
#if defined(Q_EARLY_STX)
    g.st_x = ld64(g.data + range_start + (uint32_t)t);
    g.st_x_valid = 1;
#endif
  }
  g.apply_random_heuristics = g.position + 2u * g.sr_len + J.spree_window;
  const uint32_t dictionary_start = umin(g.position + g.stream_offset, J.max_backward_limit) + g.gap;
  const uint32_t distance_code = compute_distance_code(g.sr_dist, dictionary_start, g.dc);
  if (g.sr_dist <= dictionary_start && distance_code > 0) {
    g.dc[3] = g.dc[2]; g.dc[2] = g.dc[1]; g.dc[1] = g.dc[0]; g.dc[0] = (int32_t)g.sr_dist;
  }
  if (g.raw_cmds) {
    if (t == 0 && g.role == 0) {
      Command c;
      c.insert_len = g.insert_length;
      c.copy_len = g.sr_len | (((uint32_t)(uint8_t)(int8_t)g.sr_delta) << 25);
      c.dist_extra = distance_code; c.cmd_prefix = CMD_RAW; c.dist_prefix = (uint16_t)(g.cmd_flags | q_dict_flags(g) | (umin(g.delayed, 3u) << CMDF_DELAYED_SHIFT));
      g.cmds[g.r.ncmds] = c;
    } else (void)q_dict_flags(g);
    g.cmd_flags = 0;
  } else if (t == 0 && g.role == 0) g.cmds[g.r.ncmds] = make_command(g.insert_length, g.sr_len, g.sr_delta, distance_code);
  ++g.r.ncmds;
  g.r.nlits += g.insert_length;
  g.insert_length = 0;
  g.position += g.sr_len;
}

// One step of the CreateBackwardReferences state machine (:44-242) for the groups in
// `act`, given the search result `cur` of the position they were waiting for.
// Returns whether a command was committed (the copied range is then pending in st_*).
DEV bool q_transition(const JobParams& J, QShard& g, bool act, const QResult& cur, uint32_t htl) {
  const int t = q_t();
  (void)t;
  bool commit = false;
  if (act && g.state == Q_SEARCH) {
    if (cur.score > K_MIN_SCORE) {
      g.sr_len = cur.len; g.sr_dist = cur.distance; g.sr_score = cur.score; g.sr_delta = cur.delta;
      g.delayed = 0;
      g.state = Q_LAZY;
    } else {
      ++g.insert_length;
      ++g.position;
      if (g.position > g.apply_random_heuristics) {
        // literal spree (:208-236): sparse insertions, no searches
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
          g.cmd_flags |= CMDF_SPREE;
          g.position += cnt * step;
          g.insert_length += cnt * step;
        }
      }
    }
  } else if (act && g.state == Q_LAZY) {
    commit = true;
    if (cur.score >= g.sr_score + 175u) {
      ++g.position;
      ++g.insert_length;
      g.sr_len = cur.len; g.sr_dist = cur.distance; g.sr_score = cur.score; g.sr_delta = cur.delta;
      if (++g.delayed < 4 && g.position + htl < g.pos_end) commit = false;
      else g.cmd_flags |= CMDF_NOPROBE;
    }
  }
  q_commit(J, g, commit, htl);
  return commit;
}

DEV uint32_t q_groups_per_wave(const JobParams& J) {
  const uint32_t v = (J.flags >> JOB_FLAG_GROUPS_SHIFT) & 3u;
  return v ? v : (uint32_t)Q_GROUPS;
}

// ---- the kernel body: up to four shards per wave ---------------------------------------
DEV void parse4_round(const JobParams& J, const ShardDesc* shards, ShardState* states,
                      uint32_t nshards, const DeviceTables* T, const uint8_t* input, uint8_t* ws,
                      uint32_t wave_index, uint8_t* lds_dup, const CompoundDict* cd = nullptr) {
  const int t = q_t();
  const uint32_t gpw = q_groups_per_wave(J);
  const bool duo = (J.flags & JOB_FLAG_DUO) != 0;          // gpw <= 2: a shard owns two groups
  const uint32_t gi = (uint32_t)(wave_lane() >> 4) >> (duo ? 1 : 0);
  const uint32_t shard = wave_index * gpw + gi;
  const bool alive = gi < gpw && shard < nshards;
  const uint32_t role = duo ? ((uint32_t)(wave_lane() >> 4) & 1u) : 0u;
  const bool writer = alive && t == 0 && role == 0;
  const uint32_t htl = hasher_htl(J.hasher_type);
  const ShardDesc& D = shards[alive ? shard : 0];
  const ShardState* S0 = &states[alive ? shard : 0];

  QShard g;
  g.data = input + D.in_off;
  g.table = ws + D.table_off;
  g.cmds = (Command*)(ws + D.cmds_off);
  g.descs = shards;
  g.wsb = ws;
  g.shard = alive ? shard : 0u;
  g.stream_offset = D.stream_offset;
  g.cd = cd;
  g.gap = cd ? cd->total_size : 0u;
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
  g.role = role;
  g.pf_val = g.pf_acc = 0;
  for (int i = 0; i < 12; ++i) g.prof[i] = 0;
  g.state = (alive && !S0->done && !S0->mb_valid && !S0->error) ? Q_PRE : Q_DONE;
  const bool participated = g.state != Q_DONE;

  while (wave_any(g.state != Q_DONE)) {
    uint64_t qt0 = QP_NOW();
    // stream driver up to the next block
    if (g.state == Q_PRE) q_driver_pre(J, g);
    if (wave_any(g.state == Q_SETUP)) q_setup_block(J, g, g.state == Q_SETUP, lds_dup);
    QP_ADD(g, 7, qt0);

    // block finished? (loop guard of CreateBackwardReferences, :44 and :239-241)
    if (g.state == Q_SEARCH && !(g.position + htl < g.pos_end)) {
      g.insert_length += g.pos_end - g.position;
      g.r.last_insert_len = g.insert_length;
      g.state = Q_POST;
    }
    const bool want = g.state == Q_SEARCH || g.state == Q_LAZY;
    if (wave_any(want)) {
      const uint32_t P0 = g.position + (g.state == Q_LAZY ? 1u : 0u);
      const uint32_t P = P0 + g.role;
      QIns ins;
      DictAhead dahead;
      dahead.s0 = dahead.n = dahead.off0 = dahead.off1 = 0;
      QResult mine = q_search(J, T, g, want, P, ins, &dahead);
      uint64_t qt = QP_NOW();
      q_insert(g, want && g.role == 0, P, ins);
      wave_sync();
      QP_ADD(g, 3, qt);
      // static dictionary when nothing was found (hash.h:179-202); its two counters are
      // stream state, so the scout probes after the primary and continues from its counts
      q_dict_search(J, T, g, want && g.role == 0 && mine.score == K_MIN_SCORE, P, g.pos_end - P, mine);
      // attached dictionaries (backward_references_inc.h:115-119, 147-152; the planner gives such jobs no scout groups)
      q_compound_lookup16(J, g, want && g.role == 0, P, g.pos_end - P, mine, dahead);
      QResult r0 = mine, r1 = mine;
      bool r1_ok = false;
      uint32_t dl1 = 0, dm1 = 0;
      if (duo) {
        const int lane = wave_lane();
        const uint32_t dl0 = wave_shfl(g.dict_lookups, lane & ~16), dm0 = wave_shfl(g.dict_matches, lane & ~16);
        if (g.role == 1) { g.dict_lookups = dl0; g.dict_matches = dm0; }
        q_dict_search(J, T, g, want && g.role == 1 && mine.score == K_MIN_SCORE, P, g.pos_end - P, mine);
        QResult other;
        other.len = wave_shfl(mine.len, lane ^ 16);
        other.distance = wave_shfl(mine.distance, lane ^ 16);
        other.score = wave_shfl(mine.score, lane ^ 16);
        other.delta = (int32_t)wave_shfl((uint32_t)mine.delta, lane ^ 16);
        const uint32_t other_key = wave_shfl(ins.key_tag & 0xFFFFu, lane ^ 16);
        r0 = g.role == 0 ? mine : other;
        r1 = g.role == 0 ? other : mine;
        dl1 = wave_shfl(g.dict_lookups, lane | 16);
        dm1 = wave_shfl(g.dict_matches, lane | 16);
        g.dict_lookups = dl0;
        g.dict_matches = dm0;
        r1_ok = (ins.key_tag & 0xFFFFu) != other_key;      // equal keys: the scout read the bucket before P went in
      }
      QP_ADD(g, 4, qt);
      if (want) g.stat_searches++;
      bool commit = q_transition(J, g, want, r0, htl);
      if (duo) {
        // second step: the position the scout searched is the one the state machine
        // asks for next, its bucket was not touched by the primary's insertion, and
        // nothing is pending in between
        const bool st2 = want && r1_ok && !commit && g.st_count == 0 &&
            (g.state == Q_SEARCH ? (g.position == P0 + 1u && g.position + htl < g.pos_end)
                                 : (g.state == Q_LAZY && g.position == P0));
        if (wave_any(st2)) {
          q_insert(g, st2 && g.role == 1, P0 + 1u, ins);   // the scout's own position, now that its search counts
          wave_sync();
          if (st2) { g.dict_lookups = dl1; g.dict_matches = dm1; g.stat_searches++; }
          commit = q_transition(J, g, st2, r1, htl) || commit;
        }
      }
      // After a copy the next search position is known: fetch its record
      // while the copied range is being inserted.
#if defined(Q_COMMIT_PREFETCH)
      if (wave_any(commit)) {
        KeyTag kn;
        kn.key = 0;
        const bool pf = commit && g.position + htl < g.pos_end;
        if (pf) kn = hash_pos(ld64(g.data + g.position), J.hasher_type, J.bucket_bits);
        q_prefetch_record(g, pf, kn.key);
      }
#endif
#if !defined(Q_NO_NEXT32)
      {
        // the next search position is settled: SEARCH at position, LAZY at position + 1
        const bool nx = g.state == Q_SEARCH || g.state == Q_LAZY;
        const uint32_t pn = g.position + (g.state == Q_LAZY ? 1u : 0u);
        g.n32 = load_b32(g.data + (nx ? pn + g.role : 0u));
        g.n32_pos = nx ? pn + g.role : 0xFFFFFFFFu;
      }
#endif
      QP_ADD(g, 5, qt);
      q_drain_stores(J, g, lds_dup);
      QP_ADD(g, 6, qt);
    }
    if (g.state == Q_POST) q_driver_post(J, g, writer);
  }

  wave_sync();
  if (writer && participated) {
    ShardState* S = &states[shard];      // (recomputed: not worth two registers across the loop)
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
    S->stat_b_used = g.pf_acc ^ g.pf_val;   // keeps the prefetch loads observable
#if defined(Q_PROFILE)
    for (int i = 0; i < 12; ++i) S->prof[i] += g.prof[i];
#endif
  }
  wave_sync();
}

#endif  // BROTLI_AMD_CSRC_K_PARSE4_H_
