// brotli_amd/csrc/k_wide.h — a LONG meta-block built and written by many waves.
//
// k_build / k_store give a meta-block one wave: right for the 8192 shards of 128 KiB of a partition plan, 50 ms for
// the one 4 MiB meta-block of a stock one-shot call and 111 ms for the 8 MiB meta-blocks of a stream (measured:
// profiles/r04_a_stock_256MiB_stages.log) — almost all of it work that is linear in the commands: the literal and
// distance symbol streams (metablock.c:741-769), the per-literal codes, the command stream's bits
// (brotli_bit_stream.c:1073-1111).  Here that work is cut into PARTS of WIDE_CMD_PART commands / WIDE_LIT_PART
// literals; what a part needs to know about the parts before it (first literal, source position, first distance,
// bit offset) is an exclusive scan over the parts' totals, so every linear phase is "totals per part -> scan ->
// write per part", WIDE_K waves per meta-block.  What is sequential by definition stays on one wave per meta-block,
// with the three greedy block splitters (metablock_inc.h:48-183) side by side on a wave each; the prefix codes are
// independent jobs spread over the waves.  The pieces are those of k_build.h / k_store.h (store_commands<true>,
// build_streams_range, ...): the bits are the same whichever way a meta-block is written, and the simulator tests
// run both ways against the oracle.
//
// Kernels, in launch order (kernels.h; "x K" = grid of nmb * K waves, K = JobArgs::wide_k <= WIDE_K_MAX, wave k of a meta-block takes the parts
// k, k + K, ...):
//   k_wide_head      ShouldCompress, literal-context decision, output capacity        (build_round's first steps)
//   k_wide_count xK  per command part: literals, bytes, distance symbols
//   k_wide_scan1     offsets of the parts; nlits / ndist of the meta-block
//   k_wide_streams xK  lits[] / dsym[] of a part
//   k_wide_split x3  the literal / command / distance block splitter
//   k_wide_prep      histogram smoothing, small-code histograms, the literal context map
//   k_wide_codes xK  zeroes a slice of the output; prefix-code jobs j = k, k + K, ...
//   k_wide_header    block-switch codes, the meta-block header                        (where the command stream starts)
//   k_wide_bits xK   per literal part: codes + running bit sums; per command part: bits of its command / distance codes
//   k_wide_scan2     bit offsets of the literal parts and of the command parts
//   k_wide_emit xK   the command stream of a part at its bit offset (shared edge dwords by atomic OR)
//   k_wide_tail      size check, raw fallback, state update                           (store_finish)
#ifndef BROTLI_AMD_CSRC_K_WIDE_H_
#define BROTLI_AMD_CSRC_K_WIDE_H_

#include "k_store.h"

#define WIDE_K_MAX 64u                 // waves per meta-block in the part kernels, at most (JobArgs::wide_k)

struct WideMb {
  MbInfo* info;
  WidePart* parts;
  uint32_t* lpart_bits;                // [max_lparts]
  uint32_t* lpart_off;                 // [max_lparts]
  uint32_t nparts, nlparts;
};
DEV bool wide_live(const ShardState* S) { return S->mb_valid != 0 && S->error == 0; }
DEV void wide_views(WideMb& w, uint8_t* mb, const MbLayout& L, uint32_t ncmds, uint32_t nlits) {
  w.info = (MbInfo*)(mb + L.info);
  w.parts = (WidePart*)(mb + L.parts);
  w.lpart_bits = (uint32_t*)(mb + L.lparts);
  w.lpart_off = w.lpart_bits + L.max_lparts;
  w.nparts = (ncmds + WIDE_CMD_PART - 1u) / WIDE_CMD_PART;
  w.nlparts = (nlits + WIDE_LIT_PART - 1u) / WIDE_LIT_PART;
}
// Sum of v over the wave (uniform).
DEV uint32_t wide_sum(uint32_t v) { return wave_bcast(wave_incl_scan(v), 63); }

DEV void wide_head(const JobParams& J, const ShardDesc& D, ShardState* S, const DeviceTables* T, const uint8_t* input,
                   uint8_t* ws, uint32_t* lds, double* lds_ent, double* lds_last, double* lds_terms) {
  const int lane = wave_lane();
  if (!wide_live(S)) return;
  BuildCtx b;
  build_ctx_init(b, J, D, T, input, ws, lds, lds_ent, lds_last, lds_terms);
  const uint32_t start = S->mb_start, bytes = S->mb_bytes, ncmds = S->ncmds, nlits_state = S->nlits;
  MbInfo* info = (MbInfo*)(b.mb + b.L.info);
  // (store_round's capacity check, before anybody writes)
  const bool stream = (J.flags & JOB_FLAG_STREAMT) != 0;
  if (S->out_bytes + 2ull * bytes + 520ull + 16ull > D.out_cap) {
    wave_sync();
    if (lane == 0) S->error = 2;
    wave_sync();
    return;
  }
  (void)stream;
  if (!should_compress(b, start, bytes, nlits_state, ncmds)) {
    if (lane == 0) S->mb_raw = 1;
    wave_sync();
    return;
  }
  decide_contexts(b, start, bytes);
  if (lane == 0) {
    info->num_contexts = b.nc;
    info->map_kind = b.map_kind;
    info->ncmds = ncmds;
  }
  wave_sync();
}

DEV void wide_count(const JobParams& J, const ShardDesc& D, const ShardState* S, uint8_t* ws, uint32_t k, uint32_t K) {
  const int lane = wave_lane();
  if (!wide_live(S) || S->mb_raw) return;
  MbLayout L;
  mb_layout(umin(D.len, J.max_metablock_size), &L);
  WideMb w;
  const uint32_t ncmds = S->ncmds;
  wide_views(w, ws + D.mb_off, L, ncmds, 0u);
  const Command* cmds = (const Command*)(ws + D.cmds_off);
  for (uint32_t p = k; p < w.nparts; p += K) {
    const uint32_t c0 = p * WIDE_CMD_PART, c1 = umin(c0 + WIDE_CMD_PART, ncmds);
    uint32_t ins = 0, adv = 0, nd = 0;
    for (uint32_t i = c0 + (uint32_t)lane; i < c1; i += 64u) {
      const Command c = cmds[i];
      const uint32_t cpy = c.copy_len & 0x1FFFFFFu;
      ins += c.insert_len;
      adv += c.insert_len + cpy;
      nd += (cpy != 0 && c.cmd_prefix >= 128) ? 1u : 0u;
    }
    ins = wide_sum(ins); adv = wide_sum(adv); nd = wide_sum(nd);
    if (lane == 0) { w.parts[p].ins = ins; w.parts[p].adv = adv; w.parts[p].ndist = nd; }
  }
  wave_sync();
}

DEV void wide_scan1(const JobParams& J, const ShardDesc& D, const ShardState* S, uint8_t* ws) {
  const int lane = wave_lane();
  if (!wide_live(S) || S->mb_raw) return;
  MbLayout L;
  mb_layout(umin(D.len, J.max_metablock_size), &L);
  WideMb w;
  wide_views(w, ws + D.mb_off, L, S->ncmds, 0u);
  uint32_t lit = 0, pos = 0, dist = 0;
  for (uint32_t p0 = 0; p0 < w.nparts; p0 += 64u) {
    const uint32_t p = p0 + (uint32_t)lane;
    const bool have = p < w.nparts;
    const uint32_t a = have ? w.parts[p].ins : 0u, b = have ? w.parts[p].adv : 0u, c = have ? w.parts[p].ndist : 0u;
    const uint32_t ia = wave_incl_scan(a), ib = wave_incl_scan(b), ic = wave_incl_scan(c);
    if (have) { w.parts[p].lit_off = lit + ia - a; w.parts[p].pos_off = pos + ib - b; w.parts[p].dist_off = dist + ic - c; }
    lit += wave_bcast(ia, 63); pos += wave_bcast(ib, 63); dist += wave_bcast(ic, 63);
  }
  if (lane == 0) { w.info->nlits = lit; w.info->ndist = dist; }
  wave_sync();
}

DEV void wide_streams(const JobParams& J, const ShardDesc& D, const ShardState* S, const DeviceTables* T,
                      const uint8_t* input, uint8_t* ws, uint32_t k, uint32_t K, uint32_t* lds) {
  if (!wide_live(S) || S->mb_raw) return;
  BuildCtx b;
  build_ctx_init(b, J, D, T, input, ws, lds, nullptr, nullptr, nullptr);
  WideMb w;
  const uint32_t ncmds = S->ncmds;
  wide_views(w, b.mb, b.L, ncmds, 0u);
  b.nc = w.info->num_contexts;
  b.map_kind = w.info->map_kind;
  for (uint32_t p = k; p < w.nparts; p += K) {
    const uint32_t c0 = p * WIDE_CMD_PART, c1 = umin(c0 + WIDE_CMD_PART, ncmds);
    uint32_t nl, nd;
    build_streams_range(b, c0, c1, S->mb_start + w.parts[p].pos_off, w.parts[p].lit_off, w.parts[p].dist_off, &nl, &nd, true);
  }
}

// cat 0 literals, 1 commands, 2 distances: one wave each, side by side.
DEV void wide_split(const JobParams& J, const ShardDesc& D, const ShardState* S, const DeviceTables* T, const uint8_t* input,
                    uint8_t* ws, uint32_t cat, uint32_t* lds, double* lds_ent, double* lds_last, double* lds_terms) {
  if (!wide_live(S) || S->mb_raw) return;
  BuildCtx b;
  build_ctx_init(b, J, D, T, input, ws, lds, lds_ent, lds_last, lds_terms);
  const MbInfo* info = (const MbInfo*)(b.mb + b.L.info);
  b.nc = info->num_contexts;
  b.map_kind = info->map_kind;
  if (cat == 0) run_splitter<0>(b, info->nlits);
  else if (cat == 1) run_splitter<1>(b, info->ncmds);
  else run_splitter<2>(b, info->ndist);
}

DEV void wide_prep(const JobParams& J, const ShardDesc& D, const ShardState* S, const DeviceTables* T, const uint8_t* input,
                   uint8_t* ws, uint32_t* lds, double* lds_ent, double* lds_last, double* lds_terms) {
  const int lane = wave_lane();
  if (!wide_live(S) || S->mb_raw) return;
  BuildCtx b;
  build_ctx_init(b, J, D, T, input, ws, lds, lds_ent, lds_last, lds_terms);
  MbInfo* info = (MbInfo*)(b.mb + b.L.info);
  if (J.quality >= 4) build_smooth_histograms(b, info);       // encode.c:587: from quality 4 on
  StoreCtx s;
  StoreMeta M;
  store_ctx_init(s, M, J, D, input, ws);
  uint32_t nrle = 0, maxp = 0;
  store_small_histos(s, M, lds, nrle, maxp);
  if (lane == 0) { info->cmap_nrle = nrle; info->cmap_max_prefix = maxp; }
  wave_sync();
}

DEV void wide_codes(const JobParams& J, const ShardDesc& D, const ShardState* S, const uint8_t* input, uint8_t* ws,
                    uint32_t k, uint32_t K, uint32_t* lds_store) {
  const int lane = wave_lane();
  if (!wide_live(S) || S->mb_raw) return;
  {
    // this wave's slice of everything the meta-block can touch (store_round's zero_output, spread over the waves)
    uint8_t* out = ws + D.out_off;
    const uint64_t from = S->out_bytes, bytes = 2ull * S->mb_bytes + 520ull;
    const uint64_t head = (4u - (from & 3u)) & 3u;
    if (k == 0 && (uint64_t)lane < head) out[from + (uint64_t)lane] = 0;
    uint32_t* p = (uint32_t*)(out + from + head);
    const uint64_t nw = (bytes - head + 3) >> 2;
    const uint64_t per = (nw + K - 1u) / K;
    const uint64_t lo = (uint64_t)k * per, hi = lo + per < nw ? lo + per : nw;
    for (uint64_t i = lo + (uint64_t)lane; i < hi; i += 64) p[i] = 0;
  }
  StoreCtx s;
  StoreMeta M;
  store_ctx_init(s, M, J, D, input, ws);
  const uint32_t maxp = s.info->cmap_max_prefix;
  for (uint32_t j = k; j < M.njobs; j += K) store_code_job(s, M, j, maxp, lds_store);
  wave_sync();
}

DEV void wide_header(const JobParams& J, const ShardDesc& D, const ShardState* S, const uint8_t* input, uint8_t* ws) {
  const int lane = wave_lane();
  if (!wide_live(S) || S->mb_raw) return;
  StoreCtx s;
  StoreMeta M;
  store_ctx_init(s, M, J, D, input, ws);
  MbInfo* info = (MbInfo*)(s.mb + s.L.info);
  store_switch_codes(s, M);
  uint8_t* out = ws + D.out_off;
  BitSink sink;
  sink.base = (uint32_t*)(out + (S->out_bytes & ~(uint64_t)3));
  sink.bitpos = (S->out_bytes & 3) * 8;
  const uint64_t bit0 = sink.bitpos;
  store_header(s, M, sink, S->last_bytes, S->last_bytes_bits, S->mb_bytes, S->mb_is_last != 0, info->cmap_nrle, info->cmap_max_prefix);
  if (lane == 0) { info->wide_bit0 = (uint32_t)bit0; info->wide_bit_cmds = (uint32_t)sink.bitpos; }
  wave_sync();
}

// Bits of the command and distance codes of command i (what store_commands calls `own`).
DEV uint32_t wide_own_bits(const StoreCtx& s, uint32_t i, const Command& c, bool has_dist, uint32_t my_dist) {
  const SymBits cb = symbol_bits<1>(s, i, c.cmd_prefix, 0);
  const uint32_t copylen_code = cmd_copy_len_code(c);
  const uint32_t inscode = insert_length_code(c.insert_len);
  const uint32_t copycode = copy_length_code(copylen_code);
  uint32_t n = cb.nsw + cb.ncode + k_ins_extra[inscode] + k_copy_extra[copycode];
  if (has_dist) {
    const SymBits db = symbol_bits<2>(s, my_dist, c.dist_prefix & 0x3FFu, 0);
    n += db.nsw + db.ncode + (c.dist_prefix >> 10);
  }
  return n;
}

DEV void wide_bits(const JobParams& J, const ShardDesc& D, const ShardState* S, const uint8_t* input, uint8_t* ws, uint32_t k, uint32_t K) {
  const int lane = wave_lane();
  if (!wide_live(S) || S->mb_raw) return;
  StoreCtx s;
  StoreMeta M;
  store_ctx_init(s, M, J, D, input, ws);
  WideMb w;
  const uint32_t nlits = s.info->nlits;
  wide_views(w, s.mb, s.L, M.ncmds, nlits);
  for (uint32_t lp = k; lp < w.nlparts; lp += K) {
    const uint32_t k0 = lp * WIDE_LIT_PART, k1 = umin(k0 + WIDE_LIT_PART, nlits);
    const uint32_t t = store_literal_codes(s, k0, k1, 0u);
    if (lane == 0) w.lpart_bits[lp] = t;
  }
  for (uint32_t p = k; p < w.nparts; p += K) {
    const uint32_t c0 = p * WIDE_CMD_PART, c1 = umin(c0 + WIDE_CMD_PART, M.ncmds);
    uint32_t dist_base = w.parts[p].dist_off, bits = 0;
    for (uint32_t base = c0; base < c1; base += 64u) {
      const uint32_t i = base + (uint32_t)lane;
      const bool valid = i < c1;
      Command c;
      c.insert_len = 0; c.copy_len = 0; c.dist_extra = 0; c.cmd_prefix = 0; c.dist_prefix = 0;
      if (valid) c = s.cmds[i];
      const bool has_dist = valid && (c.copy_len & 0x1FFFFFFu) != 0 && c.cmd_prefix >= 128;
      const uint64_t dm = wave_ballot(has_dist);
      const uint32_t my_dist = dist_base + (uint32_t)dev_popc64(dm & ((1ull << lane) - 1ull));
      if (valid) bits += wide_own_bits(s, i, c, has_dist, my_dist);
      dist_base += (uint32_t)dev_popc64(dm);
    }
    bits = wide_sum(bits);
    if (lane == 0) w.parts[p].bits = bits;
  }
  wave_sync();
}

DEV void wide_scan2(const JobParams& J, const ShardDesc& D, const ShardState* S, uint8_t* ws) {
  const int lane = wave_lane();
  if (!wide_live(S) || S->mb_raw) return;
  MbLayout L;
  mb_layout(umin(D.len, J.max_metablock_size), &L);
  WideMb w;
  MbInfo* info = (MbInfo*)(ws + D.mb_off + L.info);
  wide_views(w, ws + D.mb_off, L, info->ncmds, info->nlits);
  uint32_t run = 0;
  for (uint32_t p0 = 0; p0 < w.nlparts; p0 += 64u) {
    const uint32_t p = p0 + (uint32_t)lane;
    const uint32_t v = p < w.nlparts ? w.lpart_bits[p] : 0u;
    const uint32_t incl = wave_incl_scan(v);
    if (p < w.nlparts) w.lpart_off[p] = run + incl - v;
    run += wave_bcast(incl, 63);
  }
  if (lane == 0) info->wide_lit_bits = run;
  run = 0;
  for (uint32_t p0 = 0; p0 < w.nparts; p0 += 64u) {
    const uint32_t p = p0 + (uint32_t)lane;
    const uint32_t v = p < w.nparts ? w.parts[p].bits : 0u;
    const uint32_t incl = wave_incl_scan(v);
    if (p < w.nparts) w.parts[p].bit_off = run + incl - v;
    run += wave_bcast(incl, 63);
  }
  if (lane == 0) info->wide_cmd_bits = run;
  wave_sync();
}

DEV void wide_emit(const JobParams& J, const ShardDesc& D, const ShardState* S, const uint8_t* input, uint8_t* ws, uint32_t k,
                   uint32_t K, uint32_t* lds_store) {
  if (!wide_live(S) || S->mb_raw) return;
  StoreCtx s;
  StoreMeta M;
  store_ctx_init(s, M, J, D, input, ws);
  WideMb w;
  wide_views(w, s.mb, s.L, M.ncmds, s.info->nlits);
  LitSums LS;
  LS.lsum = s.lsum; LS.part_off = w.lpart_off; LS.nlits = s.info->nlits; LS.total = s.info->wide_lit_bits;
  uint32_t* sink_base = (uint32_t*)(ws + D.out_off + (S->out_bytes & ~(uint64_t)3));
  const uint64_t bit_cmds = s.info->wide_bit_cmds;
  for (uint32_t p = k; p < w.nparts; p += K) {
    const uint32_t c0 = p * WIDE_CMD_PART, c1 = umin(c0 + WIDE_CMD_PART, M.ncmds);
    (void)store_commands<true>(s, sink_base, bit_cmds, c0, c1, w.parts[p].bit_off, w.parts[p].lit_off, w.parts[p].dist_off, LS, lds_store);
  }
}

DEV void wide_tail(const JobParams& J, const ShardDesc& D, ShardState* S, const uint8_t* input, uint8_t* ws) {
  const int lane = wave_lane();
  if (!wide_live(S)) return;
  const bool stream = (J.flags & JOB_FLAG_STREAMT) != 0;
  const bool raw = S->mb_raw != 0;
  if (stream && raw) {            // (as store_round: k_stream_scan / k_stream_place emit a stream's raw meta-blocks)
    wave_sync();
    if (lane == 0) { S->mb_was_raw = 1; S->mb_valid = 0; }
    wave_sync();
    return;
  }
  const uint8_t* data = input + D.in_off;
  uint8_t* out = ws + D.out_off;
  RoundRegs r;
  regs_load(r, S);
  int32_t dc[4];
  for (int i = 0; i < 4; ++i) dc[i] = S->dist_cache[i];
  uint64_t total_bits = 0;
  if (!raw) {
    MbLayout L;
    mb_layout(umin(D.len, J.max_metablock_size), &L);
    const MbInfo* info = (const MbInfo*)(ws + D.mb_off + L.info);
    uint64_t end = (uint64_t)info->wide_bit_cmds + info->wide_cmd_bits + info->wide_lit_bits;
    if (S->mb_is_last != 0 && !stream) end = (end + 7u) & ~(uint64_t)7u;
    total_bits = end - info->wide_bit0;
  }
  store_finish(J, D, S, r, dc, data, out, raw, total_bits, 2ull * S->mb_bytes + 520ull);
}

#endif  // BROTLI_AMD_CSRC_K_WIDE_H_
