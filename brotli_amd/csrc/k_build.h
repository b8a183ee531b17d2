// brotli_amd/csrc/k_build.h — K3/K4/K5: per-shard meta-block modelling by one
// wavefront: literal-context decision, symbol streams, greedy block splitting
// with histograms, and the RLE-friendly count smoothing.
//
// Semantics (bit-exact): ShouldCompress (c/enc/encode.c:457-483),
// DecideOverLiteralContextModeling (encode.c:278-455),
// BrotliBuildMetaBlockGreedy (c/enc/metablock.c:463-839,
// c/enc/metablock_inc.h:48-183), BrotliBitsEntropy (c/enc/bit_cost.c:18-44),
// BrotliOptimizeHistograms (metablock.c:841-859,
// c/enc/entropy_encode.c:241-370).
//
// Design: the reference streams symbols one by one through three splitter
// objects.  Block decisions only happen every min_block_size symbols, so the
// wave (i) materialises the literal and distance symbol streams with wave
// scans over the command list, (ii) histograms one chunk of min_block_size
// symbols at a time into LDS with all 64 lanes (ds_add), and (iii) at a
// decision point evaluates all 3 x num_contexts entropies in parallel, one
// lane per histogram, each lane summing in the reference's index order so the
// doubles are identical.  The running block lives in LDS; finished block-type
// histograms live in HBM.
#ifndef BROTLI_AMD_CSRC_K_BUILD_H_
#define BROTLI_AMD_CSRC_K_BUILD_H_

#include "device_common.h"
#include "mb_layout.h"

#if defined(B_PROFILE) && !defined(BROTLI_AMD_SIMT_SIM)
#define BP_NOW() __builtin_amdgcn_s_memtime()
#define BP_ADD(S, i, t0) do { const uint64_t bp_n = BP_NOW(); if (wave_lane() == 0) (S)->prof[i] += bp_n - (t0); (t0) = bp_n; } while (0)
#else
#define BP_NOW() 0ull
#define BP_ADD(S, i, t0) do { (void)(t0); } while (0)
#endif

#define LDS_ROW(A) ((A) + 1u)   // padded row: lanes walking different rows hit different banks
#define BUILD_LDS_WORDS (13u * 257u)
#define BUILD_TERMS 768u          // 3 x 256 (three literal evaluations at once) >= 704

// Static literal context maps (encode.c:283-295, 347-364); index = map_kind.
static __device__ const uint8_t k_ctx_maps[4][64] = {
    {0},
    {1, 1, 2, 2},   // continuation bytes: 3 contexts
    {0, 0, 1, 1},   // simple UTF-8: 2 contexts
    {11, 11, 12, 12, 0, 0, 0, 0, 1, 1, 9, 9, 2, 2, 2, 2, 1, 1, 1, 1, 8, 3, 3, 3,
     1, 1, 1, 1, 2, 2, 2, 2, 8, 4, 4, 4, 8, 7, 4, 4, 8, 0, 0, 0, 3, 3, 3, 3,
     5, 5, 10, 5, 5, 5, 10, 5, 6, 6, 6, 6, 6, 6, 6, 6}};

// ---- entropy (summation order of the reference) ------------------------------
// BitsEntropy of a[k] + g[k] (either pointer may be null), bit_cost.c:18-44.
DEV double bits_entropy2(const uint32_t* a, const uint32_t* g, uint32_t n, const double* lut) {
  // One lane walks a whole histogram; the sum has to be taken in index order (the doubles are the reference's), but
  // the loads need not wait for it: eight counts and their eight log2 values are fetched at once, then added in
  // order.  A zero count needs no branch: lut[0] = 0, and r - 0.0 == r.  (The loop used to take one L1 round trip
  // per entry for lut[p]: 37 us per decision of the literal splitter with 13 contexts.)
  uint64_t sum = 0;
  double r = 0.0;
  uint32_t k = 0;
  for (; k + 8u <= n; k += 8u) {
    uint32_t p[8];
    double l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) p[j] = (a ? a[k + j] : 0u) + (g ? g[k + j] : 0u);
#pragma unroll
    for (int j = 0; j < 8; ++j) l[j] = lut[p[j]];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sum += p[j]; r -= (double)p[j] * l[j]; }
  }
  for (; k < n; ++k) {
    const uint32_t p = (a ? a[k] : 0u) + (g ? g[k] : 0u);
    sum += p;
    r -= (double)p * lut[p];
  }
  if (sum) r += (double)sum * lut[sum];
  if (r < (double)sum) r = (double)sum;
  return r;
}
// EstimateEntropy, encode.c:258-269.
DEV double estimate_entropy(const uint32_t* a, uint32_t n, const double* lut) {
  uint64_t total = 0;
  double r = 0.0;
  for (uint32_t k = 0; k < n; ++k) {
    const uint32_t p = a[k];
    total += p;
    r += (double)p * lut[p];
  }
  return (double)total * lut[total] - r;
}

// BitsEntropy (bit_cost.c:18-44) of `ne` histograms at once, whole wave: every
// lane computes the p * log2(p) terms of a strided share of the entries into
// LDS (independent, coalesced loads), then lane e < ne adds the terms of
// histogram e in index order — the doubles are those of the reference's loop.
// Histogram e is a[e][k] + g[e][k] (either may be null); results in out[e].
DEV void bits_entropy_wave(const uint32_t* const* a, const uint32_t* const* g, uint32_t ne, uint32_t n,
                           const double* lut, double* terms, uint32_t* sums, double* out) {
  const int lane = wave_lane();
  if ((uint32_t)lane < ne) sums[lane] = 0;
  wave_sync();
#pragma unroll
  for (uint32_t e = 0; e < 3u; ++e) {          // (ne <= 3; unrolled: a[] / g[] stay in registers)
    if (e >= ne) break;
    uint32_t part = 0;
    for (uint32_t k = (uint32_t)lane; k < n; k += 64) {
      const uint32_t p = (a[e] ? a[e][k] : 0u) + (g[e] ? g[e][k] : 0u);
      terms[e * n + k] = (double)p * lut[p];
      part += p;
    }
    if (part) lds_atomic_add(&sums[e], part);   // exact integer total of the histogram
  }
  wave_sync();
  if ((uint32_t)lane < ne) {
    const double* t = terms + (uint32_t)lane * n;
    double r = 0.0;
    for (uint32_t k = 0; k < n; ++k) r -= t[k];
    const uint32_t sum = sums[lane];
    if (sum) r += (double)sum * lut[sum];
    if (r < (double)sum) r = (double)sum;
    out[lane] = r;
  }
  wave_sync();
}

struct BuildCtx {
  const JobParams* J;
  const uint8_t* data;     // shard byte 0
  const DeviceTables* T;
  uint8_t* mb;             // MetaBlockWork
  MbLayout L;
  const Command* cmds;
  uint16_t* lits;
  uint16_t* dsym;
  uint32_t* lds;           // BUILD_LDS_WORDS
  double* lds_ent;         // [3] results of one bits_entropy_wave call
  double* lds_ent3;        // [3 * 13] entropies of the current decision
  uint32_t* lds_sums;      // [3]
  double* lds_last;        // [2 * 13] last_entropy
  double* lds_terms;       // [BUILD_TERMS] p * log2(p) terms of the histograms being evaluated
  uint32_t nc, map_kind;
};

// ---- ShouldCompress (encode.c:457-483) ---------------------------------------
DEV bool should_compress(BuildCtx& b, uint32_t start, uint32_t bytes, uint32_t nlits,
                         uint32_t ncmds) {
  const int lane = wave_lane();
  if (bytes <= 2) return false;
  if (ncmds < (bytes >> 8) + 2) {
    if ((double)nlits > 0.99 * (double)bytes) {
      for (uint32_t k = (uint32_t)lane; k < 256; k += 64) b.lds[k] = 0;
      wave_sync();
      const uint32_t t = (bytes + 12u) / 13u;
      for (uint32_t i = (uint32_t)lane; i < t; i += 64) lds_atomic_add(&b.lds[b.data[start + 13u * i]], 1u);
      wave_sync();
      const double thr = (double)bytes * 7.92 * (1.0 / 13.0);
      const double e = bits_entropy2(b.lds, nullptr, 256, b.T->log2_lut);
      wave_sync();
      if (e > thr) return false;
    }
  }
  return true;
}

// ---- DecideOverLiteralContextModeling (encode.c:278-455) ----------------------
DEV void decide_contexts(BuildCtx& b, uint32_t start, uint32_t length) {
  const int lane = wave_lane();
  const JobParams& J = *b.J;
  const double* lut2 = b.T->log2_lut;
  b.nc = 1;
  b.map_kind = 0;
  if (J.quality < 5 || length < 64 || (J.flags & JOB_FLAG_NO_LITCTX) != 0) return;
  const uint32_t end = start + length;
  const uint8_t* clut = b.T->context_lut;
  if (J.size_hint >= (1u << 20)) {
    // ShouldUseComplexStaticContextMap, :342-420: 64-byte strides every 4 KiB.
    for (uint32_t k = (uint32_t)lane; k < 32u * 14u; k += 64) b.lds[k] = 0;
    wave_sync();
    uint32_t total = 0;
    for (uint32_t sp = start; sp + 64 <= end; sp += 4096) {
      if (lane >= 2) {
        const uint32_t pos = sp + (uint32_t)lane;
        const uint32_t literal = b.data[pos];
        const uint32_t ctx = k_ctx_maps[3][clut[b.data[pos - 1]] | clut[256 + b.data[pos - 2]]];
        lds_atomic_add(&b.lds[literal >> 3], 1u);
        lds_atomic_add(&b.lds[32u + (ctx << 5) + (literal >> 3)], 1u);
      }
      total += 62;
    }
    wave_sync();
    double e1 = estimate_entropy(b.lds, 32, lut2);
    double e2 = 0.0;
    for (uint32_t i = 0; i < 13; ++i) e2 += estimate_entropy(b.lds + 32u + (i << 5), 32, lut2);
    const double inv = 1.0 / (double)total;
    e1 *= inv;
    e2 *= inv;
    wave_sync();
    if (!(e2 > 3.0 || e1 - e2 < 0.2)) {
      b.nc = 13;
      b.map_kind = 3;
      return;
    }
  }
  // ChooseContextMap, :278-338: bigram histogram of the top two bits.
  for (uint32_t k = (uint32_t)lane; k < 16; k += 64) b.lds[k] = 0;
  wave_sync();
  for (uint32_t sp = start; sp + 64 <= end; sp += 4096) {
    if (lane >= 1) {
      const uint32_t pos = sp + (uint32_t)lane;
      const uint32_t l3 = (0x2100u >> ((b.data[pos] >> 6) * 4u)) & 3u;        // {0,0,1,2}
      const uint32_t p3 = (0x2100u >> ((b.data[pos - 1] >> 6) * 4u)) & 3u;
      lds_atomic_add(&b.lds[p3 * 3u + l3], 1u);
    }
  }
  wave_sync();
  uint32_t monogram[3] = {0, 0, 0}, two_prefix[6] = {0, 0, 0, 0, 0, 0}, bigram[9];
  for (uint32_t i = 0; i < 9; ++i) bigram[i] = b.lds[i];
  for (uint32_t i = 0; i < 9; ++i) {
    monogram[i % 3] += bigram[i];
    two_prefix[i % 6] += bigram[i];
  }
  double e1 = estimate_entropy(monogram, 3, lut2);
  double e2 = estimate_entropy(two_prefix, 3, lut2) + estimate_entropy(two_prefix + 3, 3, lut2);
  double e3 = 0.0;
  for (uint32_t i = 0; i < 3; ++i) e3 += estimate_entropy(bigram + 3 * i, 3, lut2);
  const uint32_t total = monogram[0] + monogram[1] + monogram[2];
  const double inv = 1.0 / (double)total;
  e1 *= inv;
  e2 *= inv;
  e3 *= inv;
  if (J.quality < 7) e3 = e1 * 10;
  wave_sync();
  if (e1 - e2 < 0.2 && e1 - e3 < 0.2) {
    b.nc = 1;
  } else if (e2 - e3 < 0.02) {
    b.nc = 2;
    b.map_kind = 2;
  } else {
    b.nc = 3;
    b.map_kind = 1;
  }
}

// ---- symbol streams ------------------------------------------------------------
// lits[k] = literal | context << 8 in stream order; dsym[k] = distance symbol of
// the k-th command that carries one (metablock.c:741-769).  64 commands per
// step: wave scans give every command its first literal index and source
// position; the literals of the step are then copied FLAT, one literal per
// lane, each lane finding its command by a binary search over the 64 start
// indices kept in LDS (so one long insert run is spread over all lanes and
// every load of a step is independent).
// Commands [c0, c1) of the meta-block (c0 a multiple of 64), given where they begin: source position `pos`, first
// literal index `nlits`, first distance index `ndist`.  The whole meta-block in one call (k_build) or one part of it
// per wave (k_wide.h: the parts' starting values come from a scan over their totals).
// lds_tables: b.lds has room behind word 160 for the two context LUTs and the context map (k_build: yes) — a literal's
// context then costs one 4-byte load of the input (the literal and the two bytes before it) and three LDS reads instead
// of three dependent trips to memory; the literal loop is a third of k_build on text and most of it on noise.
DEV void build_streams_range(BuildCtx& b, uint32_t c0, uint32_t c1, uint32_t pos, uint32_t nlits, uint32_t ndist,
                             uint32_t* nlits_out, uint32_t* ndist_out, bool lds_tables = false) {
  const int lane = wave_lane();
  const uint8_t* clut = b.T->context_lut;
  const uint8_t* cmap = k_ctx_maps[b.map_kind];
  const bool use_ctx = b.nc > 1;
  uint32_t* s_start = b.lds;        // [65] first literal index of command i of the step
  uint32_t* s_src = b.lds + 65;     // [64] source position of that literal
  uint8_t* t_clut = (uint8_t*)(b.lds + 160);     // [512]
  uint8_t* t_cmap = t_clut + 512;                // [64]
  if (lds_tables && use_ctx) {
    for (uint32_t i = (uint32_t)lane; i < 128u; i += 64u) ((uint32_t*)t_clut)[i] = ((const uint32_t*)clut)[i];
    if (lane < 16) ((uint32_t*)t_cmap)[lane] = ((const uint32_t*)cmap)[lane];
    wave_sync();
  }
  for (uint32_t base = c0; base < c1; base += 64) {
    const uint32_t i = base + (uint32_t)lane;
    const bool valid = i < c1;
    uint32_t ins = 0, cpy = 0, prefix = 0, dprefix = 0;
    if (valid) {
      const Command c = b.cmds[i];
      ins = c.insert_len;
      cpy = c.copy_len & 0x1FFFFFFu;
      prefix = c.cmd_prefix;
      dprefix = c.dist_prefix;
    }
    const uint32_t ins_incl = wave_incl_scan(ins);
    const uint32_t adv_incl = wave_incl_scan(ins + cpy);
    const uint32_t my_lit = nlits + ins_incl - ins;
    const uint32_t my_pos = pos + adv_incl - (ins + cpy);
    const bool has_dist = valid && cpy != 0 && prefix >= 128;
    const uint64_t dm = wave_ballot(has_dist);
    if (has_dist) {
      const uint32_t k = ndist + (uint32_t)dev_popc64(dm & ((1ull << lane) - 1ull));
      b.dsym[k] = (uint16_t)(dprefix & 0x3FFu);
    }
    const uint32_t total = wave_bcast(ins_incl, 63);
    s_start[lane] = my_lit;
    s_src[lane] = my_pos;
    if (lane == 63) s_start[64] = nlits + total;
    wave_sync();
    for (uint32_t L = nlits + (uint32_t)lane; L < nlits + total; L += 64) {
      // largest c with s_start[c] <= L
      uint32_t lo = 0, hi = 63;
      while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (s_start[mid] <= L) lo = mid; else hi = mid - 1;
      }
      const uint32_t p = s_src[lo] + (L - s_start[lo]);
      uint32_t v;
      if (lds_tables && use_ctx && p >= 2u) {
        const uint32_t w3 = ld32(b.data + p - 2u);               // bytes p - 2, p - 1, p (and one more)
        v = (w3 >> 16) & 0xFFu;
        v |= (uint32_t)t_cmap[t_clut[(w3 >> 8) & 0xFFu] | t_clut[256u + (w3 & 0xFFu)]] << 8;
      } else {
        v = b.data[p];
        if (use_ctx) {
          const uint32_t p1 = p >= 1 ? b.data[p - 1] : 0u, p2 = p >= 2 ? b.data[p - 2] : 0u;
          v |= (uint32_t)cmap[clut[p1] | clut[256 + p2]] << 8;
        }
      }
      b.lits[L] = (uint16_t)v;
    }
    wave_sync();
    nlits += total;
    pos += wave_bcast(adv_incl, 63);
    ndist += (uint32_t)dev_popc64(dm);
  }
  wave_sync();
  *nlits_out = nlits;
  *ndist_out = ndist;
}
DEV void build_streams(BuildCtx& b, uint32_t start, uint32_t ncmds, uint32_t* nlits_out, uint32_t* ndist_out) {
  build_streams_range(b, 0u, ncmds, start, 0u, 0u, nlits_out, ndist_out, true);
}

// ---- greedy block splitter -------------------------------------------------------
// CAT 0: literals (nc contexts), 1: commands, 2: distances.
template <int CAT>
DEV void run_splitter(BuildCtx& b, uint32_t nsym) {
  constexpr uint32_t A = CAT == 0 ? 256u : CAT == 1 ? 704u : 64u;
  constexpr uint32_t MINB = CAT == 0 ? MB_LIT_MIN_BLOCK : CAT == 1 ? MB_CMD_MIN_BLOCK : MB_DIST_MIN_BLOCK;
  constexpr uint32_t ROW = LDS_ROW(A);
  const double threshold = CAT == 0 ? 400.0 : CAT == 1 ? 500.0 : 100.0;
  const int lane = wave_lane();
  const uint32_t nc = CAT == 0 ? b.nc : 1u;
  constexpr uint32_t EPR = (A * 3u <= BUILD_TERMS) ? 3u : 1u;   // evaluations per round
  const uint32_t max_types = MB_MAX_TYPES / nc;
  const double* lut2 = b.T->log2_lut;
  uint8_t* types = b.mb + b.L.types[CAT];
  uint32_t* lengths = (uint32_t*)(b.mb + b.L.lengths[CAT]);
  uint16_t* blkmap = (uint16_t*)(b.mb + b.L.blkmap[CAT]);
  uint32_t* G = (uint32_t*)(b.mb + b.L.histos[CAT]);
  uint32_t* cur = b.lds;
  // qualities 2 - 3 do not split (BrotliStoreMetaBlockTrivial / Fast, brotli_bit_stream.c:1196-1314):
  // everything lands in the one block the final decision opens
  const bool single = b.J->quality < 4;

  for (uint32_t k = (uint32_t)lane; k < nc * ROW; k += 64) cur[k] = 0;
  wave_sync();

  uint32_t num_blocks = 0, num_types = 0, block_size = 0, target = MINB, merge_last_count = 0;
  uint32_t last_t0 = 0, last_t1 = 0;
  uint32_t chunk_lo = 0, chunk_hi = 0;   // chunks of the running block

  // One decision of BlockSplitterFinishBlock / ContextBlockSplitterFinishBlock.
  auto finish = [&](bool is_final) {
    if (block_size < MINB) block_size = MINB;
    uint32_t blk_index;
    if (num_blocks == 0) {
      if (nc > 3) {
        if ((uint32_t)lane < nc) {
          const double e = bits_entropy2(cur + (uint32_t)lane * ROW, nullptr, A, lut2);
          b.lds_last[lane] = e;
          b.lds_last[nc + (uint32_t)lane] = e;
        }
        wave_sync();
      }
      for (uint32_t i0 = 0; nc <= 3 && i0 < nc; i0 += EPR) {
        const uint32_t ne = umin(EPR, nc - i0);
        const uint32_t* ap[3];
        const uint32_t* gp[3];
        for (uint32_t e = 0; e < 3; ++e) { ap[e] = cur + (i0 + (e < ne ? e : 0)) * ROW; gp[e] = nullptr; }
        bits_entropy_wave(ap, gp, ne, A, lut2, b.lds_terms, b.lds_sums, b.lds_ent);
        if ((uint32_t)lane < ne) {
          const double e = b.lds_ent[lane];
          b.lds_last[i0 + (uint32_t)lane] = e;
          b.lds_last[nc + i0 + (uint32_t)lane] = e;
        }
        wave_sync();
      }
      if (lane == 0) { lengths[0] = block_size; types[0] = 0; }
      for (uint32_t k = (uint32_t)lane; k < nc * A; k += 64) {
        const uint32_t i = k / A, s = k % A;
        G[k] = cur[i * ROW + s];
        cur[i * ROW + s] = 0;
      }
      blk_index = 0;
      num_blocks = 1;
      num_types = 1;
      block_size = 0;
    } else {
      // lane = 3 * ctx + which: 0 current block, 1 merged with the last type,
      // 2 merged with the second last type.
      // For every context: the current block alone, merged with the last type
      // and merged with the second last type (lds_ent3[3 * ctx + which]).
      if (nc > 3) {
        // 13 literal contexts: 39 independent evaluations, one lane each
        if ((uint32_t)lane < 3u * nc) {
          const uint32_t i = (uint32_t)lane / 3u, w = (uint32_t)lane % 3u;
          const uint32_t* g = w == 0 ? nullptr : G + ((w == 1 ? last_t0 : last_t1) * nc + i) * A;
          b.lds_ent3[lane] = bits_entropy2(cur + i * ROW, g, A, lut2);
        }
        wave_sync();
      } else if (A * 3u <= BUILD_TERMS) {
        for (uint32_t i = 0; i < nc; ++i) {
          const uint32_t* ap[3] = {cur + i * ROW, cur + i * ROW, cur + i * ROW};
          const uint32_t* gp[3] = {nullptr, G + (last_t0 * nc + i) * A, G + (last_t1 * nc + i) * A};
          bits_entropy_wave(ap, gp, 3, A, lut2, b.lds_terms, b.lds_sums, b.lds_ent);
          if (lane < 3) b.lds_ent3[3 * i + (uint32_t)lane] = b.lds_ent[lane];
          wave_sync();
        }
      } else {
        for (uint32_t w = 0; w < 3; ++w) {      // 704-symbol alphabet: one evaluation at a time
          const uint32_t* ap[3] = {cur, cur, cur};
          const uint32_t* gp[3] = {w == 0 ? nullptr : G + (w == 1 ? last_t0 : last_t1) * A, nullptr, nullptr};
          bits_entropy_wave(ap, gp, 1, A, lut2, b.lds_terms, b.lds_sums, b.lds_ent);
          if (lane == 0) b.lds_ent3[w] = b.lds_ent[0];
          wave_sync();
        }
      }
      double diff0 = 0.0, diff1 = 0.0;
      for (uint32_t i = 0; i < nc; ++i) {
        const double e = b.lds_ent3[3 * i];
        diff0 += b.lds_ent3[3 * i + 1] - e - b.lds_last[i];
        diff1 += b.lds_ent3[3 * i + 2] - e - b.lds_last[nc + i];
      }
      wave_sync();
#if defined(BROTLI_AMD_SIMT_SIM)
      if (lane == 0 && getenv("SIM_DEBUG2")) fprintf(stderr, "cat %d nb=%u bs=%u e=%f c0=%f c1=%f l0=%f l1=%f d0=%f d1=%f\n", CAT, num_blocks, block_size, b.lds_ent3[0], b.lds_ent3[1], b.lds_ent3[2], b.lds_last[0], b.lds_last[nc], diff0, diff1);
#endif
      if (num_types < max_types && diff0 > threshold && diff1 > threshold) {
        // New block type.
        if (lane == 0) { lengths[num_blocks] = block_size; types[num_blocks] = (uint8_t)num_types; }
        last_t1 = last_t0;
        last_t0 = num_types;
        if ((uint32_t)lane < nc) {
          b.lds_last[nc + (uint32_t)lane] = b.lds_last[lane];
          b.lds_last[lane] = b.lds_ent3[3 * lane];
        }
        uint32_t* dst = G + (size_t)num_types * nc * A;
        for (uint32_t k = (uint32_t)lane; k < nc * A; k += 64) {
          const uint32_t i = k / A, s = k % A;
          dst[k] = cur[i * ROW + s];
          cur[i * ROW + s] = 0;
        }
        blk_index = num_blocks;
        ++num_blocks;
        ++num_types;
        block_size = 0;
        merge_last_count = 0;
        target = MINB;
      } else if (diff1 < diff0 - 20.0) {
        // Back to the second last type.
        if (lane == 0) { lengths[num_blocks] = block_size; types[num_blocks] = types[num_blocks - 2]; }
        const uint32_t t = last_t0; last_t0 = last_t1; last_t1 = t;
        if ((uint32_t)lane < nc) {
          b.lds_last[nc + (uint32_t)lane] = b.lds_last[lane];
          b.lds_last[lane] = b.lds_ent3[3 * lane + 2];
        }
        uint32_t* dst = G + (size_t)last_t0 * nc * A;
        for (uint32_t k = (uint32_t)lane; k < nc * A; k += 64) {
          const uint32_t i = k / A, s = k % A;
          dst[k] += cur[i * ROW + s];
          cur[i * ROW + s] = 0;
        }
        blk_index = num_blocks;
        ++num_blocks;
        block_size = 0;
        merge_last_count = 0;
        target = MINB;
      } else {
        // Extend the last block.
        if (lane == 0) lengths[num_blocks - 1] += block_size;
        if ((uint32_t)lane < nc) {
          b.lds_last[lane] = b.lds_ent3[3 * lane + 1];
          if (num_types == 1) b.lds_last[nc + (uint32_t)lane] = b.lds_last[lane];
        }
        uint32_t* dst = G + (size_t)last_t0 * nc * A;
        for (uint32_t k = (uint32_t)lane; k < nc * A; k += 64) {
          const uint32_t i = k / A, s = k % A;
          dst[k] += cur[i * ROW + s];
          cur[i * ROW + s] = 0;
        }
        blk_index = num_blocks - 1;
        block_size = 0;
        if (++merge_last_count > 1) target += MINB;
      }
    }
    for (uint32_t c = chunk_lo + (uint32_t)lane; c < chunk_hi; c += 64) blkmap[c] = (uint16_t)blk_index;
    chunk_lo = chunk_hi;
    wave_sync();
    (void)is_final;
  };

  for (uint32_t s0 = 0; s0 < nsym; s0 += MINB) {
    const uint32_t n = umin(MINB, nsym - s0);
    for (uint32_t k = (uint32_t)lane; k < n; k += 64) {
      uint32_t sym, ctx = 0;
      if (CAT == 0) { const uint32_t v = b.lits[s0 + k]; sym = v & 0xFFu; ctx = v >> 8; }
      else if (CAT == 1) sym = b.cmds[s0 + k].cmd_prefix;
      else sym = b.dsym[s0 + k];
      lds_atomic_add(&cur[ctx * ROW + sym], 1u);
    }
    wave_sync();
    block_size += n;
    ++chunk_hi;
    if (!single && block_size == target) finish(false);
  }
  finish(true);

  if (lane == 0) {
    MbInfo* info = (MbInfo*)(b.mb + b.L.info);
    info->split[CAT].num_types = num_types;
    info->split[CAT].num_blocks = num_blocks;
    info->split[CAT].num_histograms = num_types * nc;
    info->split[CAT].nsym = nsym;
  }
  wave_sync();
}

// ---- histogram smoothing before the prefix codes are built ------------------------------
// What BrotliOptimizeHuffmanCountsForRle (entropy_encode.c:241-370) does to a histogram, said in
// terms of runs and segments (one lane per histogram; `sticky` is that lane's flag array):
//   * nothing for fewer than 16 used symbols; the tail of unused symbols never takes part;
//   * when some symbol is rarer than 4 and fewer than 6 symbols inside are unused, an unused
//     symbol between two used ones counts as seen once;
//   * nothing more for fewer than 28 used symbols;
//   * long runs of one value (>= 5 zeros, >= 7 equal non-zeros) are "sticky": they already
//     run-length code well and stay as they are, each of their elements a segment of its own;
//   * the rest is cut, left to right, into segments whose members stay within 1240 / 256 of the
//     segment's running mean (seeded by the mean of the first three values + 420 / 256, and
//     given 120 / 256 of slack when the fourth member joins); a segment of four or more — or of
//     three zeros — is flattened to its rounded mean (at least 1 unless all were 0).
// The reference's mix of 32-bit products and 64-bit sums is kept: the sums wrap the same way.
DEV void smooth_histogram_for_rle(uint32_t length, uint32_t* counts, uint8_t* sticky) {
  uint32_t used = 0, rarest = 1u << 30;
  for (uint32_t i = 0; i < length; ++i) {
    if (counts[i] != 0) { ++used; rarest = umin(rarest, counts[i]); }
  }
  if (used < 16u) return;
  while (counts[length - 1u] == 0) --length;                 // (used != 0: this stops)
  if (rarest < 4u && length - used < 6u) {
    // (a filled slot never makes its neighbour fillable: that one would have to be unused too)
    for (uint32_t i = 1; i + 1u < length; ++i) {
      if (counts[i] == 0 && counts[i - 1u] != 0 && counts[i + 1u] != 0) counts[i] = 1;
    }
  }
  if (used < 28u) return;
  for (uint32_t a = 0; a < length;) {
    uint32_t b = a + 1u;
    while (b < length && counts[b] == counts[a]) ++b;
    const uint8_t keep = (b - a >= (counts[a] == 0 ? 5u : 7u)) ? 1 : 0;
    for (uint32_t i = a; i < b; ++i) sticky[i] = keep;
    a = b;
  }
  for (uint32_t start = 0; start < length;) {
    uint64_t members = 0, sum = 0, mean256;
    if (start + 2u < length) mean256 = (uint64_t)(256u * (counts[start] + counts[start + 1u] + counts[start + 2u]) / 3u + 420u);
    else mean256 = (uint64_t)(256u * counts[start]);
    uint32_t end = start;
    do {
      ++members;
      sum += counts[end];
      if (members >= 4u) mean256 = (256u * sum + members / 2u) / members + (members == 4u ? 120u : 0u);
      ++end;
    } while (end < length && !sticky[end] && !sticky[end - 1u] &&
             (uint64_t)(256u * counts[end]) - mean256 + 1240u < 2480u);
    if (members >= 4u || (members == 3u && sum == 0)) {
      uint64_t flat = (sum + members / 2u) / members;
      if (flat == 0) flat = 1;
      if (sum == 0) flat = 0;
      for (uint32_t i = start; i < end; ++i) counts[i] = (uint32_t)flat;
    }
    start = end;
  }
}

// ---- the round --------------------------------------------------------------------
DEV void build_ctx_init(BuildCtx& b, const JobParams& J, const ShardDesc& D, const DeviceTables* T, const uint8_t* input,
                        uint8_t* ws, uint32_t* lds, double* lds_ent, double* lds_last, double* lds_terms) {
  b.J = &J;
  b.data = input + D.in_off;
  b.T = T;
  b.mb = ws + D.mb_off;
  mb_layout(umin(D.len, J.max_metablock_size), &b.L);
  b.cmds = (const Command*)(ws + D.cmds_off);
  b.lits = (uint16_t*)(ws + D.lits_off);
  b.dsym = (uint16_t*)(ws + D.dsym_off);
  b.lds = lds;
  b.lds_ent = lds_ent;               // [0..2] call results, [4..4+39) per-decision table
  b.lds_ent3 = lds_ent ? lds_ent + 4 : nullptr;        // (a caller that only makes the symbol streams has none of these)
  b.lds_last = lds_last;
  b.lds_terms = lds_terms;
  b.lds_sums = lds_terms ? (uint32_t*)(lds_terms + BUILD_TERMS) : nullptr;
  b.nc = 1;
  b.map_kind = 0;
}

// BrotliOptimizeHistograms: one lane per histogram, but on LDS copies (the
// smoothing is a serial scan with data-dependent rewrites: ~5 passes of
// dependent accesses per entry).  Batches of as many histograms as fit the
// block-splitter's LDS row buffer; the flag bytes live in the term buffer.
DEV void build_smooth_histograms(BuildCtx& b, const MbInfo* info) {
  const int lane = wave_lane();
  const uint32_t nh[3] = {info->split[0].num_histograms, info->split[1].num_histograms,
                          info->split[2].num_histograms};
  const uint32_t alpha[3] = {256u, 704u, 64u};
  uint8_t* flags = (uint8_t*)b.lds_terms;
  for (int c = 0; c < 3; ++c) {
    uint32_t* G = (uint32_t*)(b.mb + b.L.histos[c]);
    const uint32_t A = alpha[c];
    uint32_t per = umin(BUILD_LDS_WORDS / A, (BUILD_TERMS * 8u) / A);
    per = umin(per, 64u);
    for (uint32_t h0 = 0; h0 < nh[c]; h0 += per) {
      const uint32_t nb = umin(per, nh[c] - h0);
      for (uint32_t k = (uint32_t)lane; k < nb * A; k += 64) b.lds[k] = G[(size_t)h0 * A + k];
      wave_sync();
      if ((uint32_t)lane < nb) smooth_histogram_for_rle(A, b.lds + (uint32_t)lane * A, flags + (uint32_t)lane * A);
      wave_sync();
      for (uint32_t k = (uint32_t)lane; k < nb * A; k += 64) G[(size_t)h0 * A + k] = b.lds[k];
      wave_sync();
    }
  }
  wave_sync();
}

DEV void build_round(const JobParams& J, const ShardDesc& D, ShardState* S,
                     const DeviceTables* T, const uint8_t* input, uint8_t* ws,
                     uint32_t* lds, double* lds_ent, double* lds_last, double* lds_terms) {
  const int lane = wave_lane();
  if (!S->mb_valid || S->error) return;
  BuildCtx b;
  build_ctx_init(b, J, D, T, input, ws, lds, lds_ent, lds_last, lds_terms);

  const uint32_t start = S->mb_start, bytes = S->mb_bytes;
  const uint32_t ncmds = S->ncmds, nlits_state = S->nlits;
  MbInfo* info = (MbInfo*)(b.mb + b.L.info);

  uint64_t bpt = BP_NOW();
  if (!should_compress(b, start, bytes, nlits_state, ncmds)) {
    if (lane == 0) S->mb_raw = 1;
    wave_sync();
    return;
  }
  decide_contexts(b, start, bytes);
  uint32_t nlits = 0, ndist = 0;
  BP_ADD(S, 0, bpt);
  build_streams(b, start, ncmds, &nlits, &ndist);
  BP_ADD(S, 1, bpt);
  if (lane == 0) {
    info->num_contexts = b.nc;
    info->map_kind = b.map_kind;
    info->nlits = nlits;
    info->ndist = ndist;
    info->ncmds = ncmds;
  }
  run_splitter<0>(b, nlits);
  BP_ADD(S, 2, bpt);
  run_splitter<1>(b, ncmds);
  BP_ADD(S, 3, bpt);
  run_splitter<2>(b, ndist);
  BP_ADD(S, 4, bpt);

  if (J.quality < 4) { wave_sync(); BP_ADD(S, 5, bpt); return; }   // encode.c:587: from quality 4 on
  build_smooth_histograms(b, info);
  BP_ADD(S, 5, bpt);
}

#endif  // BROTLI_AMD_CSRC_K_BUILD_H_
