// brotli_amd/csrc/k_store.h — K6/K7/K9: prefix-code construction and bit-stream
// emission of one meta-block per shard by one wavefront.
//
// Semantics (bit-exact): BrotliStoreMetaBlock (c/enc/brotli_bit_stream.c:947-1114)
// with BuildAndStoreHuffmanTree (:349-397), BrotliStoreHuffmanTree (:283-345),
// BuildAndStoreBlockSplitCode (:760-791), EncodeContextMap (:574-734),
// StoreTrivialContextMap (:794-830), StoreSymbol/StoreBlockSwitch (:737-756,
// 879-918); BrotliCreateHuffmanTree / BrotliWriteHuffmanTree /
// BrotliConvertBitDepthsToSymbols (c/enc/entropy_encode.c:20-147, 160-239,
// 372-497); BrotliStoreUncompressedMetaBlock (:1321-1352); the size check and
// state update of WriteMetaBlockInternal / EncodeData (c/enc/encode.c:598-614,
// 1188-1216).
//
// Design: the reference writes one bit field after another.  Here
//  * every prefix code of the meta-block (literal / command / distance
//    histograms, block-type and block-length codes, context-map codes) is a
//    job the whole wave works through (k_prefix.h: LDS sort of the leaves,
//    two-queue merge, depths by pointer jumping, canonical codes and the
//    run-length coded description by ballots and scans) into a 512-byte buffer;
//  * the header is then assembled in order, splicing those buffers with
//    wave-wide shifted copies;
//  * the command stream is emitted 64 commands at a time: each lane sizes its
//    command (symbol, extra bits, literals, distance, block switches), a wave
//    scan turns sizes into bit offsets, and the lanes OR their bits into the
//    zeroed output with dword atomics.  Block boundaries are multiples of the
//    splitter's min_block_size, so "which block is symbol k in" is a table
//    lookup (blkmap) instead of the reference's running counters.
#ifndef BROTLI_AMD_CSRC_K_STORE_H_
#define BROTLI_AMD_CSRC_K_STORE_H_

#include "k_build.h"
#include "k_round.h"
#include "k_prefix.h"

// Optional phase timing (-DS_PROFILE): cycles per phase into ShardState::prof[].
#if defined(S_PROFILE) && !defined(BROTLI_AMD_SIMT_SIM)
#define SP_NOW() __builtin_amdgcn_s_memtime()
#define SP_ADD(S, i, t0) do { const uint64_t sp_n = SP_NOW(); if (wave_lane() == 0) (S)->prof[i] += sp_n - (t0); (t0) = sp_n; } while (0)
#else
#define SP_NOW() 0ull
#define SP_ADD(S, i, t0) do { (void)(t0); } while (0)
#endif

// ---- output bit sink ---------------------------------------------------------
struct BitSink {
  uint32_t* base;     // 4-byte aligned
  uint64_t bitpos;    // next bit, relative to base
};

// ORs the low `n` bits of `v` (n <= 64) at bit position `pos`.  One lane.
DEV void or_bits(uint32_t* base, uint64_t pos, uint32_t n, uint64_t v) {
  if (n == 0) return;
  if (n < 64) v &= (1ull << n) - 1ull;
  const uint64_t w = pos >> 5;
  const uint32_t sh = (uint32_t)pos & 31u;
  const uint64_t lo = v << sh;
  const uint32_t w0 = (uint32_t)lo, w1 = (uint32_t)(lo >> 32);
  const uint32_t w2 = sh ? (uint32_t)(v >> (64u - sh)) : 0u;
  if (w0) glb_atomic_or(base + w, w0);
  if (w1) glb_atomic_or(base + w + 1, w1);
  if (w2) glb_atomic_or(base + w + 2, w2);
}
// The same into an LDS window whose dword 0 starts at bit 0 of `w`.
DEV void lds_or_bits(uint32_t* w, uint32_t pos, uint32_t n, uint64_t v) {
  if (n == 0) return;
  if (n < 64) v &= (1ull << n) - 1ull;
  const uint32_t d = pos >> 5, sh = pos & 31u;
  const uint64_t lo = v << sh;
  const uint32_t w0 = (uint32_t)lo, w1 = (uint32_t)(lo >> 32);
  const uint32_t w2 = sh ? (uint32_t)(v >> (64u - sh)) : 0u;
  if (w0) lds_atomic_or(w + d, w0);
  if (w1) lds_atomic_or(w + d + 1, w1);
  if (w2) lds_atomic_or(w + d + 2, w2);
}
// Uniform call: lane 0 writes, every lane advances.
DEV void sink_put(BitSink& s, uint32_t n, uint64_t v) {
  if (wave_lane() == 0) or_bits(s.base, s.bitpos, n, v);
  s.bitpos += n;
}
// Splices `nbits` bits that start at byte `src` (whole wave).
DEV void sink_splice(BitSink& s, const uint8_t* src, uint32_t nbits) {
  const int lane = wave_lane();
  const uint32_t nwords = (nbits + 31u) >> 5;
  for (uint32_t j = (uint32_t)lane; j < nwords; j += 64) {
    const uint32_t nb = umin(32u, nbits - 32u * j);
    or_bits(s.base, s.bitpos + 32ull * j, nb, ld32(src + 4u * j));
  }
  s.bitpos += nbits;
}
DEV void sink_varlen_uint8(BitSink& s, uint32_t n) {   // StoreVarLenUint8
  if (n == 0) {
    sink_put(s, 1, 0);
  } else {
    const uint32_t nbits = log2floor(n);
    sink_put(s, 1, 1);
    sink_put(s, 3, nbits);
    sink_put(s, nbits, n - (1u << nbits));
  }
}

#define STORE_WIN_DW 2048u   // LDS output window of the command stream (64 Kbit)
// LDS of the kernel: the window + 132 words of step bookkeeping — or, before the command stream
// starts, the work area of the prefix-code builder (k_prefix.h)
#define STORE_LDS_WORDS (PFX_LDS_WORDS > 132u + STORE_WIN_DW + 4u ? PFX_LDS_WORDS : 132u + STORE_WIN_DW + 4u)

// ---- block-split bookkeeping ------------------------------------------------------
// brotli_bit_stream.c:34-46 and the block-length prefix table of
// c/common/constants.h:195-196.
static __device__ const uint16_t k_blocklen_offset[26] = {
    1, 5, 9, 13, 17, 25, 33, 41, 49, 65, 81, 97, 113, 145, 177, 209, 241, 305, 369, 497,
    753, 1265, 2289, 4337, 8433, 16625};
static __device__ const uint8_t k_blocklen_nbits[26] = {
    2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 6, 6, 7, 8, 9, 10, 11, 12, 13, 24};
DEV uint32_t block_length_prefix_code(uint32_t len) {
  uint32_t code = (len >= 177) ? (len >= 753 ? 20 : 14) : (len >= 41 ? 7 : 0);
  while (code < 25 && len >= k_blocklen_offset[code + 1]) ++code;
  return code;
}
// NextBlockTypeCode for block b (b >= 0), brotli_bit_stream.c:709-716 with the
// initial calculator state {last_type = 1, second_last_type = 0}.
DEV uint32_t block_type_code(const uint8_t* types, uint32_t b) {
  const uint32_t t = types[b];
  const uint32_t last = b >= 1 ? types[b - 1] : 1u;
  const uint32_t second = b >= 2 ? types[b - 2] : (b == 1 ? 1u : 0u);
  return (t == last + 1) ? 1u : (t == second) ? 0u : t + 2u;
}

// command.h:31-88 tables.
static __device__ const uint32_t k_ins_base[24] = {0, 1, 2, 3, 4, 5, 6, 8, 10, 14, 18, 26,
    34, 50, 66, 98, 130, 194, 322, 578, 1090, 2114, 6210, 22594};
static __device__ const uint8_t k_ins_extra[24] = {0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4,
    5, 5, 6, 7, 8, 9, 10, 12, 14, 24};
static __device__ const uint32_t k_copy_base[24] = {2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 18,
    22, 30, 38, 54, 70, 102, 134, 198, 326, 582, 1094, 2118};
static __device__ const uint8_t k_copy_extra[24] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3,
    4, 4, 5, 5, 6, 7, 8, 9, 10, 24};

// Element i (0 .. 2, not a compile-time constant: a lane's own category, a job number) of a three-element member of
// StoreCtx / StoreMeta, chosen between VALUES: an index through the member's address keeps the whole struct — and
// through StoreCtx::J the kernel's arguments — in scratch memory (k_store had .private_segment_fixed_size 1160 until
// round 6; the same story as q_dc_entry, k_parse4.h).
template <class T>
DEV T sel3(const T (&a)[3], uint32_t i) {
  const T v0 = dev_opaque(a[0]), v1 = dev_opaque(a[1]), v2 = dev_opaque(a[2]);
  T r = v2;
  r = i == 1u ? v1 : r;
  r = i == 0u ? v0 : r;
  return r;
}

struct StoreCtx {
  const JobParams* J;
  const uint8_t* data;
  uint8_t* mb;
  MbLayout L;
  const MbInfo* info;
  SmallCodes* small;
  const Command* cmds;
  const uint16_t* lits;
  const uint16_t* dsym;
  uint64_t* sw;          // per category, per block: switch bits (low 56) | nbits << 56
  uint32_t* lsum;        // [nlits + 1] bits of the literals before literal k
  uint32_t* lcode;       // [nlits] code | length << 16 of literal k
  uint32_t sw_off[3];
  uint32_t nc;
  // per category views
  const uint8_t* types[3];
  const uint32_t* lengths[3];
  const uint16_t* blkmap[3];
  const uint8_t* depths[3];
  const uint16_t* bits[3];
};

// Code of stream symbol k of category CAT and, when k opens a new block, the
// block switch that precedes it (StoreSymbol / StoreSymbolWithContext).
struct SymBits {
  uint64_t sw;      // block switch bits (<= 54)
  uint32_t nsw;
  uint32_t code;    // prefix code of the symbol (<= 15 bits)
  uint32_t ncode;
};
template <int CAT>
DEV SymBits symbol_bits(const StoreCtx& s, uint32_t k, uint32_t sym, uint32_t ctx) {
  constexpr uint32_t A = CAT == 0 ? 256u : CAT == 1 ? 704u : 64u;
  constexpr uint32_t MINB = CAT == 0 ? MB_LIT_MIN_BLOCK : CAT == 1 ? MB_CMD_MIN_BLOCK : MB_DIST_MIN_BLOCK;
  const uint32_t chunk = k / MINB;
  const uint32_t blk = s.blkmap[CAT][chunk];
  SymBits r;
  r.sw = 0;
  r.nsw = 0;
  if ((k % MINB) == 0 && chunk > 0 && s.blkmap[CAT][chunk - 1] != blk) {
    const uint64_t e = s.sw[s.sw_off[CAT] + blk];
    r.nsw = (uint32_t)(e >> 56);
    r.sw = e & ((1ull << 56) - 1ull);
  }
  const uint32_t type = s.types[CAT][blk];
  const uint32_t h = CAT == 0 ? type * s.nc + ctx : type;
  r.ncode = s.depths[CAT][h * A + sym];
  r.code = s.bits[CAT][h * A + sym];
  return r;
}
DEV void or_sym(uint32_t* base, uint64_t pos, const SymBits& b) {
  or_bits(base, pos, b.nsw, b.sw);
  or_bits(base, pos + b.nsw, b.ncode, b.code);
}

// ---- the round -----------------------------------------------------------------------
DEV void zero_output(uint8_t* out, uint64_t from, uint64_t bytes) {
  const int lane = wave_lane();
  // bytes up to the next dword boundary, then whole dwords
  const uint64_t head = (4u - (from & 3u)) & 3u;
  if ((uint64_t)lane < head) out[from + (uint64_t)lane] = 0;
  uint32_t* p = (uint32_t*)(out + from + head);
  const uint64_t nw = (bytes - head + 3) >> 2;
  for (uint64_t i = (uint64_t)lane; i < nw; i += 64) p[i] = 0;
  wave_mem_barrier();   // zeros are in the L2 before any atomic OR is issued
}

// ---- the literal context map (EncodeContextMap, brotli_bit_stream.c:574-734), the whole wave on it ----
// context_map[t * 64 + j] = t * nc + static_map[j] (metablock.c:677-697): row t is the static map shifted
// by t * nc.  Two rows share no value, so the map is a sequence of value runs none of which crosses a
// row, and the run heads sit in the same columns of every row (one ballot, taken once).  Under
// move-to-front (:592-618) a run of length L becomes the list index of its value followed by L - 1 zeros,
// and the zero coder (:626-673) turns those into run-length codes.  Hence
//   pass A, per row: only the heads consult the list — 256 entries in four registers x 64 lanes, entry e
//     in register e / 64 of lane e % 64; "where is the value" is one ballot per register, "shift the
//     entries before it" is every lane taking its left neighbour's entry;
//   pass B, per row: every head sizes its symbols (the index unless it is zero, then the codes of its
//     zero run), a wave scan places them; the symbols overwrite the array they are read from — a row
//     never yields more symbols than it has elements, and it is in registers before anything is written.
// cmap_rle[i] = symbol | extra bits << 9 as the header writer expects; histo[] = the symbol counts.
DEV void store_literal_context_map(uint32_t ntypes, uint32_t nc, const uint8_t* static_map, uint32_t* cmap_rle,
                                   uint32_t* histo, uint32_t* lds_hist, uint32_t& nrle_out, uint32_t& max_prefix_out) {
  const uint32_t lane = (uint32_t)wave_lane();
  uint32_t m[4];
#pragma unroll
  for (uint32_t r = 0; r < 4; ++r) m[r] = r * 64u + lane;
  const uint32_t sm = static_map[lane];
  const uint32_t sm_left = wave_shfl(sm, (int)((lane - 1u) & 63u));
  const bool head = lane == 0 || sm != sm_left;
  const uint64_t heads = wave_ballot(head);
  const uint64_t above = (heads >> lane) >> 1;
  const uint32_t run = above ? (uint32_t)dev_ctz64(above) + 1u : 64u - lane;   // (meaningful in head lanes)
  uint32_t max_reps = 0;
  for (uint32_t t = 0; t < ntypes; ++t) {
    uint32_t my_index = 0;
    for (uint64_t hm = heads; hm != 0; hm &= hm - 1) {
      const int j = dev_ctz64(hm);
      const uint32_t value = t * nc + wave_bcast(sm, j);
      uint32_t index = 0;
#pragma unroll
      for (uint32_t r = 0; r < 4; ++r) {
        const uint64_t at = wave_ballot(m[r] == value);
        if (at != 0) index = r * 64u + (uint32_t)dev_ctz64(at);
      }
      if ((int)lane == j) my_index = index;
      // entries 1 .. index take their predecessor, the value goes to the front (register 3 first: a
      // register's lane 0 needs the old lane 63 of the register below it)
#pragma unroll
      for (int r = 3; r >= 0; --r) {
        if ((uint32_t)r * 64u > index) continue;
        const uint32_t up = wave_shfl(m[r], (int)((lane - 1u) & 63u));
        const uint32_t carry = r > 0 ? wave_bcast(m[r > 0 ? r - 1 : 0], 63) : value;
        if ((uint32_t)r * 64u + lane <= index) m[r] = lane == 0 ? carry : up;
      }
    }
    // the very first element can sit at index 0 already (the list starts as the identity): it is a zero itself
    const uint32_t zeros = head ? run - 1u + ((t == 0 && lane == 0 && my_index == 0) ? 1u : 0u) : 0u;
    max_reps = umax(max_reps, wave_max_u32(zeros));
    cmap_rle[t * 64u + lane] = head ? (my_index | (zeros << 16)) : 0xFFFFFFFFu;
  }
  const uint32_t max_prefix = max_reps != 0 ? umin(log2floor(max_reps), 6u) : 0u;
  for (uint32_t i = lane; i < MB_MAX_CMAP_SYMS; i += 64u) lds_hist[i] = 0;
  wave_sync();
  uint32_t base = 0;
  for (uint32_t t = 0; t < ntypes; ++t) {
    const uint32_t w = cmap_rle[t * 64u + lane];
    const bool is_head = w != 0xFFFFFFFFu;
    const uint32_t index = w & 0xFFFFu, zeros = is_head ? w >> 16 : 0u;
    uint32_t n = (is_head && index != 0) ? 1u : 0u;
    for (uint32_t reps = zeros; reps != 0;) {
      ++n;
      if (reps < (2u << max_prefix)) break;
      reps -= (2u << max_prefix) - 1u;
    }
    const uint32_t incl = wave_incl_scan(n);
    wave_sync();
    uint32_t at = base + incl - n;
    if (is_head && index != 0) {
      cmap_rle[at++] = index + max_prefix;
      lds_atomic_add(&lds_hist[index + max_prefix], 1u);
    }
    for (uint32_t reps = zeros; reps != 0;) {
      uint32_t code;
      if (reps < (2u << max_prefix)) {
        const uint32_t p = log2floor(reps);
        code = p + ((reps - (1u << p)) << 9);
        reps = 0;
      } else {
        code = max_prefix + (((1u << max_prefix) - 1u) << 9);
        reps -= (2u << max_prefix) - 1u;
      }
      cmap_rle[at++] = code;
      lds_atomic_add(&lds_hist[code & 511u], 1u);
    }
    base += wave_bcast(incl, 63);
    wave_sync();
  }
  for (uint32_t i = lane; i < MB_MAX_CMAP_SYMS; i += 64u) histo[i] = lds_hist[i];
  wave_sync();
  nrle_out = base;
  max_prefix_out = max_prefix;
}

// ---- the pieces of a meta-block's store ------------------------------------------------------------
// store_round (one wave does everything, below) and the kernels of k_wide.h (a long meta-block spread over many
// waves) are built from the same pieces, so the bits are the same whichever way a meta-block is written.
struct StoreMeta {          // wave-uniform facts about the meta-block
  uint32_t ntypes[3], nblocks[3], nhist[3];
  uint32_t ncmds, njobs;
};
DEV void store_ctx_init(StoreCtx& s, StoreMeta& M, const JobParams& J, const ShardDesc& D, const uint8_t* input, uint8_t* ws) {
  s.J = &J;
  s.data = input + D.in_off;
  s.mb = ws + D.mb_off;
  mb_layout(umin(D.len, J.max_metablock_size), &s.L);
  s.info = (const MbInfo*)(s.mb + s.L.info);
  s.small = (SmallCodes*)(s.mb + s.L.small);
  s.cmds = (const Command*)(ws + D.cmds_off);
  s.lits = (const uint16_t*)(ws + D.lits_off);
  s.dsym = (const uint16_t*)(ws + D.dsym_off);
  // scratch: switch codes (<= mb / 256 + 64 blocks in total), then two words per literal
  const uint32_t mb_cap = umin(D.len, J.max_metablock_size);
  s.sw = (uint64_t*)(ws + D.scratch_off);
  s.lsum = (uint32_t*)(ws + D.scratch_off + ((uint64_t)mb_cap / 256u + 64u) * 8u);
  s.lcode = s.lsum + (mb_cap + 16u);
  s.nc = s.info->num_contexts;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    s.types[c] = s.mb + s.L.types[c];
    s.lengths[c] = (const uint32_t*)(s.mb + s.L.lengths[c]);
    s.blkmap[c] = (const uint16_t*)(s.mb + s.L.blkmap[c]);
    s.depths[c] = s.mb + s.L.depths[c];
    s.bits[c] = (const uint16_t*)(s.mb + s.L.bits[c]);
    M.ntypes[c] = s.info->split[c].num_types;
    M.nblocks[c] = s.info->split[c].num_blocks;
    M.nhist[c] = s.info->split[c].num_histograms;
  }
  s.sw_off[0] = 0;
  s.sw_off[1] = M.nblocks[0];
  s.sw_off[2] = M.nblocks[0] + M.nblocks[1];
  M.ncmds = s.info->ncmds;
  M.njobs = 8 + M.nhist[0] + M.nhist[1] + M.nhist[2];
}

// phase 0: histograms of the small codes (a lane each), the literal context map (the whole wave)
DEV void store_small_histos(StoreCtx& s, const StoreMeta& M, uint32_t* lds_store, uint32_t& cmap_nrle, uint32_t& cmap_max_prefix) {
  const int lane = wave_lane();
  SmallCodes* sc = s.small;
  cmap_nrle = 0; cmap_max_prefix = 0;
  if (lane < 3) {
    const int c = lane;
    const uint32_t ntypes_c = sel3(M.ntypes, (uint32_t)c), nblocks_c = sel3(M.nblocks, (uint32_t)c);
    const uint8_t* types_c = sel3(s.types, (uint32_t)c);
    const uint32_t* lengths_c = sel3(s.lengths, (uint32_t)c);
    for (uint32_t i = 0; i < ntypes_c + 2; ++i) sc->type_histo[c][i] = 0;
    for (uint32_t i = 0; i < 26; ++i) sc->len_histo[c][i] = 0;
    for (uint32_t b = 0; b < nblocks_c; ++b) {
      if (b != 0) ++sc->type_histo[c][block_type_code(types_c, b)];
      ++sc->len_histo[c][block_length_prefix_code(lengths_c[b])];
    }
  } else if (lane == 3 || lane == 4) {
    // Trivial context maps (:794-830): literal (only when nc == 1), distance.
    const int m = lane - 3;
    const uint32_t num_types = m == 0 ? M.nhist[0] : M.nhist[2];
    const uint32_t context_bits = m == 0 ? 6u : 2u;
    if ((m == 1 || s.nc == 1) && num_types > 1) {
      const uint32_t repeat_code = context_bits - 1u;
      const uint32_t alphabet_size = num_types + repeat_code;
      for (uint32_t i = 0; i < alphabet_size; ++i) sc->cmap_histo[m][i] = 0;
      sc->cmap_histo[m][repeat_code] = num_types;
      sc->cmap_histo[m][0] = 1;
      for (uint32_t i = context_bits; i < alphabet_size; ++i) sc->cmap_histo[m][i] = 1;
    }
  }
  wave_sync();
  if (s.nc > 1) store_literal_context_map(M.ntypes[0], s.nc, k_ctx_maps[s.info->map_kind], (uint32_t*)(s.mb + s.L.cmap_rle),
                                          sc->cmap_histo[0], lds_store, cmap_nrle, cmap_max_prefix);
  wave_sync();
}

// phase 1, one job: a prefix code, the whole wave on it.  job ids: 0-2 block types, 3-5 block lengths, 6 literal
// context map, 7 distance context map, then literal / command / distance histograms.
DEV void store_code_job(StoreCtx& s, const StoreMeta& M, uint32_t j, uint32_t cmap_max_prefix, uint32_t* lds_store) {
  const int lane = wave_lane();
  const JobParams& J = *s.J;
  const uint32_t alpha[3] = {256u, 704u, 64u};
  SmallCodes* sc = s.small;
  uint8_t* tree_bufs = s.mb + s.L.tree_bufs;
  uint32_t* job_nbits = (uint32_t*)(s.mb + s.L.jobs);
  const uint32_t lit_cmap_alpha = s.nc > 1 ? M.nhist[0] + cmap_max_prefix : M.nhist[0] + 5u;
  const uint32_t* histo = nullptr;
  uint8_t* depth = nullptr;
  uint16_t* bits = nullptr;
  uint32_t length = 0;
  bool skip = false;
  if (j < 3) {
    histo = sc->type_histo[j]; depth = sc->type_depth[j]; bits = sc->type_bits[j];
    length = sel3(M.ntypes, j) + 2; skip = sel3(M.ntypes, j) <= 1;
  } else if (j < 6) {
    const uint32_t c = j - 3;
    histo = sc->len_histo[c]; depth = sc->len_depth[c]; bits = sc->len_bits[c];
    length = 26; skip = sel3(M.ntypes, c) <= 1;
  } else if (j == 6) {
    histo = sc->cmap_histo[0]; depth = sc->cmap_depth[0]; bits = sc->cmap_bits[0];
    length = lit_cmap_alpha; skip = M.nhist[0] <= 1;
  } else if (j == 7) {
    histo = sc->cmap_histo[1]; depth = sc->cmap_depth[1]; bits = sc->cmap_bits[1];
    length = M.nhist[2] + 1u; skip = M.nhist[2] <= 1;
  } else {
    uint32_t h = j - 8;
    int c = 0;
    if (h >= M.nhist[0]) { h -= M.nhist[0]; c = 1; if (h >= M.nhist[1]) { h -= M.nhist[1]; c = 2; } }
    const uint32_t alpha_c = c == 0 ? 256u : c == 1 ? 704u : 64u;
    histo = (const uint32_t*)(s.mb + sel3(s.L.histos, (uint32_t)c)) + (size_t)h * alpha_c;
    depth = (uint8_t*)(s.mb + sel3(s.L.depths, (uint32_t)c)) + (size_t)h * alpha_c;
    bits = (uint16_t*)(s.mb + sel3(s.L.bits, (uint32_t)c)) + (size_t)h * alpha_c;
    length = alpha_c;
  }
  uint32_t nb = 0;
  if (!skip && J.quality == 2 && j >= 8) {
    // BrotliStoreMetaBlockFast (brotli_bit_stream.c:1242-1314): count-only trees; up to 128
    // commands the command and distance codes are the static ones (entropy_encode_static.h:
    // 448 command symbols of 9 bits + 256 of 11, 64 distance symbols of 6 bits, canonical,
    // bits reversed; their serialised forms are the constants of :524-541)
    uint8_t* buf = tree_bufs + (size_t)j * MB_TREE_BUF_BYTES;
    if (j == 8 || M.ncmds > 128u) {
      nb = pfx_build_and_store<true>(histo, length, length, lds_store, depth, bits, buf);
    } else if (j == 9) {
      for (uint32_t i = (uint32_t)lane; i < 704u; i += 64) {
        const uint32_t code = i < 448u ? i : 1792u + (i - 448u), nbits = i < 448u ? 9u : 11u;
        depth[i] = (uint8_t)nbits;
        bits[i] = (uint16_t)(dev_bitrev32(code) >> (32u - nbits));
      }
      if (lane == 0) { st32(buf, 0x16307003u); st32(buf + 4, 0x00926244u); }
      nb = 59;
    } else {
      if (lane < 64) { depth[lane] = 6; bits[lane] = (uint16_t)(dev_bitrev32((uint32_t)lane) >> 26); }
      if (lane == 0) st32(buf, 0x0369DC03u);
      nb = 28;
    }
    wave_sync();
  } else if (!skip) {
    nb = pfx_build_and_store(histo, length, length, lds_store, depth, bits,
                             tree_bufs + (size_t)j * MB_TREE_BUF_BYTES);
  }
  if (lane == 0) job_nbits[j] = nb;
}

// Block switch codes for every block b >= 1 (StoreBlockSwitch :737-756).
DEV void store_switch_codes(StoreCtx& s, const StoreMeta& M) {
  const int lane = wave_lane();
  SmallCodes* sc = s.small;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    for (uint32_t b = (uint32_t)lane; b < M.nblocks[c]; b += 64) {
      uint64_t v = 0;
      uint32_t n = 0;
      if (M.ntypes[c] > 1) {
        const uint32_t lencode = block_length_prefix_code(s.lengths[c][b]);
        if (b != 0) {
          const uint32_t tc = block_type_code(s.types[c], b);
          v = sc->type_bits[c][tc];
          n = sc->type_depth[c][tc];
        }
        v |= (uint64_t)sc->len_bits[c][lencode] << n;
        n += sc->len_depth[c][lencode];
        v |= (uint64_t)(s.lengths[c][b] - k_blocklen_offset[lencode]) << n;
        n += k_blocklen_nbits[lencode];
      }
      s.sw[s.sw_off[c] + b] = v | ((uint64_t)n << 56);
    }
  }
  wave_sync();
}

// phase 2: the meta-block header, in order, behind the bits carried over from the meta-block before.
DEV void store_header(StoreCtx& s, const StoreMeta& M, BitSink& sink, uint32_t last_bytes, uint32_t last_bytes_bits,
                      uint32_t bytes, bool is_last, uint32_t cmap_nrle, uint32_t cmap_max_prefix) {
  const JobParams& J = *s.J;
  SmallCodes* sc = s.small;
  uint8_t* tree_bufs = s.mb + s.L.tree_bufs;
  const uint32_t* job_nbits = (const uint32_t*)(s.mb + s.L.jobs);
  const uint32_t* cmap_rle = (const uint32_t*)(s.mb + s.L.cmap_rle);
  sink_put(sink, last_bytes_bits, last_bytes);
  {
    // StoreCompressedMetaBlockHeader :120-143
    const uint32_t lg = (bytes == 1) ? 1u : log2floor(bytes - 1u) + 1u;
    const uint32_t mnibbles = (lg < 16u ? 16u : (lg + 3u)) / 4u;
    sink_put(sink, 1, is_last ? 1 : 0);
    if (is_last) sink_put(sink, 1, 0);
    sink_put(sink, 2, mnibbles - 4u);
    sink_put(sink, mnibbles * 4u, bytes - 1u);
    if (!is_last) sink_put(sink, 1, 0);
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    sink_varlen_uint8(sink, M.ntypes[c] - 1u);
    if (M.ntypes[c] > 1) {
      sink_splice(sink, tree_bufs + (size_t)c * MB_TREE_BUF_BYTES, job_nbits[c]);
      sink_splice(sink, tree_bufs + (size_t)(3 + c) * MB_TREE_BUF_BYTES, job_nbits[3 + c]);
      const uint64_t e = s.sw[s.sw_off[c]];
      sink_put(sink, (uint32_t)(e >> 56), e & ((1ull << 56) - 1ull));
    }
  }
  sink_put(sink, 2, 0);   // NPOSTFIX
  sink_put(sink, 4, 0);   // NDIRECT >> NPOSTFIX
  // CONTEXT_UTF8 (encode.c:486-496); the writers of qualities 2 - 3 leave the field zero ("13 zero bits")
  for (uint32_t i = 0; i < M.ntypes[0]; ++i) sink_put(sink, 2, J.quality < 4 ? 0 : 2);
  if (s.nc == 1) {
    // StoreTrivialContextMap(num literal histograms, 6 context bits)
    sink_varlen_uint8(sink, M.nhist[0] - 1u);
    if (M.nhist[0] > 1) {
      sink_put(sink, 1, 1);
      sink_put(sink, 4, 4);
      sink_splice(sink, tree_bufs + 6u * MB_TREE_BUF_BYTES, job_nbits[6]);
      for (uint32_t i = 0; i < M.nhist[0]; ++i) {
        const uint32_t code = i == 0 ? 0u : i + 5u;
        sink_put(sink, sc->cmap_depth[0][code], sc->cmap_bits[0][code]);
        sink_put(sink, sc->cmap_depth[0][5], sc->cmap_bits[0][5]);
        sink_put(sink, 5, 31);
      }
      sink_put(sink, 1, 1);
    }
  } else {
    sink_varlen_uint8(sink, M.nhist[0] - 1u);
    const bool use_rle = cmap_max_prefix > 0;
    sink_put(sink, 1, use_rle ? 1 : 0);
    if (use_rle) sink_put(sink, 4, cmap_max_prefix - 1u);
    sink_splice(sink, tree_bufs + 6u * MB_TREE_BUF_BYTES, job_nbits[6]);
    for (uint32_t i = 0; i < cmap_nrle; ++i) {
      const uint32_t sym = cmap_rle[i] & 511u, extra = cmap_rle[i] >> 9;
      sink_put(sink, sc->cmap_depth[0][sym], sc->cmap_bits[0][sym]);
      if (sym > 0 && sym <= cmap_max_prefix) sink_put(sink, sym, extra);
    }
    sink_put(sink, 1, 1);
  }
  {
    // StoreTrivialContextMap(num distance histograms, 2 context bits)
    sink_varlen_uint8(sink, M.nhist[2] - 1u);
    if (M.nhist[2] > 1) {
      sink_put(sink, 1, 1);
      sink_put(sink, 4, 0);
      sink_splice(sink, tree_bufs + 7u * MB_TREE_BUF_BYTES, job_nbits[7]);
      for (uint32_t i = 0; i < M.nhist[2]; ++i) {
        const uint32_t code = i == 0 ? 0u : i + 1u;
        sink_put(sink, sc->cmap_depth[1][code], sc->cmap_bits[1][code]);
        sink_put(sink, sc->cmap_depth[1][1], sc->cmap_bits[1][1]);
        sink_put(sink, 1, 1);
      }
      sink_put(sink, 1, 1);
    }
  }
  for (uint32_t j = 8; j < M.njobs; ++j)
    sink_splice(sink, tree_bufs + (size_t)j * MB_TREE_BUF_BYTES, job_nbits[j]);
}

// phase 3 (a): literals [k0, k1) of the meta-block, flat: code | length << 16 of every literal (lcode) and the
// running sum of their bits, block switches included (lsum[k] = `carry` + bits of literals [k0, k)).  Returns
// `carry` + the bits of the whole range.
DEV uint32_t store_literal_codes(const StoreCtx& s, uint32_t k0, uint32_t k1, uint32_t carry) {
  const int lane = wave_lane();
  for (uint32_t kk = k0; kk < k1; kk += 64) {
    const uint32_t k = kk + (uint32_t)lane;
    uint32_t nb = 0;
    if (k < k1) {
      const uint32_t v = s.lits[k];
      const SymBits lb = symbol_bits<0>(s, k, v & 0xFFu, v >> 8);
      nb = lb.nsw + lb.ncode;
      s.lcode[k] = lb.code | (lb.ncode << 16);
    }
    const uint32_t incl = wave_incl_scan(nb);
    if (k < k1) s.lsum[k] = carry + incl - nb;
    carry += wave_bcast(incl, 63);
  }
  return carry;
}

// Bits of literals [0, k): the array itself when one wave wrote it (lsum[nlits] = the total), or, for a meta-block
// written in parts (k_wide.h), the sum inside the literal's part + the part's offset.
struct LitSums {
  const uint32_t* lsum;
  const uint32_t* part_off;      // nullptr: lsum[] holds absolute sums
  uint32_t nlits, total;
};
template <bool PARTS>
DEV uint32_t lit_sum(const LitSums& a, uint32_t k) {
  if (!PARTS) return a.lsum[k];
  return k >= a.nlits ? a.total : a.lsum[k] + a.part_off[k / WIDE_LIT_PART];
}

// phase 3 (b): commands [c0, c1) of the meta-block, 64 per step: command / distance codes per lane, a wave scan
// for their offsets, then the literals of the step written flat (one literal per lane; its command is found by a
// binary search over the step's first-literal indices kept in LDS).  The bits of a step are OR-ed into an LDS
// window (ds_or) and leave as whole dwords with plain coalesced stores; only a step whose span exceeds the window
// (a very long literal run) goes to HBM with dword atomics.
// The range starts `cbits` bits of command / distance codes behind `bit_cmds`, at literal `lit_base` and distance
// `dist_base`.  PARTS: neighbouring ranges are written by other waves at the same time — the first and the last
// dword of the range are shared with them and go out as atomic ORs (the output is zero where nothing was written).
// Returns cbits behind the range.
template <bool PARTS>
DEV uint32_t store_commands(const StoreCtx& s, uint32_t* sink_base, uint64_t bit_cmds, uint32_t c0, uint32_t c1,
                            uint32_t cbits, uint32_t lit_base, uint32_t dist_base, const LitSums& LS, uint32_t* lds_store) {
  const int lane = wave_lane();
  uint32_t* s_start = lds_store;        // [65]
  uint32_t* s_base = lds_store + 65;    // [64] bit offset of literal 0 of the stream as seen from command c
  uint32_t* W = lds_store + 132;        // [STORE_WIN_DW + 4] output window
  uint64_t wbit = (bit_cmds + cbits + lit_sum<PARTS>(LS, lit_base)) & ~(uint64_t)31;   // bit position of W[0]
  for (uint32_t j = (uint32_t)lane; j < STORE_WIN_DW + 4u; j += 64) W[j] = 0;
  bool shared_head = PARTS;             // W[0] is a dword other waves write to as well: it leaves as an atomic OR
  if (!PARTS) {
    wave_mem_barrier();                 // header atomics have landed
    if (lane == 0) W[0] = glb_atomic_or(sink_base + (wbit >> 5), 0u);
  }
  wave_sync();
  for (uint32_t base = c0; base < c1; base += 64) {
    const uint32_t i = base + (uint32_t)lane;
    const bool valid = i < c1;
    Command c;
    c.insert_len = 0; c.copy_len = 0; c.dist_extra = 0; c.cmd_prefix = 0; c.dist_prefix = 0;
    if (valid) c = s.cmds[i];
    const uint32_t ins = c.insert_len;
    const uint32_t cpy = c.copy_len & 0x1FFFFFFu;
    const bool has_dist = valid && cpy != 0 && c.cmd_prefix >= 128;
    const uint64_t dm = wave_ballot(has_dist);
    const uint32_t ins_incl = wave_incl_scan(ins);
    const uint32_t my_lit = lit_base + ins_incl - ins;
    const uint32_t my_dist = dist_base + (uint32_t)dev_popc64(dm & ((1ull << lane) - 1ull));
    // command symbol + extra bits (StoreCommandExtra :82-93), distance symbol + extra
    uint64_t xv = 0;
    uint32_t cn = 0, xn = 0, dn = 0, dxn = 0;
    SymBits cb, db;
    cb.sw = db.sw = 0; cb.nsw = db.nsw = cb.code = db.code = cb.ncode = db.ncode = 0;
    uint32_t ls = 0, le = 0;
    if (valid) {
      cb = symbol_bits<1>(s, i, c.cmd_prefix, 0);
      cn = cb.nsw + cb.ncode;
      const uint32_t copylen_code = cmd_copy_len_code(c);
      const uint32_t inscode = insert_length_code(ins);
      const uint32_t copycode = copy_length_code(copylen_code);
      const uint32_t insnumextra = k_ins_extra[inscode];
      xv = ((uint64_t)(copylen_code - k_copy_base[copycode]) << insnumextra) |
           (uint64_t)(ins - k_ins_base[inscode]);
      xn = insnumextra + k_copy_extra[copycode];
      ls = lit_sum<PARTS>(LS, my_lit);
      le = lit_sum<PARTS>(LS, my_lit + ins);
    }
    if (has_dist) {
      db = symbol_bits<2>(s, my_dist, c.dist_prefix & 0x3FFu, 0);
      dn = db.nsw + db.ncode;
      dxn = c.dist_prefix >> 10;
    }
    const uint32_t own = cn + xn + dn + dxn;
    const uint32_t own_incl = wave_incl_scan(own);
    // bits before this command = codes of earlier commands + literals before its first literal
    const uint64_t p0 = bit_cmds + cbits + (own_incl - own) + ls;
    const uint32_t total_ins = wave_bcast(ins_incl, 63);
    const uint32_t total_own = wave_bcast(own_incl, 63);
    const uint64_t span_end = bit_cmds + cbits + total_own + lit_sum<PARTS>(LS, lit_base + total_ins);
    const bool in_window = span_end - wbit <= (uint64_t)STORE_WIN_DW * 32u;
    s_start[lane] = my_lit;
    s_base[lane] = (uint32_t)(p0 + cn + xn - bit_cmds) - ls;
    if (lane == 63) s_start[64] = lit_base + total_ins;
    if (!in_window) {
      // hand the partial dword back to memory; this step uses HBM atomics
      if (lane == 0 && W[0]) glb_atomic_or(sink_base + (wbit >> 5), W[0]);
      if (lane == 0) W[0] = 0;
    }
    wave_sync();
    const uint64_t pd = p0 + cn + xn + (le - ls);
    if (in_window) {
      if (valid) {
        lds_or_bits(W, (uint32_t)(p0 - wbit), cb.nsw, cb.sw);
        lds_or_bits(W, (uint32_t)(p0 - wbit) + cb.nsw, cb.ncode, cb.code);
        lds_or_bits(W, (uint32_t)(p0 - wbit) + cn, xn, xv);
      }
      if (has_dist) {
        lds_or_bits(W, (uint32_t)(pd - wbit), db.nsw, db.sw);
        lds_or_bits(W, (uint32_t)(pd - wbit) + db.nsw, db.ncode, db.code);
        lds_or_bits(W, (uint32_t)(pd - wbit) + dn, dxn, c.dist_extra);
      }
    } else {
      if (valid) {
        or_sym(sink_base, p0, cb);
        or_bits(sink_base, p0 + cn, xn, xv);
      }
      if (has_dist) {
        or_sym(sink_base, pd, db);
        or_bits(sink_base, pd + dn, dxn, c.dist_extra);
      }
    }
    for (uint32_t L = lit_base + (uint32_t)lane; L < lit_base + total_ins; L += 64) {
      uint32_t lo = 0, hi = 63;
      while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (s_start[mid] <= L) lo = mid; else hi = mid - 1;
      }
      const uint64_t pos = bit_cmds + s_base[lo] + lit_sum<PARTS>(LS, L);
      SymBits lb;
      lb.sw = 0; lb.nsw = 0;
      if ((L % MB_LIT_MIN_BLOCK) == 0 && L != 0) {
        // possibly the first literal of a block: re-derive the block switch
        const uint32_t v = s.lits[L];
        lb = symbol_bits<0>(s, L, v & 0xFFu, v >> 8);
      } else {
        const uint32_t lc = s.lcode[L];
        lb.code = lc & 0xFFFFu;
        lb.ncode = lc >> 16;
      }
      if (in_window) {
        lds_or_bits(W, (uint32_t)(pos - wbit), lb.nsw, lb.sw);
        lds_or_bits(W, (uint32_t)(pos - wbit) + lb.nsw, lb.ncode, lb.code);
      } else {
        or_sym(sink_base, pos, lb);
      }
    }
    wave_sync();
    if (in_window) {
      // whole dwords leave the window; the partial one moves to W[0]
      const uint32_t ndw = (uint32_t)((span_end - wbit) >> 5);
      const uint32_t carry = W[ndw];
      wave_sync();
      uint32_t* dst = sink_base + (wbit >> 5);
      for (uint32_t j = (uint32_t)lane; j <= ndw; j += 64) {
        if (j < ndw) {
          if (PARTS && j == 0u && shared_head) { if (W[0]) glb_atomic_or(dst, W[0]); }
          else dst[j] = W[j];
        }
        W[j] = 0;
      }
      wave_sync();
      if (lane == 0) W[0] = carry;
      if (ndw != 0) shared_head = false;
      wbit += (uint64_t)ndw * 32u;
    } else {
      wave_mem_barrier();
      wbit = span_end & ~(uint64_t)31;
      if (!PARTS) { if (lane == 0) W[0] = glb_atomic_or(sink_base + (wbit >> 5), 0u); }
      else shared_head = true;          // (what the dword holds stays in memory: this wave's bits are OR-ed to it)
    }
    wave_sync();
    cbits += total_own;
    lit_base += total_ins;
    dist_base += (uint32_t)dev_popc64(dm);
  }
  // the last partial dword
  if (!PARTS) { if (lane == 0 && W[0]) sink_base[wbit >> 5] = W[0]; }
  else if (lane == 0 && W[0]) glb_atomic_or(sink_base + (wbit >> 5), W[0]);
  wave_sync();
  return cbits;
}

// What follows the command stream: the size check (encode.c:604-613), the raw fallback, the state update of
// WriteMetaBlockInternal / EncodeData (encode.c:598-614, 1188-1216).  total_bits: bits written behind `out +
// r.out_bytes`, the carried bits included (a compressed meta-block; ignored for one ShouldCompress refused).
DEV void store_finish(const JobParams& J, const ShardDesc& D, ShardState* S, RoundRegs& r, int32_t* dc, const uint8_t* data,
                      uint8_t* out, bool raw, uint64_t total_bits, uint64_t zero_bytes) {
  const int lane = wave_lane();
  const uint32_t start = S->mb_start, bytes = S->mb_bytes;
  const bool is_last = S->mb_is_last != 0, force_flush = S->mb_force_flush != 0;
  const bool stream = (J.flags & JOB_FLAG_STREAMT) != 0;
  if (!raw) {
    // encode.c:604-613: larger than the input + 4 bytes -> store uncompressed.
    if (!stream) {
      if ((uint64_t)bytes + 4u < (total_bits >> 3)) raw = true;
    } else {
      // (the reference compares with the bytes counted from the byte the meta-block's first bit is in, padding of
      //  the last one included: between these two bounds the answer depends on where the meta-block starts)
      const uint64_t lo = total_bits >> 3, hi = ((total_bits + 7u) >> 3) + (is_last ? 1u : 0u);
      if (lane == 0) S->mb_was_raw = (uint64_t)bytes + 4u < lo ? 1u : (uint64_t)bytes + 4u < hi ? 2u : 0u;
    }
  }
  if (raw) {
    zero_output(out, r.out_bytes, zero_bytes);
    for (int i = 0; i < 4; ++i) dc[i] = r.saved_dc[i];
    BitWriter w;
    bw_init(w, out + r.out_bytes, r.last_bytes_bits, r.last_bytes);
    emit_raw_metablock(w, data, start, bytes, is_last);
    after_metablock(r, dc, data, out, w, force_flush);
  } else {
    // after_metablock for the atomically written stream
    wave_mem_barrier();
    r.out_bytes += total_bits >> 3;
    r.last_bytes_bits = (uint32_t)(total_bits & 7u);
    uint32_t lb = 0;
    if (r.last_bytes_bits) {
      uint32_t* wp = (uint32_t*)(out + (r.out_bytes & ~(uint64_t)3));
      uint32_t word = 0;
      if (lane == 0) word = glb_atomic_or(wp, 0u);
      word = wave_bcast(word, 0);
      lb = (word >> ((r.out_bytes & 3) * 8)) & 0xFFu;
    }
    r.last_bytes = lb;
    r.last_flush_pos = r.input_pos;
    r.last_processed_pos = r.input_pos;
    if (r.last_flush_pos > 0) r.prev_byte = data[r.last_flush_pos - 1];
    if (r.last_flush_pos > 1) r.prev_byte2 = data[r.last_flush_pos - 2];
    r.ncmds = 0;
    r.nlits = 0;
    for (int i = 0; i < 4; ++i) r.saved_dc[i] = dc[i];
  }
  wave_mem_barrier();
  if (S->mb_force_flush == 1) inject_flush_padding(r, out);
  const bool done = is_last || (D.len - r.input_pos) == 0;
  wave_sync();
  if (lane == 0) {
    regs_save(r, S);
    for (int i = 0; i < 4; ++i) S->dist_cache[i] = dc[i];
    S->mb_valid = 0;
    S->mb_raw = 0;
    S->done = done ? 1u : 0u;
  }
  wave_sync();
}

DEV void store_round(const JobParams& J, const ShardDesc& D, ShardState* S,
                     const DeviceTables* T, const uint8_t* input, uint8_t* ws, uint32_t* lds_store) {
  const int lane = wave_lane();
  if (!S->mb_valid || S->error) return;
  const uint8_t* data = input + D.in_off;
  uint8_t* out = ws + D.out_off;
  RoundRegs r;
  regs_load(r, S);
  int32_t dc[4];
  for (int i = 0; i < 4; ++i) dc[i] = S->dist_cache[i];
  const uint32_t bytes = S->mb_bytes;
  const bool is_last = S->mb_is_last != 0;
  const bool raw = S->mb_raw != 0;
  // A meta-block of a tiled stream (JOB_FLAG_STREAMT) is written as if it began at bit 0 and moved to its place
  // afterwards (k_stream_place).  A raw one is not written here at all (its payload is byte aligned in the STREAM):
  // mb_was_raw = 1 tells k_stream_scan / k_stream_place, which emit it; 2 = the size comparison depends on the bit the
  // meta-block starts at, k_stream_scan decides.
  const bool stream = (J.flags & JOB_FLAG_STREAMT) != 0;
  if (stream && raw) {
    wave_sync();                 // (every lane has read the state by now)
    if (lane == 0) { S->mb_was_raw = 1; S->mb_valid = 0; }
    wave_sync();
    return;
  }

  // Everything this meta-block can touch, zeroed; then the carried bits.
  const uint64_t zero_bytes = 2ull * bytes + 520ull;
  if (r.out_bytes + zero_bytes + 16 > D.out_cap) {
    if (lane == 0) S->error = 2;
    return;
  }
  uint64_t total_bits = 0;   // relative to out + r.out_bytes, carried bits included
  uint64_t spt = SP_NOW();

  if (!raw) {
    zero_output(out, r.out_bytes, zero_bytes);
    StoreCtx s;
    StoreMeta M;
    store_ctx_init(s, M, J, D, input, ws);
    SP_ADD(S, 0, spt);
    uint32_t cmap_nrle = 0, cmap_max_prefix = 0;
    store_small_histos(s, M, lds_store, cmap_nrle, cmap_max_prefix);
    SP_ADD(S, 1, spt);
    for (uint32_t j = 0; j < M.njobs; ++j) store_code_job(s, M, j, cmap_max_prefix, lds_store);
    wave_sync();
    SP_ADD(S, 2, spt);
    store_switch_codes(s, M);
    SP_ADD(S, 3, spt);
    BitSink sink;
    sink.base = (uint32_t*)(out + (r.out_bytes & ~(uint64_t)3));
    sink.bitpos = (r.out_bytes & 3) * 8;
    const uint64_t bit0 = sink.bitpos;
    store_header(s, M, sink, r.last_bytes, r.last_bytes_bits, bytes, is_last, cmap_nrle, cmap_max_prefix);
    SP_ADD(S, 4, spt);
    const uint32_t nlits = s.info->nlits;
    const uint32_t lit_bits = store_literal_codes(s, 0u, nlits, 0u);
    if (lane == 0) s.lsum[nlits] = lit_bits;
    wave_sync();
    LitSums LS;
    LS.lsum = s.lsum; LS.part_off = nullptr; LS.nlits = nlits; LS.total = lit_bits;
    const uint32_t cbits = store_commands<false>(s, sink.base, sink.bitpos, 0u, M.ncmds, 0u, 0u, 0u, LS, lds_store);
    sink.bitpos += (uint64_t)cbits + lit_bits;
    SP_ADD(S, 5, spt);
    if (is_last && !stream) sink.bitpos = (sink.bitpos + 7u) & ~(uint64_t)7u;
    total_bits = sink.bitpos - bit0;
    wave_sync();
  }
  store_finish(J, D, S, r, dc, data, out, raw, total_bits, zero_bytes);
}

#endif  // BROTLI_AMD_CSRC_K_STORE_H_
