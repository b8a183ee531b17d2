// brotli_amd/csrc/mb_layout.h — layout of the per-shard meta-block workspace
// (block splits, histograms, prefix codes) shared by the build and store
// kernels and by the host planner.
#ifndef BROTLI_AMD_CSRC_MB_LAYOUT_H_
#define BROTLI_AMD_CSRC_MB_LAYOUT_H_

#include <stdint.h>

// Greedy splitter limits (metablock.c:723-738): a block-split decision is
// taken every min_block_size symbols; at most 256 types (+1 scratch histogram).
#define MB_LIT_MIN_BLOCK 512u
#define MB_CMD_MIN_BLOCK 1024u
#define MB_DIST_MIN_BLOCK 512u
#define MB_MAX_HISTOS 257u

struct SplitHeader {
  uint32_t num_types, num_blocks, num_histograms, alphabet;
};

// Offsets (bytes) inside the MetaBlockWork region for a meta-block of at most
// `len` bytes.  Category c: 0 literal, 1 command, 2 distance.
struct MbLayout {
  uint64_t hdr[3];      // SplitHeader
  uint64_t types[3];    // u8[max_blocks]
  uint64_t lengths[3];  // u32[max_blocks]
  uint64_t histos[3];   // u32[max_histos][alphabet]
  uint64_t depths[3];   // u8[max_histos][alphabet]
  uint64_t bits[3];     // u16[max_histos][alphabet]
  uint64_t ctx_map;     // u32[256 * 64]
  uint64_t total;
};

static inline uint64_t mb_al(uint64_t x) { return (x + 63u) & ~(uint64_t)63u; }

#if defined(__HIPCC__) || defined(BROTLI_AMD_SIMT_SIM)
__host__ __device__
#endif
static inline void mb_layout(uint64_t len, MbLayout* L) {
  const uint32_t alphabet[3] = {256u, 704u, 64u};
  const uint64_t nsym[3] = {len, len / 2 + 2, len / 2 + 2};
  const uint32_t minb[3] = {MB_LIT_MIN_BLOCK, MB_CMD_MIN_BLOCK, MB_DIST_MIN_BLOCK};
  uint64_t off = 0;
  for (int c = 0; c < 3; ++c) {
    const uint64_t max_blocks = nsym[c] / minb[c] + 2;
    // Literal histograms: (256 / nc + 1) * nc <= 269 for nc in {1,2,3,13}.
    uint64_t max_h = c == 0 ? 272u : MB_MAX_HISTOS;
    const uint64_t by_blocks = c == 0 ? max_blocks * 13u : max_blocks;
    if (by_blocks < max_h) max_h = by_blocks;
    L->hdr[c] = off;     off = mb_al(off + sizeof(SplitHeader));
    L->types[c] = off;   off = mb_al(off + max_blocks);
    L->lengths[c] = off; off = mb_al(off + max_blocks * 4);
    L->histos[c] = off;  off = mb_al(off + max_h * alphabet[c] * 4);
    L->depths[c] = off;  off = mb_al(off + max_h * alphabet[c]);
    L->bits[c] = off;    off = mb_al(off + max_h * alphabet[c] * 2);
  }
  L->ctx_map = off; off = mb_al(off + 256u * 64u * 4u);
  L->total = off;
}

static inline uint64_t mb_work_bytes(uint64_t len) {
  MbLayout L;
  mb_layout(len, &L);
  return L.total;
}

#endif  // BROTLI_AMD_CSRC_MB_LAYOUT_H_
