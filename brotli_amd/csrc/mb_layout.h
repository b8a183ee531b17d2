// brotli_amd/csrc/mb_layout.h — layout of the per-shard meta-block workspace
// (symbol streams' block splits, histograms, prefix codes, tree scratch) shared
// by the build and store kernels and by the host planner.
#ifndef BROTLI_AMD_CSRC_MB_LAYOUT_H_
#define BROTLI_AMD_CSRC_MB_LAYOUT_H_

#include <stdint.h>

// Greedy splitter parameters (metablock.c:723-738): a block-split decision is
// taken every min_block_size symbols, so every block boundary is a multiple of
// min_block_size symbols ("chunks").
#define MB_LIT_MIN_BLOCK 512u
#define MB_CMD_MIN_BLOCK 1024u
#define MB_DIST_MIN_BLOCK 512u
#define MB_MAX_TYPES 256u          // BROTLI_MAX_NUMBER_OF_BLOCK_TYPES
#define MB_MAX_CMAP_SYMS 272u      // 256 clusters + 16 run-length codes

// One prefix-code construction job of the store kernel (one lane each).
struct TreeJob {
  uint64_t histo_off;   // u32[length] (workspace offset)
  uint64_t depth_off;   // u8[length]
  uint64_t bits_off;    // u16[length]
  uint32_t length;      // histogram_length
  uint32_t alphabet;    // alphabet_size (max_bits derives from it)
  uint32_t nbits;       // out: bits of the serialised code
  uint32_t pad;
};
#define MB_TREE_BUF_BYTES 512u     // >= (704 * 5 + 82) / 8
// Per-lane scratch of the tree builder: HuffmanTree pool (2 * 704 + 2 nodes of
// 8 bytes), the RLE-coded code-length sequence and its extra bits.
#define MB_LANE_SCRATCH_BYTES (8u * (2u * 704u + 2u) + 2u * 704u + 64u)

struct SplitHeader {
  uint32_t num_types, num_blocks, num_histograms, nsym;
};

struct MbInfo {
  uint32_t num_contexts;  // literal contexts: 1, 2, 3 or 13
  uint32_t map_kind;      // 0 none, 1 continuation, 2 simple UTF-8, 3 complex UTF-8
  uint32_t nlits, ndist, ncmds;
  uint32_t njobs;
  uint32_t cmap_nrle, cmap_max_prefix;
  SplitHeader split[3];   // 0 literal, 1 command, 2 distance
  // a meta-block built and written in parts (k_wide.h): what its kernels hand to each other
  uint32_t wide_bit0, wide_bit_cmds;      // bit (relative to the output dword the meta-block starts in) of its first bit / of its command stream
  uint32_t wide_lit_bits, wide_cmd_bits;  // bits of all literals (block switches included) / of all command + distance codes
};

// A meta-block in parts (k_wide.h): WIDE_CMD_PART commands / WIDE_LIT_PART literals per part, a part per wave at a time.
#define WIDE_CMD_PART 4096u
#define WIDE_LIT_PART 4096u
struct WidePart {
  uint32_t ins, adv, ndist, bits;                    // totals of the part: literals, bytes, distance symbols, bits of its command + distance codes
  uint32_t lit_off, pos_off, dist_off, bit_off;      // the same summed over the parts before it
};

struct MbLayout {
  uint64_t info;
  uint64_t types[3];     // u8[max_blocks]
  uint64_t lengths[3];   // u32[max_blocks]
  uint64_t blkmap[3];    // u16[max_chunks]: chunk -> block index
  uint64_t histos[3];    // u32[max_histos][alphabet]
  uint64_t depths[3];    // u8[max_histos][alphabet]
  uint64_t bits[3];      // u16[max_histos][alphabet]
  uint64_t small;        // block-split / context-map code work area (SmallCodes)
  uint64_t cmap_rle;     // u32[256 * 64]: context-map RLE symbols
  uint64_t rle_flags;    // u8[64][704]: OptimizeHuffmanCountsForRle scratch
  uint64_t jobs;         // TreeJob[max_jobs]
  uint64_t tree_bufs;    // u8[max_jobs][MB_TREE_BUF_BYTES]
  uint64_t lane_scratch; // u8[64][MB_LANE_SCRATCH_BYTES]
  uint64_t parts;        // WidePart[max command parts]
  uint64_t lparts;       // u32[2][max literal parts + 1]: bits of a literal part, bits of the parts before it
  uint64_t max_lparts;
  uint64_t max_histos[3];
  uint64_t max_jobs;
  uint64_t total;
};

// Codes that are built per meta-block besides the symbol histograms: block
// type / block length codes per category (brotli_bit_stream.c:760-791), the
// literal context-map code (:683-734) and the two "trivial" context-map codes
// (:794-830).
struct SmallCodes {
  uint32_t type_histo[3][MB_MAX_TYPES + 2];
  uint32_t len_histo[3][26];
  uint32_t cmap_histo[2][MB_MAX_CMAP_SYMS];   // 0 literal, 1 distance
  uint8_t type_depth[3][MB_MAX_TYPES + 2];
  uint16_t type_bits[3][MB_MAX_TYPES + 2];
  uint8_t len_depth[3][26];
  uint16_t len_bits[3][26];
  uint8_t cmap_depth[2][MB_MAX_CMAP_SYMS];
  uint16_t cmap_bits[2][MB_MAX_CMAP_SYMS];
};

#if defined(__HIPCC__)
#define MB_HD __host__ __device__
#else
#define MB_HD
#endif

MB_HD static inline uint64_t mb_al(uint64_t x) { return (x + 63u) & ~(uint64_t)63u; }

MB_HD static inline void mb_layout(uint64_t len, MbLayout* L) {
  const uint32_t alphabet[3] = {256u, 704u, 64u};
  const uint64_t nsym[3] = {len, len / 2 + 2, len / 2 + 2};
  const uint32_t minb[3] = {MB_LIT_MIN_BLOCK, MB_CMD_MIN_BLOCK, MB_DIST_MIN_BLOCK};
  uint64_t off = 0;
  uint64_t jobs = 3 * 2 + 2;
  L->info = off; off = mb_al(off + sizeof(MbInfo));
  for (int c = 0; c < 3; ++c) {
    const uint64_t max_blocks = nsym[c] / minb[c] + 2;
    // At most 256 / nc block types of nc histograms each (metablock.c:499-541).
    uint64_t max_h = MB_MAX_TYPES;
    const uint64_t by_blocks = c == 0 ? max_blocks * 13u : max_blocks;
    if (by_blocks < max_h) max_h = by_blocks;
    L->max_histos[c] = max_h;
    jobs += max_h;
    L->types[c] = off;   off = mb_al(off + max_blocks);
    L->lengths[c] = off; off = mb_al(off + max_blocks * 4);
    L->blkmap[c] = off;  off = mb_al(off + max_blocks * 2);
    L->histos[c] = off;  off = mb_al(off + max_h * alphabet[c] * 4);
    L->depths[c] = off;  off = mb_al(off + max_h * alphabet[c]);
    L->bits[c] = off;    off = mb_al(off + max_h * alphabet[c] * 2);
  }
  L->max_jobs = jobs;
  L->small = off;        off = mb_al(off + sizeof(SmallCodes));
  L->cmap_rle = off;     off = mb_al(off + 256u * 64u * 4u);
  L->rle_flags = off;    off = mb_al(off + 64u * 704u);
  L->jobs = off;         off = mb_al(off + jobs * sizeof(TreeJob));
  L->tree_bufs = off;    off = mb_al(off + jobs * MB_TREE_BUF_BYTES);
  L->lane_scratch = off; off = mb_al(off + 64u * (uint64_t)MB_LANE_SCRATCH_BYTES);
  L->parts = off;        off = mb_al(off + ((len / 2 + (len >> 14) + 64) / WIDE_CMD_PART + 2) * sizeof(WidePart));
  L->max_lparts = len / WIDE_LIT_PART + 2;
  L->lparts = off;       off = mb_al(off + 2 * L->max_lparts * 4);
  L->total = off;
}

static inline uint64_t mb_work_bytes(uint64_t len) {
  MbLayout L;
  mb_layout(len, &L);
  return L.total;
}

#endif  // BROTLI_AMD_CSRC_MB_LAYOUT_H_
