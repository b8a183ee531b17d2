// brotli_amd/csrc/host_plan.h — host-side job planning shared by the HIP layer
// and the test simulator: parameter derivation (quality.h), the partition plan
//  and the per-shard workspace layout in HBM.
#ifndef BROTLI_AMD_CSRC_HOST_PLAN_H_
#define BROTLI_AMD_CSRC_HOST_PLAN_H_

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "enc_types.h"
#include "mb_layout.h"
#include "k_index_layout.h"

struct JobPlan {
  JobParams J;
  std::vector<ShardDesc> shards;
  std::vector<TileDesc> tiles;       // JOB_FLAG_TILED
  std::vector<ShardDesc> chunks;     // JOB_FLAG_STREAMT: index chunks (what the index kernels run over)
  uint32_t mcap = 0;                 //   meta-blocks the stream can have at most
  uint64_t ws_bytes;
  uint64_t in_bytes;
  uint64_t max_out_bytes;
};

static inline uint64_t plan_align(uint64_t x) { return (x + 255u) & ~(uint64_t)255u; }

// Derives the encoder parameters exactly as the reference does for
// qualities 5..6 (c/enc/quality.h:59-223, encode.c:642-700).  Returns false for
// parameter combinations this library does not implement on the GPU.
// lgblock: BROTLI_PARAM_LGBLOCK (0 = the default; looked at from quality 4 on, clamped to 16 .. 24: quality.h:75-92).
static inline bool plan_params(int quality, int lgwin, uint32_t size_hint, JobParams* J, int lgblock = 0) {
  memset(J, 0, sizeof(*J));
  if (quality < 2 || quality > 9) return false;      // q0-1: k_fast.h; q10-11: other algorithms
  if (lgwin > 24) return false;                      // large window: out of scope
  if (lgwin < 10) return false;
  J->quality = quality;
  J->lgwin = lgwin;
  J->lgblock = quality < 4 ? 14 : 16;                // ComputeLgBlock, quality.h:75-92
  if (quality >= 9 && lgwin > 16) J->lgblock = lgwin < 18 ? lgwin : 18;
  if (quality >= 4 && lgblock != 0) J->lgblock = lgblock < 16 ? 16 : lgblock > 24 ? 24 : lgblock;
  J->size_hint = size_hint;
  if (quality < 5) {
    // the quickly family: quality.h:176-179, template parameters hash.h:251-279, 329-338
    J->hasher_type = (quality == 4 && size_hint >= (1u << 20)) ? 54 : quality;
    J->bucket_bits = J->hasher_type == 54 ? 20 : J->hasher_type == 4 ? 17 : 16;
    J->block_bits = J->hasher_type == 2 ? 0 : J->hasher_type == 3 ? 1 : 2;   // BUCKET_SWEEP_BITS
    J->ndist = 4;
    J->rec_bytes = 4;
    J->flush_symbols = quality < 4 ? 0x2FFFu : 0u;   // MAX_NUM_DELAYED_SYMBOLS, quality.h:33
    J->flags |= JOB_FLAG_DEEP | JOB_FLAG_QUICK;
  } else if (lgwin <= 16) {
    // the forgetful-chain family: quality.h:180-181, hash.h:296-326; table region = addr + head +
    // tiny hash + free-slot counters + the banks (k_parse_quick.h: FcGeom)
    J->hasher_type = quality < 7 ? 40 : quality < 9 ? 41 : 42;
    J->bucket_bits = 15;
    J->block_bits = 0;
    J->ndist = quality < 7 ? 4 : quality < 9 ? 10 : 16;
    J->rec_bytes = quality < 9 ? 17u : 41u;            // (263168 + 4 * 65536 | 4 * 512 * 512) / 32768, rounded up
    J->flags |= JOB_FLAG_DEEP | JOB_FLAG_QUICK;
  } else if (size_hint >= (1u << 20) && lgwin >= 19) {      // ChooseHasher, quality.h:186-204
    J->hasher_type = quality <= 6 ? 68 : 6;
    J->bucket_bits = 15;
  } else {
    J->hasher_type = quality <= 6 ? 58 : 5;
    J->bucket_bits = quality < 7 ? 14 : 15;
  }
  if (quality >= 5 && lgwin > 16) {
    J->block_bits = quality - 1;
    J->ndist = quality < 7 ? 4 : quality < 9 ? 10 : 16;
    J->rec_bytes = quality == 5 ? REC_BYTES : (8u << J->block_bits);
  }
  const int rb_bits = 1 + (lgwin > J->lgblock ? lgwin : J->lgblock);
  J->ring_mask = (1u << rb_bits) - 1u;
  J->max_backward_limit = (1u << lgwin) - 16u;
  J->spree_window = quality < 9 ? 64 : 512;          // quality.h:116-119
  J->max_metablock_size = 1u << (rb_bits < 24 ? rb_bits : 24);
  J->max_literals = J->max_metablock_size / 8;
  J->max_commands = J->max_metablock_size / 8;
  return true;
}

// Partition plan + workspace layout.  shard_size == 0: one shard (Mode S).
static inline bool plan_job(uint64_t len, int quality, int lgwin, uint32_t size_hint,
                            uint64_t shard_size, uint64_t stream_base, bool is_last,
                            JobPlan* plan, bool tables_in_ws = true, int lgblock = 0) {
  if (size_hint == 0) {
    const uint64_t tot = stream_base + len;
    size_hint = tot >= (1u << 30) ? (1u << 30) : (uint32_t)tot;
  }
  if (!plan_params(quality, lgwin, size_hint, &plan->J, lgblock)) return false;
  if (len == 0) return false;
  if (shard_size == 0 || shard_size >= len) shard_size = len;
  if (shard_size >= (3ull << 30)) return false;      // 32-bit positions, no wrap support
  const uint64_t nshards = (len + shard_size - 1) / shard_size;
  const JobParams& J = plan->J;
  // FastLog2 LUT covers every count a meta-block histogram can reach.
  const uint64_t max_mb = J.max_metablock_size < shard_size ? J.max_metablock_size : shard_size;
  plan->J.log2_lut_size = (uint32_t)(max_mb + 2 < 256 ? 256 : max_mb + 2);
  plan->shards.resize(nshards);
  uint64_t off = 0;
  uint64_t max_out = 0;
  for (uint64_t k = 0; k < nshards; ++k) {
    ShardDesc& D = plan->shards[k];
    memset(&D, 0, sizeof(D));
    const uint64_t in_off = k * shard_size;
    const uint64_t n = len - in_off < shard_size ? len - in_off : shard_size;
    D.in_off = in_off;
    D.len = (uint32_t)n;
    uint64_t so = stream_base + in_off;
    if (so > (1u << 30)) so = 1u << 30;
    if (so > J.max_backward_limit) so = J.max_backward_limit;   // encode.c:678-682
    D.stream_offset = (uint32_t)so;
    D.final_op = (k + 1 == nshards && is_last) ? 2u : 1u;
    D.cmd_cap = (uint32_t)(n / 2 + (n >> J.lgblock) + 16);
    const uint64_t mb_len = n < J.max_metablock_size ? n : J.max_metablock_size;
    // The HIP layer keeps the hash tables in their own allocation (and patches
    // table_off); the simulator carves them out of the workspace.
    D.table_off = off;
    if (tables_in_ws) off = plan_align(off + ((uint64_t)J.rec_bytes << J.bucket_bits));
    D.num_off = off;
    if (tables_in_ws) off = plan_align(off + ((uint64_t)2 << J.bucket_bits));
    D.cmds_off = off;  off = plan_align(off + (uint64_t)D.cmd_cap * sizeof(Command));
    D.lits_off = off;  off = plan_align(off + (mb_len + 8) * 2);
    D.dsym_off = off;  off = plan_align(off + (uint64_t)D.cmd_cap * 2);
    D.mb_off = off;    off = plan_align(off + mb_work_bytes(mb_len));
    // k_store scratch: block-switch codes + two words per literal
    D.scratch_off = off; off = plan_align(off + (mb_len / 256 + 64) * 8 + (2 * mb_len + 64) * 4);
    D.out_cap = 2 * n + 1024 + 16 * ((n >> J.lgblock) + 2);
    D.out_off = off;   off = plan_align(off + D.out_cap);
    max_out += D.out_cap;
  }
  plan->ws_bytes = off;
  plan->in_bytes = len;
  plan->max_out_bytes = max_out;
  return true;
}

// Gives every shard an index region (k_index.h) behind the rest of the workspace (simulator)
// or leaves the placement to the caller (ix_in_ws false: the HIP layer keeps the regions in
// allocations of their own and patches ix_off).  Returns the bytes one region needs.
static inline uint64_t plan_add_index_for(JobPlan* plan, std::vector<ShardDesc>& units, bool ix_in_ws);
static inline uint64_t plan_add_index(JobPlan* plan, bool ix_in_ws) { return plan_add_index_for(plan, plan->shards, ix_in_ws); }
static inline uint64_t plan_add_index_for(JobPlan* plan, std::vector<ShardDesc>& units, bool ix_in_ws) {
  uint64_t longest = 0;
  for (const ShardDesc& D : units) if (D.len > longest) longest = D.len;
  uint32_t slices = (uint32_t)((longest + 8191) / 8192);    // ~128 rows of 64 positions per slice
  if (slices < 1) slices = 1;
  if (slices > 64) slices = 64;
  plan->J.ix_slices = slices;
  uint32_t nb = IX_NB_MAX_LOG2 - 2u;                         // ~256 positions per bucket
  uint32_t per_bucket = 320u;
  if (const char* e = getenv("BROTLI_AMD_IX_TARGET")) { const int v = atoi(e); if (v >= 64 && v <= 4096) per_bucket = (uint32_t)v; }   // experiment knob
  while (nb < IX_NB_MAX_LOG2 && (longest >> nb) > per_bucket) ++nb;
  if ((int)nb > plan->J.bucket_bits - 4) nb = (uint32_t)plan->J.bucket_bits - 4u;
  plan->J.ix_nb_log2 = nb;
  // Few buckets per wave = many waves per shard: the waves of one shard run on one XCD
  // (kernels.h), and with ~2 shards in flight per XCD the res[] lines its buckets fill stay in
  // that L2 until they are complete (measured: profiles/r02_d_*).
  plan->J.ix_bpw = longest <= (160u << 10) ? 2u : 1u;      // (long shards: every bucket is searched block by block by its wave — 40.4 / 40.9 / 41.7 ms at 1 / 2 / 4, profiles/r05)
  plan->J.flags |= JOB_FLAG_INDEXED;
  IxLayout L;
  ix_layout(longest, slices, nb, &L);
  if (ix_in_ws) {
    uint64_t off = plan->ws_bytes;
    for (ShardDesc& D : units) { D.ix_off = off; off = plan_align(off + L.bytes); }
    plan->ws_bytes = off;
  }
  {
    // The buckets too big for LDS are searched in blocks of IX_BIG_BLOCK sorted entries by k_ix_big, which takes them
    // from eight lists (unit u is XCD u % 8's, as in k_ix_bucket): a unit has at most len / IX_BIG_BLOCK + buckets of them.
    // Fewer than eight units (JOB_FLAG_IXSPREAD): a unit's buckets are spread over the lists, (unit + bucket) % 8 — any
    // list may get up to all of them.
    if (units.size() < 8u) plan->J.flags |= JOB_FLAG_IXSPREAD;
    uint64_t per[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t u = 0; u < units.size(); ++u) per[(plan->J.flags & JOB_FLAG_IXSPREAD) ? 0u : (u & 7u)] += units[u].len / IX_BIG_BLOCK + (1ull << nb);
    uint64_t cap = 0;
    for (int x = 0; x < 8; ++x) if (per[x] > cap) cap = per[x];
    plan->J.big_cap = cap + 8;
    plan->J.ix_giant = 2048u;      // (1 MiB text shards: 39.6 / 40.6 / 41.2 / 45.0 ms at 2048 / 4096 / 8192 / 32768, 48.7 with every big bucket on the lists)
    if (const char* e = getenv("BROTLI_AMD_IX_GIANT")) { const long v = atol(e); if (v >= 320 && v <= (1 << 24)) plan->J.ix_giant = (uint32_t)v; }   // experiment knob
    plan->J.big_off = plan->ws_bytes;
    plan->ws_bytes = plan_align(plan->ws_bytes + IX_BIG_HEADER_BYTES + 8ull * plan->J.big_cap * sizeof(uint64_t));
  }
  return L.bytes;
}

// Chain tiles (k_chain.h, k_tile.h) for an indexed plan whose shards are longer than a tile: the tile table, the
// two slot buffers for the tiles' commands (behind the rest of the workspace).
// tile_kb: KiB per tile (rounded up to a power of two >= the input block; above 4096 — an environment knob may hold
// anything — there are no tiles), 0 = no tiles.  Returns the number of tiles.
static inline uint32_t plan_add_tiles(JobPlan* plan, uint32_t tile_kb, uint32_t warm_bytes) {
  if (!(plan->J.flags & JOB_FLAG_INDEXED) || tile_kb == 0 || tile_kb > 4096u) return 0;
  uint32_t tl = 10;
  while (tl < 22u && (1u << (tl - 10u)) < tile_kb) ++tl;
  if ((int)tl < plan->J.lgblock) tl = (uint32_t)plan->J.lgblock;
  if (tl > 22u) return 0;
  uint64_t longest = 0;
  for (const ShardDesc& D : plan->shards) if (D.len > longest) longest = D.len;
  // Tiles pay from three tiles per shard on: the first parse of all tiles (about one plain chain over a tile) plus the
  // sweeps cost what the plain chain needs for ~2 tiles in a row (measured, profiles/r03_g: 256 KiB shards 51 ms tiled
  // against 44 ms plain, 512 KiB 52 against 77).
  if (longest <= (2ull << tl) + 2u) return 0;
  plan->J.tile_log2 = tl;
  plan->J.tile_warm = warm_bytes < 256u ? 256u : warm_bytes > (1u << plan->J.lgblock) / 2u ? (1u << plan->J.lgblock) / 2u : warm_bytes;
  plan->J.flags |= JOB_FLAG_TILED;
  const uint32_t slot = tile_slot_cmds(tl, (uint32_t)plan->J.lgblock);
  uint64_t off = plan->ws_bytes;
  plan->tiles.clear();
  for (size_t k = 0; k < plan->shards.size(); ++k) {
    ShardDesc& D = plan->shards[k];
    const uint32_t first = D.stream_offset != 0 ? 2u : 0u;
    D.ntiles = tile_count(D.len, first, tl);
    D.tile_base = (uint32_t)plan->tiles.size();
    for (uint32_t t = 0; t < D.ntiles; ++t) { TileDesc d; d.shard = (uint32_t)k; d.t = t; plan->tiles.push_back(d); }
    D.cmds2_off = off;
    off = plan_align(off + 2ull * D.ntiles * slot * sizeof(Command));
  }
  plan->ws_bytes = off;
  return (uint32_t)plan->tiles.size();
}

// One unpartitioned quality-5 stream parsed in tiles (JOB_FLAG_STREAMT; k_tile.h): the stream as one shard whose
// tiles are its input blocks, index chunks of 1 << lgwin positions with a look-back of the same size, the stream's
// bitmaps, and room for its meta-blocks, which k_stream_cuts carves out by position on the device (the offsets of
// the stream's descriptor are the bases).  ix_in_ws false: the caller places the chunks' index regions.
// Returns false where the tiles do not apply (the caller takes the serial path).
static inline bool plan_stream(uint64_t len, int lgwin, uint32_t size_hint, uint32_t warm_bytes, bool ix_in_ws,
                               JobPlan* plan, uint64_t* ix_region_bytes = nullptr) {
  if (size_hint == 0) size_hint = len >= (1u << 30) ? (1u << 30) : (uint32_t)len;
  if (!plan_params(5, lgwin, size_hint, &plan->J)) return false;
  // (entries hold 24-bit positions of a chunk and its look-back: two windows up to lgwin 23; at lgwin 24 — what the
  //  CLI chooses for every file above 8 MiB, c/tools/brotli.c:1434-1447 — one chunk for a stream that fits the window,
  //  chunks of half a window for a longer one: below)
  if (lgwin < 17 || lgwin > 24) return false;
  if (len >= (1ull << 31)) return false;
  JobParams& J = plan->J;
  const uint32_t tl = (uint32_t)J.lgblock;
  const uint32_t ntiles = tile_count((uint32_t)len, 0u, tl);
  if (ntiles < 3u) return false;
  J.tile_log2 = tl;
  J.tile_warm = warm_bytes < 256u ? 256u : warm_bytes > (1u << tl) / 2u ? (1u << tl) / 2u : warm_bytes;
  J.chunk_log2 = (uint32_t)lgwin;
  // lgwin 24 and longer than one chunk: chunks of HALF a window (2^23 own positions + 2^23 of look-back = 24 bits).  The
  // searches with fewer than 16 same-key entries before them in their chunk — the only ones whose ring can hold something
  // from below the chunk — are the chain's, which goes on in the chunk before (k_index.h IxGeom::older, k_chain.h
  // c_search_exact, k_tile.h stream_events).  BROTLI_AMD_HALF_CHUNKS=1: the same for every window (tests: the simulator
  // cannot run streams of several 16 MiB windows).
  if (lgwin == 24 && len > (1ull << 24)) J.chunk_log2 = 23u;
  if (const char* e = getenv("BROTLI_AMD_HALF_CHUNKS")) { if (atoi(e) == 1 && lgwin - 1 >= (int)tl) J.chunk_log2 = (uint32_t)lgwin - 1u; }  const uint64_t C = 1ull << J.chunk_log2;
  J.nchunks = (uint32_t)((len + C - 1) / C);
  J.flags |= JOB_FLAG_INDEXED | JOB_FLAG_TILED | JOB_FLAG_STREAMT | JOB_FLAG_QUAD;
  J.log2_lut_size = J.max_metablock_size + 2u;
  uint64_t mcap = len / J.max_literals + 2u;
  if (mcap > ntiles) mcap = ntiles;
  plan->mcap = (uint32_t)mcap;
  plan->shards.resize(1);
  ShardDesc& D = plan->shards[0];
  memset(&D, 0, sizeof(D));
  D.len = (uint32_t)len;
  D.final_op = 2u;
  const uint64_t cmd_cap = len / 2 + ntiles + mcap + 64;
  D.cmd_cap = (uint32_t)cmd_cap;
  uint64_t off = 0;
  D.cmds_off = off;    off = plan_align(off + cmd_cap * sizeof(Command));
  D.lits_off = off;    off = plan_align(off + 2 * len + 512 * mcap + 1024);
  D.dsym_off = off;    off = plan_align(off + 2 * cmd_cap + 512 * mcap + 1024);
  D.mb_off = off;      off = plan_align(off + mcap * plan_align(mb_work_bytes(J.max_metablock_size)));
  D.scratch_off = off; off = plan_align(off + 8 * len + len / 32 + 1280 * mcap + ((uint64_t)J.max_metablock_size / 256 + 64) * 8 + 4096);
  D.out_off = off;     off = plan_align(off + 2 * len + len / 1024 + 4096 * mcap + 8192);
  D.out_cap = 2 * len + 8192;
  D.ntiles = ntiles;
  D.tile_base = 0;
  plan->tiles.clear();
  for (uint32_t t = 0; t < ntiles; ++t) { TileDesc d; d.shard = 0; d.t = t; plan->tiles.push_back(d); }
  D.cmds2_off = off;   off = plan_align(off + 2ull * ntiles * tile_slot_cmds(tl, tl) * sizeof(Command));
  J.sbm_stride = plan_align(len / 8 + 64);
  J.sbm_off = off;     off = plan_align(off + 3 * J.sbm_stride);
  J.skt_off = off;     off = plan_align(off + (uint64_t)J.nchunks * skt_chunk_bytes((uint32_t)J.bucket_bits));
  plan->ws_bytes = off;
  // the index chunks: chunk j searches [j C, (j + 1) C) (and the warm-up bytes in front of it) with
  // [(j - 1) C, j C) as look-back
  plan->chunks.resize(J.nchunks);
  const uint32_t warm_cap = (1u << tl) / 2u;
  for (uint32_t j = 0; j < J.nchunks; ++j) {
    ShardDesc& K = plan->chunks[j];
    memset(&K, 0, sizeof(K));
    const uint64_t base = j == 0 ? 0 : (uint64_t)(j - 1) * C;
    const uint64_t end = (uint64_t)(j + 1) * C < len ? (uint64_t)(j + 1) * C : len;
    K.in_off = base;
    K.len = (uint32_t)(end - base);
    K.ix_base = (uint32_t)base;
    K.ix_own = j == 0 ? 0u : (uint32_t)(C - warm_cap);
    K.ix_ownc = j == 0 ? 0u : (uint32_t)C;
    K.ix_glen = (uint32_t)len;
  }
  const uint64_t region = plan_add_index_for(plan, plan->chunks, ix_in_ws);
  if (ix_region_bytes) *ix_region_bytes = region;
  plan->in_bytes = len;
  plan->max_out_bytes = len + 8 * mcap + 1024;
  return true;
}

// BROTLI_AMD_FLAG_TAIL_FINISH: the caller's stream came in PROCESS calls that ended exactly on an input-block boundary
// and the FINISH brought nothing.  The reference then encodes the last block BEFORE it knows the stream ends
// (encode.c:1700-1712: EncodeData(is_last = 0) as soon as the block is full): when the rule of encode.c:1141-1166
// closes the meta-block there, it leaves with ISLAST = 0 and the FINISH adds the empty last meta-block
// (encode.c:1175-1184, brotli_bit_stream.c "ISLAST + ISEMPTY"); when the rule lets it wait for more, the FINISH
// flushes it with ISLAST = 1 — the one-shot stream.  The tiled stream writes the one-shot form; in the first case this
// turns it into the other.  The header of a last meta-block is ISLAST 1, ISLASTEMPTY 0, MNIBBLES, MLEN - 1; of one that is
// not last: ISLAST 0, MNIBBLES, MLEN - 1, ISUNCOMPRESSED 0 (RFC 7932 section 9.2; brotli_bit_stream.c StoreCompressed
// MetaBlockHeader) — the same number of bits: the length field moves down by one, nothing behind the header moves,
// and "11" + padding close the stream.  (A raw last meta-block is followed by the empty one either way; the reference's
// "is raw shorter" comparison, encode.c:604, is made without the last meta-block's padding then: k_tile.h stream_scan.)
// out holds (total_bits + 7) / 8 bytes (+ 1 spare); returns the new size in bytes.
static inline uint64_t stream_tail_fix(uint8_t* out, uint64_t islast_bit, uint64_t total_bits) {
  auto get = [&](uint64_t b) -> uint32_t { return (out[b >> 3] >> (b & 7u)) & 1u; };
  auto put = [&](uint64_t b, uint32_t v) { out[b >> 3] = (uint8_t)((out[b >> 3] & ~(1u << (b & 7u))) | (v << (b & 7u))); };
  const uint32_t nibbles = 4u + (get(islast_bit + 2u) | (get(islast_bit + 3u) << 1));
  const uint32_t field = 2u + 4u * nibbles;                   // MNIBBLES + MLEN - 1
  put(islast_bit, 0u);
  for (uint32_t k = 0; k < field; ++k) put(islast_bit + 1u + k, get(islast_bit + 2u + k));
  put(islast_bit + 1u + field, 0u);                           // ISUNCOMPRESSED
  uint64_t dst = total_bits;
  put(dst++, 1u);                                             // ISLAST
  put(dst++, 1u);                                             // ISEMPTY
  while ((dst & 7u) != 0u) put(dst++, 0u);
  return dst >> 3;
}

// Which parse kernel a plan runs on and in which wave layout (api_flags: BROTLI_AMD_FLAG_*
// of include/brotli_amd_hip.h: 1 NO_PAIR, 2 NO_QUAD, 4 FORCE_SLOW, 8 NO_HEADER).  Returns
// false when a shard is too long for the kernel its quality needs (*limit = the bound).
static inline bool plan_choose_kernels(JobPlan* plan, uint32_t api_flags, int num_cus, uint32_t* limit) {
  if (api_flags & 1u) plan->J.flags |= JOB_FLAG_NO_PAIR;
  if (api_flags & 4u) plan->J.flags |= JOB_FLAG_FORCE_SLOW;
  if (api_flags & 8u) plan->J.flags |= JOB_FLAG_NO_HEADER;
  if (api_flags & 32u) plan->J.flags |= JOB_FLAG_NO_LITCTX;
  // Four shards per wave (k_parse4.h) whenever no shard can wrap the ring or
  // see a candidate beyond the window.
  uint64_t longest = 0;
  for (const ShardDesc& D : plan->shards) if (D.len > longest) longest = D.len;
  if (plan->J.quality == 5) {
    if (!(api_flags & 2u) && longest <= plan->J.max_backward_limit) {
      plan->J.flags |= JOB_FLAG_QUAD;
      // Shards per wave: the kernel holds <= 128 VGPRs, i.e. 16 waves per CU stay resident.
      // While every shard can have a wave (or half of one) to itself, lock-stepping four
      // shards only makes each wait for the others' phases (measured, profiles/r01_k_*:
      // 4096 shards of 256 KiB: 315 / 288 / 280 ms with 4 / 2 / 1 shards per wave).
      const uint64_t resident = (uint64_t)num_cus * 16u;
      int v = plan->shards.size() <= resident ? 1 : plan->shards.size() <= 2 * resident ? 2 : 4;
      if (const char* e = getenv("BROTLI_AMD_QGROUPS")) v = atoi(e);   // experiment knob
      if (v == 1 || v == 2) {
        plan->J.flags |= (uint32_t)v << JOB_FLAG_GROUPS_SHIFT;
        // the lanes a shard does not need for itself search one position ahead (k_parse4.h)
        const char* d = getenv("BROTLI_AMD_DUO");
        if (!d || atoi(d) != 0) plan->J.flags |= JOB_FLAG_DUO;
      }
    }
  } else {
    // deep-bucket qualities: one shard per wave (k_parse_deep.h); shards longer than the window
    // follow the ring-end and stale-byte rules there
    (void)longest; (void)limit;
    plan->J.flags |= JOB_FLAG_DEEP;
  }
  return true;
}

// ---- format tables (brotli_amd/data/brotli_tables.bin, tools/gen_tables.c) --
struct HostTables {
  uint8_t context_lut[2048];
  uint8_t size_bits_by_length[32];
  uint32_t offsets_by_length[32];
  std::vector<uint8_t> dict;
  std::vector<uint16_t> hash_words;
  std::vector<uint8_t> hash_lengths;
};

static inline bool host_tables_load(const char* path, HostTables* t) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  char magic[4];
  uint32_t ver = 0, dsz = 0;
  bool ok = fread(magic, 1, 4, f) == 4 && !memcmp(magic, "BRTB", 4) &&
            fread(&ver, 4, 1, f) == 1 && ver == 1 &&
            fread(t->context_lut, 1, 2048, f) == 2048 &&
            fread(t->size_bits_by_length, 1, 32, f) == 32 &&
            fread(t->offsets_by_length, 4, 32, f) == 32 && fread(&dsz, 4, 1, f) == 1;
  if (ok) {
    const uint32_t padded = (dsz + 3u) & ~3u;
    t->dict.resize(padded + 64);
    t->hash_words.resize(32768);
    t->hash_lengths.resize(32768);
    ok = fread(t->dict.data(), 1, padded, f) == padded &&
         fread(t->hash_words.data(), 2, 32768, f) == 32768 &&
         fread(t->hash_lengths.data(), 1, 32768, f) == 32768;
  }
  fclose(f);
  return ok;
}

// ---- quality 1 --------------------------------------------------------------------------
// Fragments of the two-pass compressor for one run of CompressStream calls
// (encode.c:1467-1540): call k hands `call_sizes[k]` bytes, cut into fragments of
// min(1 << lgwin, bytes left in the call); a zero-byte call yields an empty
// fragment (only meaningful as the carrier of the final ISLAST bits).
struct FastPlan {
  std::vector<FastFrag> frags;
  std::vector<FastBlock> blocks;
  uint64_t cmds_base, lits_base, lsum_base, scr_base, tables_base, ws_bytes;
  uint32_t nslots;
  uint64_t max_out_bytes;
};

static inline bool plan_fast(uint64_t len, int lgwin, const uint64_t* call_sizes, size_t ncalls,
                             FastPlan* plan) {
  if (lgwin < 10 || lgwin > 24) return false;
  const uint64_t limit = (uint64_t)1 << lgwin;
  plan->frags.clear();
  plan->blocks.clear();
  uint64_t pos = 0;
  for (size_t k = 0; k < ncalls; ++k) {
    uint64_t avail = call_sizes[k];
    if (pos + avail > len) return false;
    do {
      const uint64_t n = avail < limit ? avail : limit;
      FastFrag F;
      F.in_off = pos;
      F.len = (uint32_t)n;
      F.first_block = (uint32_t)plan->blocks.size();
      F.nblocks = (uint32_t)((n + FAST_BLOCK - 1) / FAST_BLOCK);
      uint32_t bits = 8;                                  // HashTableSize, encode.c:148-154
      while (bits < 17 && ((uint64_t)1 << bits) < n) ++bits;
      F.table_bits = bits;
      for (uint32_t b = 0; b < F.nblocks; ++b) {
        FastBlock B;
        B.off_in_frag = b * FAST_BLOCK;
        B.in_off = pos + B.off_in_frag;
        B.left = (uint32_t)n - B.off_in_frag;
        B.len = B.left < FAST_BLOCK ? B.left : FAST_BLOCK;
        B.frag = (uint32_t)plan->frags.size();
        plan->blocks.push_back(B);
      }
      plan->frags.push_back(F);
      pos += n;
      avail -= n;
    } while (avail != 0);
  }
  if (pos != len) return false;
  const uint64_t nb = plan->blocks.size(), nf = plan->frags.size();
  auto al = [](uint64_t v) { return (v + 255) & ~(uint64_t)255; };
  uint64_t off = 0;
  plan->cmds_base = off;   off = al(off + 4 * len + 64);
  plan->lits_base = off;   off = al(off + len + 64);
  plan->lsum_base = off;   off = al(off + 4 * (len + nb) + 64);
  plan->scr_base = off;    off = al(off + 2 * len + 1024 * nb + 64);
  plan->nslots = (uint32_t)(nf < 8192 ? (nf ? nf : 1) : 8192);
  plan->tables_base = off; off = al(off + (uint64_t)plan->nslots * FAST_TABLE_BYTES);
  plan->ws_bytes = off;
  plan->max_out_bytes = len + 8 * nf + 64;   // a fragment never exceeds its raw form by more than 31 bits + padding (:622)
  return true;
}

// FastLog2 (c/enc/fast_log.h:51-59): a table of float literals widened to
// double below 256 (fast_log.c:14), libm log2() above.  Built with the host's
// libm so device entropy decisions see exactly the reference's values.
static inline void host_log2_lut(uint32_t n, std::vector<double>* lut) {
  lut->resize(n);
  (*lut)[0] = 0.0;
  for (uint32_t i = 1; i < n; ++i) {
    (*lut)[i] = i < 256 ? (double)(float)log2((double)i) : log2((double)i);
  }
}

// Host-memory DeviceTables (simulator).  The HIP layer does the same with
// device copies.
static inline void host_tables_fill(const HostTables& h, uint32_t log2_n,
                                    std::vector<double>* lut, DeviceTables* T) {
  host_log2_lut(log2_n, lut);
  T->context_lut = h.context_lut + (2 << 9);  // CONTEXT_UTF8, context.h:104
  T->dict = h.dict.data();
  T->dict_hash_words = h.hash_words.data();
  T->dict_hash_lengths = h.hash_lengths.data();
  T->log2_lut = lut->data();
  memcpy(T->dict_offsets_by_length, h.offsets_by_length, sizeof(T->dict_offsets_by_length));
  memcpy(T->dict_size_bits_by_length, h.size_bits_by_length, sizeof(T->dict_size_bits_by_length));
}

// ---- decoder side: the word transforms (brotli_amd/data/brotli_transforms.bin, tools/gen_transforms.c)
struct HostTransforms {
  std::vector<uint8_t> records;   // n x 8 bytes (DecTransform, k_decode.h)
  std::vector<uint8_t> text;
  uint32_t n = 0;
};
static inline bool host_transforms_load(const char* tables_path, HostTransforms* t) {
  // sits next to the tables blob
  std::string path(tables_path);
  const size_t slash = path.find_last_of('/');
  path = (slash == std::string::npos ? std::string() : path.substr(0, slash + 1)) + "brotli_transforms.bin";
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  char magic[4];
  uint32_t ver = 0, n = 0, ts = 0;
  bool ok = fread(magic, 1, 4, f) == 4 && !memcmp(magic, "BRTT", 4) && fread(&ver, 4, 1, f) == 1 && ver == 1 &&
            fread(&n, 4, 1, f) == 1 && fread(&ts, 4, 1, f) == 1 && n == 121 && ts < 4096;
  if (ok) {
    t->records.resize((size_t)n * 8);
    t->text.resize(ts + 64, 0);
    ok = fread(t->records.data(), 8, n, f) == n && fread(t->text.data(), 1, ts, f) == ts;
    t->n = n;
  }
  fclose(f);
  return ok;
}
// Dwords of arena one piece needs at most (k_decode.h): 256 literal + 256 command + 256 distance codes.
static inline uint32_t dec_arena_words_max() { return 4416u + 3u * 176u + 256u * (144u + 368u + 16u + 260u) + 64u; }

#endif  // BROTLI_AMD_CSRC_HOST_PLAN_H_
