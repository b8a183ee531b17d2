// brotli_amd/csrc/kernels.h — __global__ entry points.  One 64-lane wavefront
// (= one workgroup) per shard for the encoder stages; plain data-parallel
// grids for table initialisation and output gathering.
#ifndef BROTLI_AMD_CSRC_KERNELS_H_
#define BROTLI_AMD_CSRC_KERNELS_H_

#include "k_round.h"
#include "k_parse4.h"
#include "k_parse_deep.h"
#include "k_parse_quick.h"
#include "k_index.h"
#include "k_chain.h"
#include "k_tile.h"

// Register budgets (waves per SIMD the compiler must leave room for).
#ifndef DEEP_WAVES
#define DEEP_WAVES 2                   // waves per SIMD k_parse_deep is compiled for (161 VGPRs + scratch at 2)
#endif
#ifndef PARSE4_WAVES
#define PARSE4_WAVES 4
#endif
#ifndef CHAIN_WAVES
#define CHAIN_WAVES 3
#endif
#ifndef IXB_WAVES
#define IXB_WAVES 4                    // k_ix_bucket: 8.7 KB of LDS per wave admit 18 waves per CU either way; at five waves per SIMD
#endif                                 //   (96 VGPRs) the search spills (profiles/r05_ix*: 29.4 against 28.3 ms)
#ifndef TILE_WAVES
#define TILE_WAVES CHAIN_WAVES         // k_chain_tiles / k_chain_sweep (experiment knob: they spill at 168 VGPRs)
#endif
#ifndef BUILD_WAVES
#define BUILD_WAVES 4
#endif
#ifndef STORE_WAVES
#define STORE_WAVES 3
#endif
#include "k_build.h"
#include "k_store.h"
#include "k_wide.h"

struct JobArgs {
  JobParams J;
  const ShardDesc* shards;
  ShardState* states;
  const DeviceTables* T;
  const uint8_t* input;
  uint8_t* ws;
  uint32_t nshards;
  uint32_t init_blocks_per_shard;
  uint32_t* counters;   // [0] shards with work left after this round, [1] faults
  const CompoundDict* cd = nullptr;   // attached dictionaries of a single stream (k_dict.h)
  // JOB_FLAG_TILED (k_chain.h, k_tile.h): the job's chain tiles and their records
  const TileDesc* tiles = nullptr;
  TileRec* trecs = nullptr;
  uint32_t ntiles = 0;
  // JOB_FLAG_STREAMT: the stream's index chunks (the index kernels run with them as `shards`)
  const ShardDesc* chunks = nullptr;
  ShardDesc* mdesc = nullptr;         //   its meta-blocks as shards of their own for k_build / k_store (k_stream_cuts fills them)
  ShardState* mstate = nullptr;
  uint64_t* moff = nullptr;           //   [mcap + 1] bit offsets of the meta-blocks in the stream
  uint8_t* sout = nullptr;            //   the stream's output
  uint32_t mcap = 0;
  uint32_t aux = 0;                   //   k_stream_cuts: 1 = describe the meta-blocks
  uint32_t wide_k = 1;                // k_wide.h: waves per meta-block in the part kernels (run_build_store sets it)
};

// grid = nshards * init_blocks_per_shard, block = 256
__global__ void __launch_bounds__(256) k_init(JobArgs a) {
  const uint32_t shard = blockIdx.x / a.init_blocks_per_shard;
  const uint32_t b = blockIdx.x % a.init_blocks_per_shard;
  if (shard >= a.nshards) return;
  const ShardDesc& D = a.shards[shard];
  if (a.J.flags & JOB_FLAG_INDEXED) {
    // no hash table: the index kernels clear what they use
  } else if ((a.J.flags & JOB_FLAG_QUICK) && a.J.hasher_type >= 40 && a.J.hasher_type <= 42) {
    // Prepare, hash_forgetful_chain_inc.h:90-118: addr = 0xCCCCCCCC, head / tiny hash / free-slot
    // counters = 0 (the banks are only ever read behind a node that was written)
    uint32_t* w = (uint32_t*)(a.ws + D.table_off);
    const uint32_t n_addr = 32768u, n_rest = (32768u * 2u + 65536u + 1024u) / 4u;
    for (uint32_t p = b * blockDim.x + threadIdx.x; p < n_addr + n_rest;
         p += a.init_blocks_per_shard * blockDim.x) w[p] = p < n_addr ? 0xCCCCCCCCu : 0u;
  } else if (a.J.flags & JOB_FLAG_QUICK) {
    // Prepare, hash_longest_match_quickly_inc.h:49-77: every slot holds position 0
    uint32_t* table = (uint32_t*)(a.ws + D.table_off);
    for (uint32_t p = b * blockDim.x + threadIdx.x; p < (1u << a.J.bucket_bits);
         p += a.init_blocks_per_shard * blockDim.x) table[p] = 0u;
  } else if (a.J.flags & JOB_FLAG_DEEP) {
    // only the counters: 0xFFFF counting down (H68 / H58), 0 counting up (H5 / H6)
    uint32_t* nums = (uint32_t*)(a.ws + D.num_off);
    const uint32_t v = a.J.hasher_type >= 58 ? 0xFFFFFFFFu : 0u;
    for (uint32_t p = b * blockDim.x + threadIdx.x; p < (1u << a.J.bucket_bits) / 2u;
         p += a.init_blocks_per_shard * blockDim.x) nums[p] = v;
  } else {
    init_shard_table(a.ws + D.table_off, 1u << a.J.bucket_bits,
                     b * blockDim.x + threadIdx.x, a.init_blocks_per_shard * blockDim.x,
                     (a.J.flags & JOB_FLAG_QUAD) != 0);
  }
  if (b == 0 && threadIdx.x == 0) init_shard_state(a.J, D, &a.states[shard]);
}

// grid = nshards, block = 64
__global__ void __launch_bounds__(64) k_parse(JobArgs a) {
  const uint32_t shard = blockIdx.x;
  if (shard >= a.nshards) return;
  parse_round(a.J, a.shards[shard], &a.states[shard], a.T, a.input, a.ws, a.cd);
  if (threadIdx.x == 0 && a.states[shard].error) glb_atomic_add(&a.counters[1], 1u);
}

// grid = ceil(nshards / 4), block = 64: four shards per wave.
__global__ void __launch_bounds__(64, PARSE4_WAVES) k_parse4(JobArgs a) {
  __shared__ uint8_t lds_dup[Q_GROUPS * Q_DUP_SLOTS];
  parse4_round(a.J, a.shards, a.states, a.nshards, a.T, a.input, a.ws, blockIdx.x, lds_dup, a.cd);
  const uint32_t gpw = q_groups_per_wave(a.J);
  const bool duo = (a.J.flags & JOB_FLAG_DUO) != 0;
  const uint32_t gi = (threadIdx.x >> 4) >> (duo ? 1 : 0);
  const uint32_t shard = blockIdx.x * gpw + gi;
  if ((threadIdx.x & (duo ? 31 : 15)) == 0 && gi < gpw && shard < a.nshards && a.states[shard].error)
    glb_atomic_add(&a.counters[1], 1u);
}

// ---- indexed quality-5 parse (k_index.h, k_chain.h) ----------------------------------------
// grid = nshards * ix_slices, block = 64
__global__ void __launch_bounds__(64) k_ix_count(JobArgs a) {
  __shared__ uint32_t lds_cnt[IX_NB_MAX];
  const uint32_t shard = blockIdx.x / a.J.ix_slices, w = blockIdx.x % a.J.ix_slices;
  if (shard >= a.nshards) return;
  ix_count(a.J, a.shards[shard], a.input, a.ws, w, lds_cnt, blockIdx.x == 0);
}
// grid = nshards, block = 64
__global__ void __launch_bounds__(64) k_ix_scan(JobArgs a) {
  if (blockIdx.x >= a.nshards) return;
  ix_scan(a.J, a.shards[blockIdx.x], a.ws);
}
// grid = nshards * ix_slices, block = 64
__global__ void __launch_bounds__(64) k_ix_scatter(JobArgs a) {
#if defined(BROTLI_AMD_SIMT_SIM)
  __shared__ uint32_t lds_sc[IX_SCATTER_LDS_WORDS];
#else
  extern __shared__ uint32_t lds_sc[];            // (IX_CHUNK + 3 << ix_nb_log2) dwords
#endif
  const uint32_t shard = blockIdx.x / a.J.ix_slices, w = blockIdx.x % a.J.ix_slices;
  if (shard >= a.nshards) return;
  ix_scatter(a.J, a.shards[shard], a.input, a.ws, w, lds_sc);
}
// grid = ceil(nshards / 8) * 8 * (buckets per shard / ix_bpw), block = 64: a wave works through
// ix_bpw buckets.  Workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md): all waves of a shard are
// given the same b % 8, so the shard's input, entries and the res[] lines its buckets fill
// together stay in ONE XCD's L2 instead of being written back partially by eight.
template <bool STREAM>
DEV void ix_bucket_kernel(const JobArgs& a, uint32_t* lds_b) {
  const uint32_t per = (1u << a.J.ix_nb_log2) / a.J.ix_bpw;
  const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  uint32_t shard = (slot / per) * 8u + xcd, b0 = (slot % per) * a.J.ix_bpw;
  if (a.J.flags & JOB_FLAG_IXSPREAD) { shard = blockIdx.x / per; b0 = (blockIdx.x % per) * a.J.ix_bpw; }     // (grid = nshards * per)
  if (shard >= a.nshards) return;
  for (uint32_t b = b0; b < b0 + a.J.ix_bpw; ++b) ix_bucket<STREAM>(a.J, a.shards[shard], a.input, a.ws, b, lds_b, shard);
}
static inline uint32_t ix_bucket_grid(const JobParams& J, uint32_t nshards) {
  if (J.flags & JOB_FLAG_IXSPREAD) return nshards * ((1u << J.ix_nb_log2) / J.ix_bpw);
  return ((nshards + 7u) / 8u) * 8u * ((1u << J.ix_nb_log2) / J.ix_bpw);
}
__global__ void __launch_bounds__(64, IXB_WAVES) k_ix_bucket(JobArgs a) {
  __shared__ uint32_t lds_b[IX_BUCKET_LDS_WORDS];
  ix_bucket_kernel<false>(a, lds_b);
}
// The same over the index chunks of a tiled stream (k_index.h: ShardDesc::ix_glen != 0).
__global__ void __launch_bounds__(64, IXB_WAVES) k_ix_bucket_s(JobArgs a) {
  __shared__ uint32_t lds_b[IX_BUCKET_LDS_WORDS];
  ix_bucket_kernel<true>(a, lds_b);
}
// The buckets too big for LDS, block by block (k_index.h: ix_big): a fixed grid of IX_BIG_GRID workgroups — 8 XCDs x
// the waves one XCD holds — takes the blocks from the lists k_ix_bucket wrote; few blocks = few waves with work.
#define IX_BIG_GRID (8u * 32u * 4u * IXB_WAVES)
// (A list that did not take all the records k_ix_bucket had for it — the planner's bound, plan_add_index_for, is
//  coupled to IX_BIG_BLOCK, ix_giant and the assignment of units to XCDs — would leave blocks unsearched and their
//  res[] / srt[] entries stale: the job is failed through the fault counter the host reads, not finished with wrong
//  bytes.  ADVICE round 5.)
DEV void ix_big_check_overflow(const JobArgs& a) {
  if (blockIdx.x < 8u && threadIdx.x == 0) {
    const uint32_t* hdr = (const uint32_t*)(a.ws + a.J.big_off);
    if (hdr[blockIdx.x] > (uint32_t)a.J.big_cap) glb_atomic_add(&a.counters[1], 1u);
  }
}
__global__ void __launch_bounds__(64, IXB_WAVES) k_ix_big(JobArgs a) {
  __shared__ uint32_t lds_b[IX_BUCKET_LDS_WORDS];
  ix_big_check_overflow(a);
  ix_big<false>(a.J, a.shards, a.input, a.ws, blockIdx.x & 7u, lds_b);
}
__global__ void __launch_bounds__(64, IXB_WAVES) k_ix_big_s(JobArgs a) {
  __shared__ uint32_t lds_b[IX_BUCKET_LDS_WORDS];
  ix_big_check_overflow(a);
  ix_big<true>(a.J, a.shards, a.input, a.ws, blockIdx.x & 7u, lds_b);
}
// grid = ceil(nshards / shards per wave), block = 64; dynamic LDS: shards per wave * C_GROUP_LDS_WORDS * 4 bytes
__global__ void __launch_bounds__(64, CHAIN_WAVES) k_chain(JobArgs a) {
#if defined(BROTLI_AMD_SIMT_SIM)
  __shared__ uint32_t lds_c[C_LDS_WORDS];
#else
  extern __shared__ uint32_t lds_c[];
#endif
  chain_round<0>(a.J, a.shards, a.states, a.nshards, a.T, a.input, a.ws, blockIdx.x, lds_c, nullptr, nullptr, 0);
  const uint32_t gpw = q_groups_per_wave(a.J);
  const uint32_t gi = threadIdx.x >> 4;
  const uint32_t shard = blockIdx.x * gpw + gi;
  if ((threadIdx.x & 15) == 0 && gi < gpw && shard < a.nshards && a.states[shard].error)
    glb_atomic_add(&a.counters[1], 1u);
}
// The same for the tiles of a tiled job (grid = ceil(ntiles / tiles per wave)): their first parse, and a sweep.
// (k_tile_verify looks at the tiles' records; errors surface there.)
__global__ void __launch_bounds__(64, TILE_WAVES) k_chain_tiles(JobArgs a) {
#if defined(BROTLI_AMD_SIMT_SIM)
  __shared__ uint32_t lds_c[C_LDS_WORDS];
#else
  extern __shared__ uint32_t lds_c[];
#endif
  chain_round<1>(a.J, a.shards, a.states, a.nshards, a.T, a.input, a.ws, blockIdx.x, lds_c, a.tiles, a.trecs, a.ntiles, a.chunks);
}
__global__ void __launch_bounds__(64, TILE_WAVES) k_chain_sweep(JobArgs a) {
#if defined(BROTLI_AMD_SIMT_SIM)
  __shared__ uint32_t lds_c[C_LDS_WORDS];
#else
  extern __shared__ uint32_t lds_c[];
#endif
  chain_round<2>(a.J, a.shards, a.states, a.nshards, a.T, a.input, a.ws, blockIdx.x, lds_c, a.tiles, a.trecs, a.ntiles, a.chunks);
}

// ---- tiled jobs (k_tile.h) ----
// grid = nshards, block = 64
__global__ void __launch_bounds__(64) k_tile_verify(JobArgs a) {
  if (blockIdx.x < a.nshards) tile_verify(a.J, a.shards[blockIdx.x], &a.states[blockIdx.x], a.trecs, a.counters);
}
// grid = nshards * ix_slices, block = 64
__global__ void __launch_bounds__(64) k_tile_events(JobArgs a) {
  const uint32_t shard = blockIdx.x / a.J.ix_slices, w = blockIdx.x % a.J.ix_slices;
  if (shard < a.nshards) tile_events(a.J, a.shards[shard], a.input, a.ws, a.trecs, w, a.counters);
}
// the gate hypothesis (k_tile.h): grid = nshards
__global__ void __launch_bounds__(64) k_tile_restart(JobArgs a) {
  if (blockIdx.x < a.nshards) tile_restart(a.shards[blockIdx.x], a.trecs, a.counters);
}
// grid = ntiles
__global__ void __launch_bounds__(64) k_tile_restart_clear(JobArgs a) {
  if (blockIdx.x >= a.ntiles) return;
  const TileDesc d = a.tiles[blockIdx.x];
  tile_restart_clear(a.J, a.shards[d.shard], a.ws, a.trecs, d.t);
}
// grid = ntiles, block = 64
__global__ void __launch_bounds__(64) k_tile_finish(JobArgs a) {
  if (blockIdx.x >= a.ntiles) return;
  const TileDesc d = a.tiles[blockIdx.x];
  tile_finish(a.J, a.shards[d.shard], &a.states[d.shard], a.ws, a.trecs, d.t);
}
// grid = nshards, block = 64: the sweeps did not settle — every tiled shard leaves the tiled path
__global__ void __launch_bounds__(64) k_tile_giveup(JobArgs a) {
  if (blockIdx.x >= a.nshards || threadIdx.x != 0) return;
  const ShardDesc& D = a.shards[blockIdx.x];
  if (D.ntiles > 1u) a.trecs[D.tile_base].flags |= TILE_BAD;
}
// grid = nshards, block = 64: the shards that left the tiled path start over for the plain chain
__global__ void __launch_bounds__(64) k_tile_fallback(JobArgs a) {
  if (blockIdx.x >= a.nshards) return;
  const ShardDesc& D = a.shards[blockIdx.x];
  if (D.ntiles <= 1u || !(a.trecs[D.tile_base].flags & TILE_BAD)) return;
  IxLayout L;
  ix_layout(D.len, a.J.ix_slices, a.J.ix_nb_log2, &L);
  uint32_t* skip = (uint32_t*)(a.ws + D.ix_off + L.skip);
  for (uint32_t i = threadIdx.x; i < (D.len + 128u + 31u) / 32u; i += 64u) skip[i] = 0;
  if (threadIdx.x == 0) init_shard_state(a.J, D, &a.states[blockIdx.x]);
}

// ---- a tiled stream (JOB_FLAG_STREAMT, k_tile.h) ----
// grid = 1, block = 64
__global__ void __launch_bounds__(64) k_stream_cuts(JobArgs a) {
  stream_cuts(a.J, a.shards[0], a.trecs, a.input, a.mdesc, a.mstate, a.mcap, a.counters, a.aux != 0, a.ws);
}
// grid = ceil(ntiles / 64), block = 64
__global__ void __launch_bounds__(64) k_stream_verify(JobArgs a) {
  stream_verify(a.J, a.shards[0], a.trecs, blockIdx.x * 64u + threadIdx.x, a.counters);
}
// grid = nchunks * ix_slices, block = 64; then grid = 1, block = 64 (one thread works)
__global__ void __launch_bounds__(64) k_stream_flips(JobArgs a) {
  const uint32_t cj = blockIdx.x / a.J.ix_slices, w = blockIdx.x % a.J.ix_slices;
  if (cj < a.J.nchunks) stream_flipcount(a.J, a.shards[0], a.ws, a.trecs, cj, w, a.counters);
}
__global__ void __launch_bounds__(64) k_stream_flipcheck(JobArgs a) {
  if (blockIdx.x == 0 && threadIdx.x == 0) stream_flipcheck(a.shards[0], a.trecs, a.counters);
}
// grid = nchunks * ix_slices, block = 64
__global__ void __launch_bounds__(64) k_stream_events(JobArgs a) {
  const uint32_t cj = blockIdx.x / a.J.ix_slices, w = blockIdx.x % a.J.ix_slices;
  if (cj < a.J.nchunks) stream_events(a.J, a.shards[0], a.chunks, cj, a.input, a.ws, a.trecs, w, a.counters);
}
// the 16-bit store counter (k_tile.h): grid = nchunks * ix_slices; keys / 64; nchunks * keys / 64; block = 64
__global__ void __launch_bounds__(64) k_stream_skclear(JobArgs a) {
  const uint32_t cj = blockIdx.x / a.J.ix_slices, w = blockIdx.x % a.J.ix_slices;
  if (cj < a.J.nchunks) stream_skclear(a.J, a.ws, cj, w);
}
__global__ void __launch_bounds__(64) k_stream_skcount(JobArgs a) {
  const uint32_t cj = blockIdx.x / a.J.ix_slices, w = blockIdx.x % a.J.ix_slices;
  if (cj < a.J.nchunks) stream_skcount(a.J, a.shards[0], a.input, a.ws, cj, w);
}
__global__ void __launch_bounds__(64) k_stream_kprefix(JobArgs a) {
  stream_kprefix(a.J, a.ws, blockIdx.x * 64u + threadIdx.x);
}
__global__ void __launch_bounds__(64) k_stream_zones(JobArgs a) {
  const uint32_t per = (1u << a.J.bucket_bits) / 64u;
  const uint32_t cj = blockIdx.x / per, kg = blockIdx.x % per;
  if (cj < a.J.nchunks) stream_zones(a.J, a.shards[0], a.chunks, a.ws, cj, kg, a.counters, a.aux != 0u);
}
// grid = ntiles, block = 64
__global__ void __launch_bounds__(64) k_stream_finish(JobArgs a) {
  if (blockIdx.x < a.ntiles) stream_finish(a.J, a.shards[0], a.ws, a.trecs, blockIdx.x);
}
// grid = 1, block = 64
__global__ void __launch_bounds__(64) k_stream_scan(JobArgs a) {
  stream_scan(a.J, a.shards[0], a.mstate, a.counters[TILE_CNT_NMB], a.moff, a.counters);
}
// grid = 1, block = 64
__global__ void __launch_bounds__(64) k_stream_rollback(JobArgs a) {
  stream_rollback(a.J, a.shards[0], a.mstate, a.counters[TILE_CNT_NMB], a.trecs, a.counters);
}
// grid = mcap * STREAM_PLACE_PARTS, block = 256
#define STREAM_PLACE_PARTS 16u
__global__ void __launch_bounds__(256) k_stream_place(JobArgs a) {
  const uint32_t m = blockIdx.x / STREAM_PLACE_PARTS, part = blockIdx.x % STREAM_PLACE_PARTS;
  if (m < a.counters[TILE_CNT_NMB]) stream_place(a.J, a.shards[0], a.mdesc, a.mstate, a.moff, a.input, a.ws, a.sout, m, part, STREAM_PLACE_PARTS, threadIdx.x, 256u);
}

// grid = nshards * CE_SPLIT, block = 64: prefix fields of the commands the chain left raw
// (CMD_RAW, enc_types.h), 64 commands per wave step.
#define CE_SPLIT 8u
__global__ void __launch_bounds__(64) k_cmd_encode(JobArgs a) {
  const uint32_t shard = blockIdx.x / CE_SPLIT, w = blockIdx.x % CE_SPLIT;
  if (shard >= a.nshards) return;
  const uint32_t n = a.states[shard].ncmds;
  Command* cmds = (Command*)(a.ws + a.shards[shard].cmds_off);
  for (uint32_t i = w * 64u + threadIdx.x; i < n; i += 64u * CE_SPLIT) {
    const Command c = cmds[i];
    if (c.cmd_prefix != CMD_RAW) continue;
    const uint32_t m = c.copy_len >> 25;
    const int32_t delta = (int8_t)((uint8_t)(m | ((m & 0x40) << 1)));
    cmds[i] = make_command(c.insert_len, c.copy_len & 0x1FFFFFFu, delta, c.dist_extra);
  }
}

// grid = nshards, block = 64: one shard per wave, E = slots / 64 entries per lane.
template <int E>
__global__ void __launch_bounds__(64, DEEP_WAVES) k_parse_deep(JobArgs a) {
  __shared__ uint8_t lds_dup[D_DUP_SLOTS];
  const uint32_t shard = blockIdx.x;
  if (shard >= a.nshards) return;
  parse_deep_round<E>(a.J, a.shards[shard], &a.states[shard], a.T, a.input, a.ws, lds_dup, a.cd);
  if (threadIdx.x == 0 && a.states[shard].error) glb_atomic_add(&a.counters[1], 1u);
}

// grid = nshards, block = 64: one shard per wave (qualities 2 - 4).
__global__ void __launch_bounds__(64) k_parse_quick(JobArgs a) {
  const uint32_t shard = blockIdx.x;
  if (shard >= a.nshards) return;
  parse_quick_round(a.J, a.shards[shard], &a.states[shard], a.T, a.input, a.ws, a.cd);
  if (threadIdx.x == 0 && a.states[shard].error) glb_atomic_add(&a.counters[1], 1u);
}

// grid = nshards, block = 64: block splits, histograms, prefix codes.
__global__ void __launch_bounds__(64, BUILD_WAVES) k_build(JobArgs a) {
  const uint32_t shard = blockIdx.x;
  if (shard >= a.nshards) return;
  __shared__ uint32_t lds[BUILD_LDS_WORDS];
  __shared__ double lds_ent[4 + 3 * 13 + 1];
  __shared__ double lds_last[2 * 13];
  __shared__ double lds_terms[BUILD_TERMS + 2];
  build_round(a.J, a.shards[shard], &a.states[shard], a.T, a.input, a.ws, lds, lds_ent, lds_last, lds_terms);
}

// grid = nshards, block = 64: bit-stream emission of the pending meta-block.
__global__ void __launch_bounds__(64, STORE_WAVES) k_store(JobArgs a) {
  const uint32_t shard = blockIdx.x;
  if (shard >= a.nshards) return;
  __shared__ uint32_t lds_store[STORE_LDS_WORDS];
  store_round(a.J, a.shards[shard], &a.states[shard], a.T, a.input, a.ws, lds_store);
  if (threadIdx.x == 0) {
    if (a.states[shard].error) glb_atomic_add(&a.counters[1], 1u);
    else if (!a.states[shard].done) glb_atomic_add(&a.counters[0], 1u);
  }
}

// ---- a long meta-block on many waves (k_wide.h); grids: nshards, nshards * WIDE_K ("x K") or nshards * 3 ----
#define WIDE_BUILD_LDS \
  __shared__ uint32_t lds[BUILD_LDS_WORDS]; \
  __shared__ double lds_ent[4 + 3 * 13 + 1]; \
  __shared__ double lds_last[2 * 13]; \
  __shared__ double lds_terms[BUILD_TERMS + 2];
__global__ void __launch_bounds__(64) k_wide_head(JobArgs a) {
  const uint32_t m = blockIdx.x;
  if (m >= a.nshards) return;
  WIDE_BUILD_LDS
  wide_head(a.J, a.shards[m], &a.states[m], a.T, a.input, a.ws, lds, lds_ent, lds_last, lds_terms);
}
__global__ void __launch_bounds__(64) k_wide_count(JobArgs a) {
  const uint32_t m = blockIdx.x / a.wide_k;
  if (m >= a.nshards) return;
  wide_count(a.J, a.shards[m], &a.states[m], a.ws, blockIdx.x % a.wide_k, a.wide_k);
}
__global__ void __launch_bounds__(64) k_wide_scan1(JobArgs a) {
  const uint32_t m = blockIdx.x;
  if (m >= a.nshards) return;
  wide_scan1(a.J, a.shards[m], &a.states[m], a.ws);
}
__global__ void __launch_bounds__(64) k_wide_streams(JobArgs a) {
  const uint32_t m = blockIdx.x / a.wide_k;
  if (m >= a.nshards) return;
  __shared__ uint32_t lds[160 + 144];      // [0, 129) the step's literal starts / sources, [160, 304) the context tables (k_build.h)
  wide_streams(a.J, a.shards[m], &a.states[m], a.T, a.input, a.ws, blockIdx.x % a.wide_k, a.wide_k, lds);
}
__global__ void __launch_bounds__(64) k_wide_split(JobArgs a) {
  const uint32_t m = blockIdx.x / 3u;
  if (m >= a.nshards) return;
  WIDE_BUILD_LDS
  wide_split(a.J, a.shards[m], &a.states[m], a.T, a.input, a.ws, blockIdx.x % 3u, lds, lds_ent, lds_last, lds_terms);
}
__global__ void __launch_bounds__(64) k_wide_prep(JobArgs a) {
  const uint32_t m = blockIdx.x;
  if (m >= a.nshards) return;
  WIDE_BUILD_LDS
  wide_prep(a.J, a.shards[m], &a.states[m], a.T, a.input, a.ws, lds, lds_ent, lds_last, lds_terms);
}
__global__ void __launch_bounds__(64) k_wide_codes(JobArgs a) {
  const uint32_t m = blockIdx.x / a.wide_k;
  if (m >= a.nshards) return;
  __shared__ uint32_t lds_store[STORE_LDS_WORDS];
  wide_codes(a.J, a.shards[m], &a.states[m], a.input, a.ws, blockIdx.x % a.wide_k, a.wide_k, lds_store);
}
__global__ void __launch_bounds__(64) k_wide_header(JobArgs a) {
  const uint32_t m = blockIdx.x;
  if (m >= a.nshards) return;
  wide_header(a.J, a.shards[m], &a.states[m], a.input, a.ws);
}
__global__ void __launch_bounds__(64) k_wide_bits(JobArgs a) {
  const uint32_t m = blockIdx.x / a.wide_k;
  if (m >= a.nshards) return;
  wide_bits(a.J, a.shards[m], &a.states[m], a.input, a.ws, blockIdx.x % a.wide_k, a.wide_k);
}
__global__ void __launch_bounds__(64) k_wide_scan2(JobArgs a) {
  const uint32_t m = blockIdx.x;
  if (m >= a.nshards) return;
  wide_scan2(a.J, a.shards[m], &a.states[m], a.ws);
}
__global__ void __launch_bounds__(64) k_wide_emit(JobArgs a) {
  const uint32_t m = blockIdx.x / a.wide_k;
  if (m >= a.nshards) return;
  __shared__ uint32_t lds_store[STORE_LDS_WORDS];
  wide_emit(a.J, a.shards[m], &a.states[m], a.input, a.ws, blockIdx.x % a.wide_k, a.wide_k, lds_store);
}
__global__ void __launch_bounds__(64) k_wide_tail(JobArgs a) {
  const uint32_t m = blockIdx.x;
  if (m >= a.nshards) return;
  wide_tail(a.J, a.shards[m], &a.states[m], a.input, a.ws);
  if (threadIdx.x == 0) {
    if (a.states[m].error) glb_atomic_add(&a.counters[1], 1u);
    else if (!a.states[m].done) glb_atomic_add(&a.counters[0], 1u);
  }
}

// The build + store stage of a round over `n` meta-blocks (shards of a plan, or the meta-blocks of a tiled stream):
// one wave per meta-block (k_build, k_store), or — `wide` — the kernels of k_wide.h.  One routine for the HIP layer
// and the simulator's drivers (R::operator()(kernel, args, grid, block) launches; `mid` runs between the half that
// models and the half that writes, where the HIP layer records its event).
template <class R, class Mid>
static inline void run_build_store(R& run, const JobArgs& m0, uint32_t n, uint32_t wide, Mid mid) {
  JobArgs m = m0;
  m.wide_k = wide < 1u ? 1u : wide > WIDE_K_MAX ? WIDE_K_MAX : wide;     // wide: 0 = one wave does it all, else the waves per meta-block
  const uint32_t WIDE_K = m.wide_k;
  if (!wide) {
    run(k_build, m, n, 64u);
    mid();
    run(k_store, m, n, 64u);
    return;
  }
  run(k_wide_head, m, n, 64u);
  run(k_wide_count, m, n * WIDE_K, 64u);
  run(k_wide_scan1, m, n, 64u);
  run(k_wide_streams, m, n * WIDE_K, 64u);
  run(k_wide_split, m, n * 3u, 64u);
  run(k_wide_prep, m, n, 64u);
  mid();
  run(k_wide_codes, m, n * WIDE_K, 64u);
  run(k_wide_header, m, n, 64u);
  run(k_wide_bits, m, n * WIDE_K, 64u);
  run(k_wide_scan2, m, n, 64u);
  run(k_wide_emit, m, n * WIDE_K, 64u);
  run(k_wide_tail, m, n, 64u);
}

// grid = 1, block = 1024: exclusive scan of the shard output sizes.
__global__ void __launch_bounds__(1024) k_scan_sizes(JobArgs a, uint64_t* scan, uint64_t* sizes_out) {
  __shared__ uint64_t part[1024];
  const uint32_t t = threadIdx.x;
  const uint32_t per = (a.nshards + 1023u) / 1024u;
  const uint32_t lo = t * per, hi = lo + per < a.nshards ? lo + per : a.nshards;
  uint64_t sum = 0;
  for (uint32_t k = lo; k < hi; ++k) sum += a.states[k].out_bytes;
  part[t] = sum;
  block_sync();
  if (t == 0) {
    uint64_t run = 0;
    for (uint32_t i = 0; i < 1024; ++i) { const uint64_t v = part[i]; part[i] = run; run += v; }
    scan[a.nshards] = run;
  }
  block_sync();
  uint64_t run = part[t];
  for (uint32_t k = lo; k < hi; ++k) {
    scan[k] = run;
    const uint64_t n = a.states[k].out_bytes;
    if (sizes_out) sizes_out[k] = n;
    run += n;
  }
}

// grid = nshards * blocks_per_shard, block = 256: 16 bytes per lane per step.
__global__ void __launch_bounds__(256) k_gather(JobArgs a, const uint64_t* scan, uint8_t* out,
                                                uint32_t blocks_per_shard) {
  const uint32_t shard = blockIdx.x / blocks_per_shard;
  const uint32_t part = blockIdx.x % blocks_per_shard;
  if (shard >= a.nshards) return;
  const uint64_t n = a.states[shard].out_bytes;
  const uint8_t* src = a.ws + a.shards[shard].out_off;
  uint8_t* dst = out + scan[shard];
  const uint64_t chunk = ((n + blocks_per_shard - 1) / blocks_per_shard + 15u) & ~(uint64_t)15u;
  const uint64_t lo = (uint64_t)part * chunk;
  const uint64_t hi = lo + chunk < n ? lo + chunk : n;
  for (uint64_t i = lo + (uint64_t)threadIdx.x * 16u; i < hi; i += 256u * 16u) {
    if (i + 16u <= hi) {
      uint32_t v[4];
      __builtin_memcpy(v, src + i, 16);
      __builtin_memcpy(dst + i, v, 16);
    } else {
      for (uint64_t j = i; j < hi; ++j) dst[j] = src[j];
    }
  }
}

// ---- quality 1 (k_fast.h) ---------------------------------------------------------------
#include "k_fast.h"

// grid = nslots (<= nfrags), block = 64: wave w takes fragments w, w + nslots, ... and
// owns hash-table slot w.
__global__ void __launch_bounds__(64) k_fast_parse(FastArgs a) {
  uint32_t* table = (uint32_t*)(a.ws + a.tables_base + (uint64_t)blockIdx.x * FAST_TABLE_BYTES);
  __shared__ uint8_t lds_sb[FAST_SB_SLOTS];
  for (uint32_t f = blockIdx.x; f < a.nfrags; f += a.nslots) fast_parse_fragment(a, f, table, lds_sb);
}

// grid = nblocks, block = 64
__global__ void __launch_bounds__(64) k_fast_store(FastArgs a) {
  __shared__ uint32_t lds_fast[FAST_STORE_LDS_WORDS];
  if (blockIdx.x < a.nblocks) fast_store_block(a, blockIdx.x, lds_fast);
}

// grid = ceil(8 * nfrags / 256), block = 256
__global__ void __launch_bounds__(256) k_fast_sizes(FastArgs a) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < 8u * a.nfrags) fast_fragment_sizes(a, t >> 3, t & 7u);
}

// grid = 1, block = 64
__global__ void __launch_bounds__(64) k_fast_scan(FastArgs a) { fast_scan_fragments(a); }

// grid = nblocks, block = 256
__global__ void __launch_bounds__(256) k_fast_emit(FastArgs a) {
  if (blockIdx.x < a.nblocks) fast_emit_block(a, blockIdx.x, threadIdx.x, blockDim.x);
}

// ---- decoder (k_decode.h) -----------------------------------------------------------------
#include "k_decode.h"

// grid = npieces, block = 64: one piece per wave.  W = waves per SIMD the register budget leaves room for
// (4: everything in registers; 8: all 8192 pieces of a 1 GiB / 128 KiB plan resident at once, some scratch).
#ifndef DECODE_WAVES
#define DECODE_WAVES 4
#endif
template <int W>
__global__ void __launch_bounds__(64, W) k_decode(DecArgs a) {
  __shared__ uint32_t lds_dec[DEC_LDS_WORDS];
  if (blockIdx.x < a.npieces) decode_piece(a, blockIdx.x, lds_dec);
}

#endif  // BROTLI_AMD_CSRC_KERNELS_H_
