// brotli_amd/csrc/k_index_layout.h — constants and the HBM layout of a shard's position
// index (k_index.h), shared by the kernels and the host-side planner (host_plan.h).
#ifndef BROTLI_AMD_CSRC_K_INDEX_LAYOUT_H_
#define BROTLI_AMD_CSRC_K_INDEX_LAYOUT_H_

#include <stdint.h>

#if defined(__HIPCC__) || defined(BROTLI_AMD_SIMT_SIM)
#define IX_HD __host__ __device__
#else
#define IX_HD
#endif

#define IX_NB_MAX_LOG2 10u                    // first-level buckets by the top key bits: 2^8 .. 2^10 per shard
#define IX_NB_MAX (1u << IX_NB_MAX_LOG2)      //   (JobParams::ix_nb_log2, chosen so that a bucket holds ~256 positions)
#ifndef IX_LROWS
#define IX_LROWS 5u                           // a bucket of <= 64 * IX_LROWS entries is sorted and searched in LDS: 6.6 KB per
                                              //   wave, five waves per SIMD (8 rows: 11 KB, 3.5 waves, +5.7 ms per GiB — profiles/r03_c)
#endif
#define IX_BIG_BLOCK 256u                     // a bucket too big for LDS is searched in blocks of this many sorted entries (k_ix_big)
#define IX_BIG_HEADER_BYTES 256u              //   the lists' header (words): [0, 8) records per list, [8, 16) the next record to hand out,
                                              //   16 / 17 buckets the LDS-atomic placement had to place again (small / big: diagnostics)
#define IX_CAP 40u                            // bytes compared per candidate by ix_bucket
#define IX_KIND_NONE 0u
#define IX_KIND_EXACT 1u                      // (len, distance) is the bucket loop's result
#define IX_KIND_LONG 2u                       // one candidate matches >= IX_CAP bytes and wins however long it is
#define IX_KIND_SLOW 3u                       // not decidable here: the chain searches this position itself
// res[p], high word: sorted index (24 bits) | successors of p in its key run, capped at 16 (5 bits) << 24 | flags
#define IX_NSUCC_SHIFT 24u
#define IX_TAINT 0x20000000u                  // set by the chain: one of the 16 predecessors of p in its key run was NOT
                                              //   stored by the parse, so the index result of p does not hold (k_chain.h)
#define IX_FULLRUN 0x40000000u                // the window of p held EVERY same-key predecessor (rank <= 16): whatever the parse
                                              //   left unstored, the reference's ring is a subset of it — a tainted result still
                                              //   holds when it found nothing, or when its winner was stored (k_chain.h)
#define IX_DANGER 0x80000000u                 // the bucket counter may have wrapped (>= 65520 stores of one key)

// Entry of the sort as it travels through HBM: position | (low bits of the key: the ones the first level did not
// sort by) << 24 (4 bytes) — a bucket too big for LDS is sorted by them without looking at the input again.  The
// bucket pass re-reads the 16 bytes at the position from the shard's input (which sits in the L2) once and keeps
// {position | tag << 24, key, bytes 0..7, bytes 8..15} per entry in registers / LDS; srt[] gets position | tag << 24.
struct IxEntry { uint32_t w0, w1; uint64_t d, d2; };

// Index region of one shard, offsets relative to ShardDesc::ix_off.
struct IxLayout { uint64_t cnt, skip, skip_prev, ev, srt, res, ent, ent2, bytes; };
static inline IX_HD uint64_t ix_align(uint64_t x) { return (x + 255u) & ~(uint64_t)255u; }
static inline IX_HD void ix_layout(uint64_t n, uint32_t slices, uint32_t nb_log2, IxLayout* L) {
  uint64_t off = 0;
  L->cnt = off;   off = ix_align(off + 4ull * (((uint64_t)slices << nb_log2) + 2));
  L->skip = off;  off = ix_align(off + n / 8 + 32);
  L->skip_prev = off;  off = ix_align(off + n / 8 + 32);    // JOB_FLAG_TILED: `skip` as the last k_tile_events pass saw it
  L->ev = off;    off = ix_align(off + n / 8 + 32);         // JOB_FLAG_TILED: positions whose search has to be done again
  L->srt = off;   off = ix_align(off + 4 * n + 16);
  L->res = off;   off = ix_align(off + 8 * n + 16);
  L->ent = off;   off = ix_align(off + 4 * n + 16);
  L->ent2 = off;  off = ix_align(off + 4 * n + 16);
  L->bytes = off;
}

// ---- a tiled stream (JOB_FLAG_STREAMT): per chunk and bucket key, what the 16-bit store counter needs (k_tile.h) ----
// run start / run length in the chunk's sorted array, entries of the run in the chunk's own part, how many of those
// the parse did not store, stores of the key before the chunk's look-back, and the sorted-index range the last pass
// marked counter-wrap positions in.
#define SKT_RS 0u
#define SKT_RL 1u
#define SKT_OWN 2u
#define SKT_SK 3u
#define SKT_B 4u
#define SKT_ZLO 5u
#define SKT_ZHI 6u
#define SKT_DIRTY 7u                          // != 0: k_stream_zones has to walk the run again — a store bit of the run changed
                                              //   (k_stream_events) or the stores before it did (k_stream_kprefix); cleared by the walk
#define SKT_WORDS 8u
static inline IX_HD uint64_t skt_chunk_bytes(uint32_t bucket_bits) { return (uint64_t)SKT_WORDS * 4u << bucket_bits; }

// ---- chain tiles (JOB_FLAG_TILED, enc_types.h) ----------------------------------------------
// Tile t of a shard of n bytes whose input blocks start at `first` (2 behind a stream offset: the "flint" bytes are a
// block of their own, encode.c:1686-1694): tile 0 = [0, first + T), tile t = [first + t T, first + (t + 1) T), T = 1 << tile_log2.
static inline IX_HD uint32_t tile_count(uint32_t n, uint32_t first, uint32_t tile_log2) {
  return n <= first + 1u ? 1u : (uint32_t)((((uint64_t)(n - first)) + ((1ull << tile_log2) - 1u)) >> tile_log2);
}
static inline IX_HD uint32_t tile_lo(uint32_t first, uint32_t t, uint32_t tile_log2) { return t == 0 ? 0u : first + (t << tile_log2); }
static inline IX_HD uint32_t tile_hi(uint32_t n, uint32_t first, uint32_t t, uint32_t tile_log2) {
  const uint64_t e = (uint64_t)first + (((uint64_t)t + 1u) << tile_log2);
  return e < n ? (uint32_t)e : n;
}
// Commands a tile's slot holds: index 0 = the copy of the last command before the tile ("ghost", what
// ExtendLastCommand may lengthen), then the tile's own (<= bytes / 2 per block + one trailing insert).
static inline IX_HD uint32_t tile_slot_cmds(uint32_t tile_log2, uint32_t lgblock) {
  return (1u << (tile_log2 - 1u)) + (1u << (tile_log2 - lgblock)) + 16u;
}

#endif  // BROTLI_AMD_CSRC_K_INDEX_LAYOUT_H_
