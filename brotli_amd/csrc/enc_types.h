// brotli_amd/csrc/enc_types.h — plain structs shared by host code and kernels.
//
// Vocabulary follows the reference: an encoder *shard* is one independent
// encoder instance (BROTLI_PARAM_STREAM_OFFSET contract, encode.h:231-246);
// it consumes its input in *blocks* of 1<<lgblock bytes (quality.h:75-92) and
// emits *meta-blocks* (encode.c:1141-1166).
#ifndef BROTLI_AMD_CSRC_ENC_TYPES_H_
#define BROTLI_AMD_CSRC_ENC_TYPES_H_

#include <stdint.h>

// 16-byte command, field meaning as c/enc/command.h:106-116.
struct Command {
  uint32_t insert_len;
  uint32_t copy_len;    // low 25 bits: length; high 7 bits: len_code - len
  uint32_t dist_extra;
  uint16_t cmd_prefix;
  uint16_t dist_prefix;  // low 10 bits: distance symbol; high 6: #extra bits
};
// A command the serial chain (k_chain.h) left unencoded: cmd_prefix = CMD_RAW, dist_extra = the
// distance CODE (backward_references.c:87-109); k_cmd_encode fills in the prefix fields for 64
// commands at a time before the meta-block is built.
#define CMD_RAW 0xFFFFu

// Hash-table record, one per bucket key: everything one FindLongestMatch /
// Store touches for a key sits in ONE 128-byte line (the reference keeps
// num_/tags_/buckets_ in three arrays, hash_longest_match64_simd_inc.h:100-110).
// tag2 is an extra 16-bit fingerprint of the first four bytes at the stored
// position: candidates whose first four bytes differ are rejected by the
// reference anyway (:277-278), so filtering on it first cannot change results
// and saves the random window read.
#define REC_BYTES 128
#define REC_SLOT_DW 0    // u32 slot[16]        dwords 0..15
#define REC_TAG2_DW 16   // u16 tag2[16]        dwords 16..23
#define REC_TAG_DW 24    // u8  tag[16]         dwords 24..27
#define REC_NUM_DW 28    // u16 num             dword 28 (low half)

struct JobParams {
  int32_t quality, lgwin, lgblock;
  uint32_t size_hint;
  int32_t hasher_type;  // 68 / 58 (tagged 16-slot buckets), 6 / 5 (deep buckets), 2 / 3 / 4 / 54 (quickly family)
  int32_t bucket_bits, block_bits, ndist;
  uint32_t ring_mask;           // (1 << (1 + max(lgwin, lgblock))) - 1
  uint32_t max_backward_limit;  // (1 << lgwin) - 16
  uint32_t spree_window;        // quality.h:116-119
  uint32_t max_metablock_size, max_literals, max_commands;  // encode.c:1142-1145
  uint32_t log2_lut_size;
  uint32_t flags;
  uint32_t rec_bytes;           // bytes of one bucket record (128 for 16 slots, 8 per slot otherwise)
  uint32_t ix_slices;           // JOB_FLAG_INDEXED: position slices per shard of the index kernels (k_index.h)
  uint32_t ix_nb_log2;          //   and log2 of the first-level buckets per shard
  uint32_t ix_bpw;              //   buckets one wave of k_ix_bucket works through (a power of two)
  uint32_t flush_symbols;       // qualities 2 - 3: a meta-block is cut once literals + commands reach this (encode.c:1150-1153); 0 = never
  uint32_t tile_log2;           // JOB_FLAG_TILED: log2 of the bytes of a chain tile (a multiple of the input block), k_chain.h
  uint32_t tile_warm;           //   bytes before a tile's first block that its speculative parse starts from
  uint32_t chunk_log2;          // JOB_FLAG_STREAMT: log2 of the bytes of an index chunk (lgwin: a chunk's look-back covers the window; lgwin - 1: IxGeom::older)
  uint32_t nchunks;
  uint64_t sbm_off;             // JOB_FLAG_STREAMT: the stream's three position bitmaps (unstored / as last seen / events), workspace offset
  uint64_t sbm_stride;          //   and the bytes of one of them
  uint64_t skt_off;             // JOB_FLAG_STREAMT: per chunk, seven words per bucket key (k_tile.h: StreamKeyTable), workspace offset
  uint64_t big_off;             // JOB_FLAG_INDEXED: the block lists of the buckets too big for LDS (k_index.h: IxBigHeader + 8 lists), workspace offset
  uint64_t big_cap;             //   and the records one list holds
  uint32_t ix_giant;            //   a bucket above this many entries goes to the lists (below: searched by the wave that sorted it)
  uint32_t ix_pad;
};
#define JOB_FLAG_NO_PAIR 1u   // debugging: disable the (p, p+1) speculative pair
#define JOB_FLAG_QUAD 2u       // four shards per wave (k_parse4.h); set by the host when legal
#define JOB_FLAG_FORCE_SLOW 4u // k_parse4: always take the step-by-step candidate resolve
#define JOB_FLAG_NO_HEADER 8u  // stream header already emitted: shard 0 starts byte aligned
#define JOB_FLAG_GROUPS_SHIFT 8  // k_parse4: bits 8-9 = shards per wave (0 = 4, else 1 / 2): fewer lock-stepped
                                //   shards per wave when the job is too small to fill the chip anyway
#define JOB_FLAG_DUO 32u       // k_parse4 with <= 2 shards per wave: every shard gets a second 16-lane group that
                               //   searches the next position in the same step (k_parse4.h)
#define JOB_FLAG_INDEXED 64u   // quality 5: match candidates come from a position index built by data-parallel
                               //   kernels (k_index.h); the serial chain (k_chain.h) only selects
#define JOB_FLAG_QUICK 128u    // with JOB_FLAG_DEEP: qualities 2 - 4, the HashLongestMatchQuickly family (k_parse_quick.h);
                               //   block_bits carries BUCKET_SWEEP_BITS
#define JOB_FLAG_DEEP 16u      // one shard per wave, 32 .. 256 slots per bucket (k_parse_deep.h)
#define JOB_FLAG_NO_LITCTX 4096u // BROTLI_PARAM_DISABLE_LITERAL_CONTEXT_MODELING (encode.c:561): k_build keeps one literal context
#define JOB_FLAG_TILED 1024u   // indexed job whose shards are parsed tile by tile, all tiles at once: a tile starts from a
                               //   speculated state, joins are verified, differences repaired by sweeps (k_chain.h, k_tile.h)
#define JOB_FLAG_STREAMT 8192u // one unpartitioned stream longer than the window, parsed in tiles: the index is built per
                               //   chunk of 1 << chunk_log2 positions plus a look-back of the same size (ShardDesc::ix_*),
                               //   the chain works in stream positions, meta-block cuts are part of the tiles' join state
#define JOB_FLAG_VIEWALL 16384u // (per launch) k_chain_tiles parses tiles again in a later pass (k_tile.h:
                               //   gate_walk): the bitmap holds what the other tiles left unstored — the events that told so
                               //   in the first pass are used up —, so every search is done exactly against it
#define JOB_FLAG_TAILFIN 32768u // BROTLI_AMD_FLAG_TAIL_FINISH of the call (host_plan.h: stream_tail_fix; k_tile.h: stream_scan)
#define JOB_FLAG_IXSPREAD 65536u // fewer than 8 index units (one shard, a short stream's chunks): a unit's buckets go to workgroups
                                //   of every XCD and its big blocks to all eight lists of k_ix_big — eight times the waves at work,
                                //   against keeping a unit's res[] lines in one L2, which pays only when every XCD has a unit (kernels.h)
#define JOB_FLAG_SWEEP 2048u   // (per launch) k_chain replays the tiles' previous commands and parses again only where an
                               //   event is pending

// Per-shard description written by the host.
struct ShardDesc {
  uint64_t in_off;         // first byte of the shard in the job input buffer
  uint32_t len;            // bytes of the shard available so far
  uint32_t stream_offset;  // already clamped as encode.c:678-682
  uint32_t final_op;       // 1 = FLUSH after the last byte, 2 = FINISH
  uint32_t cmd_cap;
  uint64_t table_off;      // J.rec_bytes * (1 << bucket_bits)
  uint64_t num_off;        // k_parse_deep: u16[1 << bucket_bits] bucket counters
  uint64_t cmds_off;       // Command[cmd_cap]
  uint64_t lits_off;       // u16[len + 8]: (literal | context << 8) in order
  uint64_t dsym_off;       // u16[cmd_cap]: distance symbols in order
  uint64_t mb_off;         // MetaBlockWork
  uint64_t scratch_off;    // u32[...] bit offsets (store kernel)
  uint64_t out_off;        // shard output bytes
  uint64_t out_cap;
  uint64_t ix_off;         // JOB_FLAG_INDEXED: the shard's index region (IxLayout, k_index.h)
  uint64_t cmds2_off;      // JOB_FLAG_TILED: the second command buffer (sweeps write the one the tile's commands are not in)
  uint32_t tile_base;      // JOB_FLAG_TILED: index of the shard's first tile in the job's tile arrays
  uint32_t ntiles;
  // JOB_FLAG_STREAMT, index chunks only (descriptors the index kernels run over; in_off / len = the chunk with its
  // look-back): stream position of the chunk's local position 0, first local position that is searched from this
  // chunk (what lies below is look-back: candidates only), and the stream's length (0 = an ordinary shard)
  uint32_t ix_base, ix_own, ix_glen;
  uint32_t ix_ownc;        //   first local position that belongs to this chunk alone (ix_own lies a tile's warm-up below it)
};

// ---- chain tiles (JOB_FLAG_TILED) -------------------------------------------------------------
// A tile = TB consecutive input blocks of one shard, parsed by one 16-lane group of k_chain.  Tile t > 0 of a
// shard does not know the encoder state it starts from: it parses `tile_warm` bytes of the block before it from a
// neutral state and takes what that parse arrives at the block boundary with (round 0), k_tile_verify compares it
// with what the tile before it really ended with, and a sweep parses again from the true state where they differ.
struct TileDesc { uint32_t shard, t; };
struct TileRec {
  // state at the tile's first block boundary as the tile's parse assumed it
  int32_t in_dc[4];
  uint32_t in_insert;      // literals carried over the boundary (last_insert_len_)
  uint32_t in_copy_len;    // the last command before the boundary: copy length (0 = none known) ...
  uint32_t in_code;        //   ... and distance code — what ExtendLastCommand looks at (encode.c:905-971)
  uint32_t in_ext;         // bytes ExtendLastCommand added to that command at the tile's first block
  // state the tile's parse ended with
  int32_t out_dc[4];
  uint32_t out_insert, out_copy_len, out_code;
  uint32_t out_ncmds, out_nlits;   // commands / literals of the tile
  uint32_t out_gate;       // 1: the static-dictionary gate was closed when the tile ended (hash.h:186)
  uint32_t flags;          // TILE_*
  uint32_t buf;            // command buffer the tile's commands are in (0: cmds_off, 1: cmds2_off)
  uint32_t cmd_off;        // index of the tile's first command in the shard's final command array (k_tile_verify)
  // the in-state the tile's last parse actually ran with (k_tile_verify may have replaced in_* since)
  int32_t used_dc[4];
  uint32_t used_insert, used_ext;
  uint32_t out_lpp;        // last tile: last_processed_pos_ and how the meta-block ended (bit 0 have, 1 is_last, 2 flush, 3 flush without seal)
  uint32_t out_mb;
  uint32_t nflips;         // tile 0: unstored-position bits of the shard that changed in the last k_tile_events pass
  // JOB_FLAG_STREAMT: a meta-block was cut in front of this tile (encode.c:1141-1216: the pending literals became a
  // command of their own and the next block's ExtendLastCommand found no command) — as k_stream_cuts sees it now,
  // and as the tile's last parse assumed it; cmd_off counts the cuts' insert-only commands in
  uint32_t cut, used_cut;
  // the static dictionary's two counters (hash.h:49-50, 186) moved by this much during the tile's parse: with the gate
  // taken as open (TILE_GATE_OPEN) the true counters at a tile's start are the sums over the tiles before
  uint32_t dlookups, dmatches;
  uint32_t pad;            // tile 0: successor walks of the event kernels in this pass, in units of 4096 entries (k_tile.h tile_walk_over)
  // the gate hypothesis this tile is (to be) parsed with: 0 closed, 1 open for good, 2 from the exact counters in_l / in_m
  // (the tile in which it may close: k_tile.h, gate_walk)
  uint32_t hyp, in_l, in_m;
  // JOB_FLAG_STREAMT: the meta-block that ends in front of this tile is stored uncompressed, so the distance cache the
  // tile starts from is the one that meta-block started from (encode.c:598-614) — known once the meta-blocks have been
  // built and placed (k_stream_scan / k_stream_rollback), part of the join like the rest
  uint32_t rb, rb_new;
  int32_t rb_dc[4], rb_dc_new[4];
};
#define TILE_START_EVENT 1u   // the in-state was replaced by k_tile_verify: the next sweep parses from the tile's start
#define TILE_BAD 2u           // the shard cannot be parsed in tiles (gate open, counter wrap, meta-block cut ...): serial path
#define TILE_RAN 4u           // the tile's parse has run at least once
#define TILE_CHANGED 8u       // the last sweep parsed something again in this tile
#define TILE_GATE_OPEN 64u    // (tile 0) the shard's tiles t > 0 are parsed with the static-dictionary gate taken as OPEN for
                              //   good (real English: the dictionary keeps matching) instead of closed: tile 0, which starts
                              //   from the true counters, ended with it open (k_tile_restart); k_tile_verify / k_stream_cuts
                              //   check with the summed counters that it cannot have closed inside any tile
// why a shard left the tiled path (diagnostics; on the record of the tile / of tile 0)
#define TILE_WHY_WRAP 0x100u      // a search past rank 65520 of its key (the 16-bit store counter, k_chain.h)
#define TILE_WHY_ERROR 0x200u     // the tile's parse failed (command capacity, an impossible state)
#define TILE_WHY_NO_MB 0x400u     // the last tile did not end with the shard's meta-block
#define TILE_WHY_NOT_RUN 0x800u
#define TILE_WHY_NO_CMD 0x1000u   // a tile without a command: nothing ExtendLastCommand could lengthen
#define TILE_WHY_GATE 0x2000u     // the static-dictionary gate still open behind a tile
#define TILE_WHY_CUT 0x4000u      // a meta-block would have been cut inside the shard (encode.c:1141-1166)
#define TILE_WHY_EVENTS 0x8000u   // too many unstored positions: the tiles would parse everything twice
#define TILE_WHY_TILE 0x10000u    // (tile 0: one of the shard's tiles carries a reason of its own)
#define TILE_WHY_RAW 0x20000u     // JOB_FLAG_STREAMT: the roll-backs behind raw meta-blocks did not settle

// Persistent per-shard encoder state (c/enc/state.h:49-110 subset).
struct ShardState {
  uint32_t input_pos, last_processed_pos, last_flush_pos;
  uint32_t last_insert_len, ncmds, nlits;
  int32_t dist_cache[4];
  int32_t saved_dist_cache[4];
  uint32_t dict_lookups, dict_matches;
  uint32_t last_bytes;       // u16 payload
  uint32_t last_bytes_bits;
  int32_t flint;             // BrotliEncoderFlintState
  uint32_t prev_byte, prev_byte2;
  uint32_t done;             // all input consumed and final op emitted
  uint32_t error;
  // meta-block handed from the parse kernel to the build/store kernels
  uint32_t mb_valid, mb_start, mb_bytes, mb_is_last, mb_force_flush;
  uint32_t mb_raw;           // ShouldCompress() said no (set by build kernel)
  uint32_t mb_was_raw;       // JOB_FLAG_STREAMT: the meta-block was (1) or might have to be (2) stored uncompressed
  uint32_t mb_num_contexts, mb_context_map_id;
  uint64_t out_bytes;        // whole bytes already final in the shard output
  uint64_t stat_searches, stat_pairs, stat_b_used;
  uint64_t prof[12];         // -DQ_PROFILE: cycles per phase of k_parse4
  uint32_t ix_frontier;      // k_chain.h: first position whose insertion has not been accounted for yet
  uint32_t ix_slow;          // k_chain.h: searches that took the exact in-chain path (statistics)
};

// Constant tables uploaded once per context.
struct DeviceTables {
  const uint8_t* context_lut;       // 512 B: CONTEXT_UTF8 pair of LUTs
  const uint8_t* dict;              // RFC 7932 dictionary
  const uint16_t* dict_hash_words;  // [32768]
  const uint8_t* dict_hash_lengths; // [32768]
  const double* log2_lut;           // FastLog2(v) for v < log2_lut_size
  uint32_t dict_offsets_by_length[32];
  uint8_t dict_size_bits_by_length[32];
};

// ---- quality 1: fragments of the two-pass compressor (k_fast.h) ------------------
// A fragment = one BrotliCompressFragmentTwoPass call (encode.c:1478-1513): its
// own zeroed hash table, positions relative to its first byte.  Its 128 KiB
// blocks (compress_fragment_two_pass.c:575-603) share the table, so a fragment is
// parsed by one wave, block after block; everything after the parse is per block.
#define FAST_BLOCK (1u << 17)
#define FAST_MAX_DISTANCE ((1u << 18) - 16u)   // compress_fragment_two_pass.c:29
#define FAST_TABLE_BYTES (4u << 17)            // int[1 << 17], encode.c:136-146
#define FAST_RAW 0xFFFFFFFFu

struct FastFrag {
  uint64_t in_off;        // first byte in the job input
  uint32_t len;
  uint32_t first_block, nblocks;
  uint32_t table_bits;    // 8 .. 17 (HashTableSize, encode.c:148-154); min match 4 up to 15 bits, else 6
};
struct FastBlock {
  uint64_t in_off;
  uint32_t len;
  uint32_t left;          // bytes of the fragment from this block's first byte on
  uint32_t frag;
  uint32_t off_in_frag;
};
struct FastBlockState {
  uint32_t ncmds, nlits;  // two-pass command words / literal bytes (parse kernel)
  uint32_t bits;          // bits of the block's compressed meta-block, FAST_RAW = store uncompressed
  uint32_t error;
};
struct FastFragState {
  uint64_t bits[8];       // bits the fragment occupies when it starts at bit residue r
  uint64_t start;         // first bit of the fragment in the job output
  uint32_t rewrite_mask;  // bit r: at residue r the fragment is rewritten as one raw meta-block (:622-627)
  uint32_t pad;
};
struct FastArgs {
  const FastFrag* frags;
  const FastBlock* blocks;
  FastBlockState* bstate;
  FastFragState* fstate;
  const DeviceTables* T;
  const uint8_t* input;
  uint8_t* ws;
  uint8_t* out;
  uint64_t* result;       // [0] total bits of the job output, [1] error flags
  uint64_t cmds_base, lits_base, lsum_base, scr_base, tables_base, out_cap;
  uint32_t nfrags, nblocks, nslots;
  uint32_t carry_bits, carry_value;   // bits already pending at the start of the output (stream header / last_bytes_)
  uint32_t is_last;
};

#endif  // BROTLI_AMD_CSRC_ENC_TYPES_H_
