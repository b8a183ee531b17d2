// brotli_amd/csrc/k_round.h — per-shard stream driver executed on the device:
// the part of BrotliEncoderCompressStream / EncodeData (c/enc/encode.c:985-1221,
// 1634-1722) that decides block boundaries, runs the parse over blocks and cuts
// meta-blocks.  One wavefront per shard; all scalar state is wave-uniform.
#ifndef BROTLI_AMD_CSRC_K_ROUND_H_
#define BROTLI_AMD_CSRC_K_ROUND_H_

#include "k_parse.h"

// BrotliStoreUncompressedMetaBlock (brotli_bit_stream.c:1321-1352) for a
// meta-block that starts byte-aligned after its header; the payload copy is
// done by the whole wave.  `w` must belong to lane 0's view; all lanes call.
DEV void emit_raw_metablock(BitWriter& w, const uint8_t* data, uint32_t pos,
                            uint32_t len, bool is_final) {
  const int lane = wave_lane();
  const uint32_t lg = (len == 1) ? 1u : log2floor(len - 1u) + 1u;
  const uint32_t mnibbles = (lg < 16u ? 16u : (lg + 3u)) / 4u;
  bw_put(w, 1, 0);
  bw_put(w, 2, mnibbles - 4u);
  bw_put(w, mnibbles * 4u, len - 1u);
  bw_put(w, 1, 1);
  bw_align_byte(w);
  bw_flush_bytes(w);  // nacc == 0 now
  uint8_t* dst = w.out + w.byte_pos;
  for (uint32_t i = (uint32_t)lane; i < len; i += 64u) dst[i] = data[pos + i];
  w.byte_pos += len;
  if (is_final) {
    bw_put(w, 1, 1);
    bw_put(w, 1, 1);
    bw_align_byte(w);
    bw_flush_bytes(w);
  }
  wave_sync();
}

// State update after a meta-block has been written (encode.c:1197-1216) and,
// when a flush was requested, the byte-padding block (encode.c:1356-1381) and
// the flint bookkeeping (encode.c:1686-1694).  Uniform; lane 0 writes bytes.
struct RoundRegs {
  uint32_t input_pos, last_processed_pos, last_flush_pos;
  uint32_t last_insert_len, ncmds, nlits;
  uint32_t last_bytes, last_bytes_bits;
  int32_t flint;
  uint32_t prev_byte, prev_byte2;
  int32_t saved_dc[4];
  uint64_t out_bytes;
};

DEV void regs_load(RoundRegs& r, const ShardState* S) {
  r.input_pos = S->input_pos;
  r.last_processed_pos = S->last_processed_pos;
  r.last_flush_pos = S->last_flush_pos;
  r.last_insert_len = S->last_insert_len;
  r.ncmds = S->ncmds;
  r.nlits = S->nlits;
  r.last_bytes = S->last_bytes;
  r.last_bytes_bits = S->last_bytes_bits;
  r.flint = S->flint;
  r.prev_byte = S->prev_byte;
  r.prev_byte2 = S->prev_byte2;
  r.out_bytes = S->out_bytes;
  for (int i = 0; i < 4; ++i) r.saved_dc[i] = S->saved_dist_cache[i];
}
// Called by one lane.
DEV void regs_save(const RoundRegs& r, ShardState* S) {
  S->input_pos = r.input_pos;
  S->last_processed_pos = r.last_processed_pos;
  S->last_flush_pos = r.last_flush_pos;
  S->last_insert_len = r.last_insert_len;
  S->ncmds = r.ncmds;
  S->nlits = r.nlits;
  S->last_bytes = r.last_bytes;
  S->last_bytes_bits = r.last_bytes_bits;
  S->flint = r.flint;
  S->prev_byte = r.prev_byte;
  S->prev_byte2 = r.prev_byte2;
  S->out_bytes = r.out_bytes;
  for (int i = 0; i < 4; ++i) S->saved_dist_cache[i] = r.saved_dc[i];
}

DEV void after_metablock(RoundRegs& r, const int32_t* dc, const uint8_t* data,
                         uint8_t* out, BitWriter& w, bool force_flush) {
  bw_flush_bytes(w);
  r.out_bytes += w.byte_pos;
  r.last_bytes = (uint32_t)(w.acc & 0xFF);
  r.last_bytes_bits = w.nacc;
  r.last_flush_pos = r.input_pos;
  r.last_processed_pos = r.input_pos;
  if (r.last_flush_pos > 0) r.prev_byte = data[r.last_flush_pos - 1];
  if (r.last_flush_pos > 1) r.prev_byte2 = data[r.last_flush_pos - 2];
  r.ncmds = 0;
  r.nlits = 0;
  for (int i = 0; i < 4; ++i) r.saved_dc[i] = dc[i];
  (void)out;
  (void)force_flush;
}

DEV void inject_flush_padding(RoundRegs& r, uint8_t* out) {
  if (r.last_bytes_bits != 0) {
    uint32_t seal = r.last_bytes | (0x6u << r.last_bytes_bits);
    const uint32_t seal_bits = r.last_bytes_bits + 6u;
    const uint32_t nb = (seal_bits + 7u) >> 3;
    if (wave_lane() == 0) {
      for (uint32_t i = 0; i < nb; ++i) out[r.out_bytes + i] = (uint8_t)(seal >> (8u * i));
    }
    r.out_bytes += nb;
    r.last_bytes = 0;
    r.last_bytes_bits = 0;
  }
  if (r.flint == -1) r.flint = -2;
}

// Runs the shard until a meta-block is ready for the build/store kernels
// (S->mb_valid = 1) or the shard is complete (S->done = 1).
DEV void parse_round(const JobParams& J, const ShardDesc& D, ShardState* S,
                     const DeviceTables* T, const uint8_t* input, uint8_t* ws, const CompoundDict* cd = nullptr) {
  const int lane = wave_lane();
  if (S->done || S->mb_valid || S->error) return;

  ParseCtx c;
  c.data = input + D.in_off;
  c.table = ws + D.table_off;
  c.T = T;
  c.hasher_type = J.hasher_type;
  c.bucket_bits = J.bucket_bits;
  c.ndist = J.ndist;
  c.htl = (int)hasher_htl(J.hasher_type);
  c.ring_mask = J.ring_mask;
  c.ring_size = J.ring_mask + 1u;
  c.max_backward_limit = J.max_backward_limit;
  c.stream_offset = D.stream_offset;
  c.pos_end = 0;
  c.dict_lookups = S->dict_lookups;
  c.dict_matches = S->dict_matches;
  c.pair_enabled = (J.flags & JOB_FLAG_NO_PAIR) == 0;
  c.cd = cd;
  c.gap = cd ? cd->total_size : 0u;
  for (int i = 0; i < 4; ++i) c.dc[i] = S->dist_cache[i];

  RoundRegs r;
  regs_load(r, S);

  Command* cmds = (Command*)(ws + D.cmds_off);
  uint8_t* out = ws + D.out_off;
  BlockStats st;
  st.searches = st.pairs = st.b_used = 0;
  const uint32_t block = 1u << J.lgblock;
  bool done = false, have_mb = false;
  uint32_t mb_is_last = 0, mb_force_flush = 0;

  for (;;) {
    const uint32_t avail = D.len - r.input_pos;
    const uint32_t d = r.input_pos - r.last_processed_pos;
    uint32_t remaining = d >= block ? 0u : block - d;
    if (r.flint >= 0 && remaining > (uint32_t)r.flint) remaining = (uint32_t)r.flint;
    if (remaining != 0 && avail != 0) {
      const uint32_t n = umin(remaining, avail);
      r.input_pos += n;   // CopyInputToRingBuffer: the data is already resident
      if (r.flint > 0) r.flint -= (int32_t)n;
      continue;
    }
    // BROTLI_OPERATION_PROCESS with a partly filled block: wait for more input
    // (encode.c:1700: EncodeData only when the block is full or op != PROCESS).
    if (remaining != 0 && avail == 0 && D.final_op == 0) { done = true; break; }
    bool is_last = avail == 0 && D.final_op == 2;
    // final_op 3: flush the pending input as a meta-block but leave the partial byte open
    // (EncodeData(is_last 0, force_flush 1) of ProcessMetadata, encode.c:1569-1573: the
    // padding block of InjectFlushOrPushOutput is not injected there)
    bool force_flush = avail == 0 && (D.final_op == 1 || D.final_op == 3);
    bool seal = avail == 0 && D.final_op == 1;
    if (!is_last && r.flint == 0) { r.flint = -1; force_flush = true; seal = true; }

    // ---- EncodeData ----
    uint32_t bytes = r.input_pos - r.last_processed_pos;
    uint32_t pos = r.last_processed_pos;
    if (r.ncmds + bytes / 2u + 2u > D.cmd_cap) { S->error = 1; return; }
    if (bytes >= (uint32_t)c.htl - 1u && pos >= 3u) {
      c.pos_end = r.input_pos;
      store_positions(c, pos - 3u, 3u, 1u);  // StitchToPreviousBlock
    }
    if (r.ncmds && r.last_insert_len == 0) {
      c.pos_end = r.input_pos;
      extend_last_command(c, cmds, r.ncmds, r.last_processed_pos, J.lgwin, c.dc[0], bytes, pos);
    }
    parse_block(c, pos, bytes, r.last_insert_len, cmds, r.ncmds, r.nlits, J.spree_window, st);
    {
      const uint32_t processed = r.input_pos - r.last_flush_pos;
      const bool next_fits = processed + block <= J.max_metablock_size;
      if (!is_last && !force_flush && next_fits && r.nlits < J.max_literals &&
          r.ncmds < J.max_commands) {
        if (bytes == 0 && avail == 0) { S->error = 2; return; }   // unknown final_op: fail, do not spin
        r.last_processed_pos = r.input_pos;
        continue;
      }
    }
    if (r.last_insert_len > 0) {
      if (lane == 0) cmds[r.ncmds] = make_insert_command(r.last_insert_len);
      ++r.ncmds;
      r.nlits += r.last_insert_len;
      r.last_insert_len = 0;
    }
    if (!is_last && r.input_pos == r.last_flush_pos) {
      // Flush with nothing new (encode.c:1175-1180): only the padding.
      r.last_processed_pos = r.input_pos;
      if (seal) inject_flush_padding(r, out);
      if (avail == 0) { done = true; break; }
      continue;
    }
    const uint32_t mbytes = r.input_pos - r.last_flush_pos;
    if (mbytes <= 2u) {
      // ShouldCompress() is false for <= 2 bytes (encode.c:461): the flint
      // block and other tiny tails are emitted right here.
      BitWriter w;
      bw_init(w, out + r.out_bytes, r.last_bytes_bits, r.last_bytes);
      if (mbytes == 0) {
        bw_put(w, 2, 3);  // ISLAST + ISEMPTY, encode.c:518-523
        bw_align_byte(w);
      } else {
        for (int i = 0; i < 4; ++i) c.dc[i] = r.saved_dc[i];
        emit_raw_metablock(w, c.data, r.last_flush_pos, mbytes, is_last);
      }
      after_metablock(r, c.dc, c.data, out, w, force_flush);
      if (seal) inject_flush_padding(r, out);
      if (is_last || avail == 0) { done = true; break; }
      continue;
    }
    have_mb = true;
    mb_is_last = is_last;
    mb_force_flush = force_flush ? (seal ? 1u : 2u) : 0u;   // 2: flushed meta-block without the padding block
    break;
  }

  wave_sync();
  if (lane == 0) {
    regs_save(r, S);
    S->dict_lookups = c.dict_lookups;
    S->dict_matches = c.dict_matches;
    for (int i = 0; i < 4; ++i) S->dist_cache[i] = c.dc[i];
    S->done = done ? 1u : 0u;
    S->mb_valid = have_mb ? 1u : 0u;
    if (have_mb) {
      S->mb_start = r.last_flush_pos;
      S->mb_bytes = r.input_pos - r.last_flush_pos;
      S->mb_is_last = mb_is_last;
      S->mb_force_flush = mb_force_flush;
      S->mb_raw = 0;
    }
    S->stat_searches += st.searches;
    S->stat_pairs += st.pairs;
    S->stat_b_used += st.b_used;
  }
  wave_sync();
}

// Initial state of one shard (BrotliEncoderInitState + EnsureInitialized,
// encode.c:642-700, 745-790) and its hash table (Prepare,
// ..64_simd_inc.h:81-98: every count 0xFFFF).  256 threads per block,
// grid-stride over records.
DEV void init_shard_table(uint8_t* table, uint32_t nrec, uint32_t tid, uint32_t nthreads, bool quad) {
  // One record = 128 B = 8 x 16 B; thread t clears 16-byte piece t.  The
  // counter starts at 0xFFFF: dword 28 in k_parse.h's layout, the aux bytes of
  // entries 0 and 1 (bytes 7 and 15) in k_parse4.h's.
  const uint32_t pieces = nrec * 8u;
  for (uint32_t p = tid; p < pieces; p += nthreads) {
    uint32_t v[4] = {0, 0, 0, 0};
    if (quad) {
      if ((p & 7u) == 0u) { v[1] = 0xFF000000u; v[3] = 0xFF000000u; }
    } else if ((p & 7u) == 7u) {
      v[0] = 0xFFFFu;
    }
    __builtin_memcpy(table + (size_t)p * 16u, v, 16);
  }
}

DEV void init_shard_state(const JobParams& J, const ShardDesc& D, ShardState* S) {
  ShardState s;
  __builtin_memset(&s, 0, sizeof(s));
  if (D.stream_offset != 0) {
    s.flint = 2;
    for (int i = 0; i < 4; ++i) s.dist_cache[i] = s.saved_dist_cache[i] = -16;
  } else {
    s.flint = -2;
    s.dist_cache[0] = 4; s.dist_cache[1] = 11; s.dist_cache[2] = 15; s.dist_cache[3] = 16;
    for (int i = 0; i < 4; ++i) s.saved_dist_cache[i] = s.dist_cache[i];
    // EncodeWindowBits, encode.c:191-211 (lgwin 10..24, no large window)
    if (J.flags & JOB_FLAG_NO_HEADER) { s.last_bytes = 0; s.last_bytes_bits = 0; }
    else if (J.lgwin == 16) { s.last_bytes = 0; s.last_bytes_bits = 1; }
    else if (J.lgwin == 17) { s.last_bytes = 1; s.last_bytes_bits = 7; }
    else if (J.lgwin > 17) { s.last_bytes = (uint32_t)(((J.lgwin - 17) << 1) | 1); s.last_bytes_bits = 4; }
    else { s.last_bytes = (uint32_t)(((J.lgwin - 8) << 4) | 1); s.last_bytes_bits = 7; }
  }
  *S = s;
}

#endif  // BROTLI_AMD_CSRC_K_ROUND_H_
