// brotli_amd/csrc/k_decode.h — a Brotli DECODER on the device: round trips of
// what the encoder kernels produce, at rate, without going through the host decoder.  Written from
// RFC 7932 (sections 3 - 10 and Appendix A / B); it decodes any stream without the large-window
// extension, so the reference encoder's own output at every quality is the test set
// (tests/test_sim_decode.py, tests/test_gpu_decode.py).  What the reference's c/dec does as a
// suspendable byte-at-a-time state machine over a ring buffer is laid out differently here:
//
//  * The unit of work is a PIECE: a byte-aligned part of a stream that can be decoded on its own —
//    a whole stream, or one shard of a partition plan (every shard ends with a flush, begins with its
//    own meta-block header, never copies from before its first byte — the encoder instance that made it
//    had nothing there — and its dictionary references are measured from the stream position, which
//    the piece is told).  One wave per piece; thousands of pieces decode concurrently.
//  * The output is linear (the whole stream resident in HBM), so a copy is `out[pos + i] =
//    out[pos - d + i mod d]` for all i at once, 64 bytes per step, overlapping or not.
//  * Control is wave-uniform (bit reader, block switching, command loop: scalar work); lanes are used
//    where the format offers width: a symbol is found by ONE ballot — lanes 1..15 each compare the next
//    15 bits (bit-reversed, so that the first bit read is the most significant) with the left-justified
//    limit of their code length — code lengths are ranked into canonical order with ballots, context
//    maps undo move-to-front on a 256-entry list spread over the lanes, copies / raw blocks / dictionary
//    words with their transforms are written a lane per byte.
//
// Prefix codes in memory: 16 header dwords (dword L: limit[L] | offs[L] << 16; dword 0: bit 31 + symbol
// for a code of one symbol) followed by the symbols sorted by (length, value) as u16.
#ifndef BROTLI_AMD_CSRC_K_DECODE_H_
#define BROTLI_AMD_CSRC_K_DECODE_H_

#include "device_common.h"

#define DEC_FLAG_HEADER 1u     // the piece starts with the stream header (WBITS)
#define DEC_FLAG_ISOLATED 2u   // a shard of a plan: copies must not reach before the piece

// error codes (DecResult.error)
#define DEC_OK 0u
#define DEC_ERR_HEADER 1u
#define DEC_ERR_PREFIX_CODE 2u
#define DEC_ERR_CONTEXT_MAP 3u
#define DEC_ERR_DISTANCE 4u
#define DEC_ERR_DICTIONARY 5u
#define DEC_ERR_OVERRUN 6u       // a meta-block announces more output than the piece's capacity has room for (a bigger buffer helps)
#define DEC_ERR_INPUT 7u         // ran past the end of the input
#define DEC_ERR_ARENA 8u         // more prefix codes than the arena of the piece holds
#define DEC_ERR_UNSUPPORTED 9u   // large window
#define DEC_ERR_BLOCK_LENGTH 10u // a command writes past the length its meta-block announced (damaged stream; no buffer helps)

struct DecPiece {
  uint64_t in_off, in_len;     // compressed bytes of the piece in the job input
  uint64_t out_off;            // where its bytes go in the job output == its position in the stream
  uint64_t out_cap;            // room there
  uint32_t flags;
  uint32_t lgwin;              // window of the stream (pieces without DEC_FLAG_HEADER)
};
struct DecResult {
  uint64_t out_bytes;
  uint64_t in_bits;            // bits consumed
  uint32_t error;
  uint32_t finished;           // the ISLAST meta-block was decoded
  uint32_t lgwin;
  uint32_t metablocks;
};
struct DecTransform { uint16_t prefix_off; uint8_t prefix_len, op; uint16_t suffix_off; uint8_t suffix_len, param; };
struct DecArgs {
  const DecPiece* pieces;
  DecResult* results;
  const DeviceTables* T;             // dictionary, context LUT
  const DecTransform* transforms;    // 121 records
  const uint8_t* transform_text;
  const uint8_t* input;
  uint8_t* out;
  uint32_t* arena;                   // npieces * arena_words
  uint32_t arena_words;
  uint32_t npieces;
  uint32_t flags;                    // DEC_ARG_*
  uint32_t pad;
};
#define DEC_ARG_NO_LDS_CACHE 1u      // measurement only: every table read goes to the arena in HBM / L2

// LDS of one wave (dwords)
#define DEC_LDS_LENS 0u            // u8 lens[768]
#define DEC_LDS_CLTREE 192u        // code-length code: 16 + 9
// what the inner loops touch once per symbol, kept next to the wave: the context LUT of the current
// literal block type's mode, its row of the context map, and the headers of the first prefix codes
#define DEC_LDS_LUT 224u           // u8 [512]
#define DEC_LDS_CMAP 352u          // u8 [64]
#define DEC_LDS_HDR_L 368u         // 16 codes x 16 dwords
#define DEC_LDS_HDR_D 624u         // 16 x 16
#define DEC_LDS_HDR_I 880u         // 4 x 16
#define DEC_LDS_WORDS 944u
#define DEC_HDR_CACHE_L 16u
#define DEC_HDR_CACHE_D 16u
#define DEC_HDR_CACHE_I 4u

// arena layout of one piece (dwords)
#define DEC_A_CMAP_L 0u            // u8 [64 * 256]
#define DEC_A_CMAP_D 4096u         // u8 [4 * 256]
#define DEC_A_MODES 4352u          // u8 [256]
#define DEC_A_BLOCK_TREES 4416u    // 3 x (types: 16 + 129, counts: 16 + 13)
#define DEC_BT_STRIDE 176u
#define DEC_A_TREES (4416u + 3u * DEC_BT_STRIDE)

// ---- bit reader (LSB first), wave-uniform ---------------------------------------------------
struct BitRd {
  const uint8_t* p;        // next byte to load
  uint64_t acc;
  uint32_t n;              // valid bits in acc
  const uint8_t* base;
  uint64_t ahead;          // the eight bytes at p, requested when p was set: a refill never waits for memory
};
DEV void br_init(BitRd& b, const uint8_t* in) { b.p = b.base = in; b.acc = 0; b.n = 0; b.ahead = ld64(in); }
DEV void br_fill(BitRd& b) {            // n >= 56 afterwards
  b.acc |= b.ahead << b.n;
  b.p += (63u - b.n) >> 3;
  b.n |= 56u;
  b.ahead = ld64(b.p);
}
DEV uint32_t br_read(BitRd& b, uint32_t k) {   // k <= 32
  if (b.n < k) br_fill(b);
  const uint32_t v = (uint32_t)(b.acc & ((1ull << k) - 1ull));
  b.acc >>= k;
  b.n -= k;
  return v;
}
DEV uint64_t br_bitpos(const BitRd& b) { return (uint64_t)(b.p - b.base) * 8u - b.n; }
// The reader has loaded past the end of the input (the buffer has BROTLI_AMD_INPUT_SLACK readable bytes
// behind it): checked inside every loop whose length the stream dictates, so that a damaged stream ends
// as DEC_ERR_INPUT instead of walking through memory.
DEV bool br_overrun(const BitRd& b, uint64_t in_len) { return (uint64_t)(b.p - b.base) > in_len + 8u; }
// Drops the rest of the current byte and returns it: padding bits, which the format wants zero.
DEV uint32_t br_align(BitRd& b) {
  const uint32_t r = b.n & 7u;
  const uint32_t v = (uint32_t)b.acc & ((1u << r) - 1u);
  b.acc >>= r;
  b.n -= r;
  return v;
}
// byte position after alignment; re-seats the reader there
DEV void br_seek(BitRd& b, uint64_t byte_pos) { b.p = b.base + byte_pos; b.acc = 0; b.n = 0; b.ahead = ld64(b.p); }

// ---- prefix codes -------------------------------------------------------------------------------
// `hdr`: the 16 header dwords (LDS copy or the arena), `sorted`: the symbols in the arena.
DEV uint32_t dec_symbol_at(BitRd& b, const uint32_t* hdr, const uint16_t* sorted);
DEV uint32_t dec_symbol(BitRd& b, const uint32_t* tree) { return dec_symbol_at(b, tree, (const uint16_t*)(tree + 16)); }
DEV uint32_t dec_symbol_at(BitRd& b, const uint32_t* hdr, const uint16_t* sorted) {
  const int lane = wave_lane();
  if (b.n < 15u) br_fill(b);
  const uint32_t h = hdr[lane & 15];
  const uint32_t h0 = wave_bcast(h, 0);
  if (h0 & 0x80000000u) return h0 & 0xFFFFu;
  const uint32_t v = dev_bitrev32((uint32_t)b.acc) >> 17;
  const uint64_t m = wave_ballot(lane >= 1 && lane <= 15 && v < (h & 0xFFFFu));
  const int L = m ? dev_ctz64(m) : 15;
  const uint32_t hl = wave_bcast(h, L);
  const uint32_t idx = ((hl >> 16) + (v >> (15 - L))) & 0xFFFFu;
  b.acc >>= L;
  b.n -= (uint32_t)L;
  return sorted[idx];
}

// Canonical code from lens[0 .. n) (u8 in LDS): header + sorted symbols.  Returns false if the
// lengths do not form a complete code.
DEV bool dec_build_tree(const uint8_t* lens, uint32_t n, uint32_t* tree) {
  const int lane = wave_lane();
  const uint64_t below = (1ull << lane) - 1ull;
  uint32_t count[16];
#pragma unroll
  for (int L = 0; L < 16; ++L) count[L] = 0;
  for (uint32_t s0 = 0; s0 < n; s0 += 64u) {
    const uint32_t s = s0 + (uint32_t)lane;
    const uint32_t len = s < n ? lens[s] : 0u;
#pragma unroll
    for (int L = 1; L < 16; ++L) count[L] += (uint32_t)dev_popc64(wave_ballot(len == (uint32_t)L));
  }
  uint32_t limit = 0, first = 0, nonzero = 0;
  uint32_t start[16];
  start[0] = 0;
#pragma unroll
  for (int L = 1; L < 16; ++L) {
    const uint32_t base = limit;
    limit += count[L] << (15 - L);
    start[L] = first;
    if (lane == L) tree[L] = (limit & 0xFFFFu) | (((first - (base >> (15 - L))) & 0xFFFFu) << 16);
    first += count[L];
    nonzero += count[L];
  }
  if (lane == 0) tree[0] = 0;
  if (limit != 32768u || nonzero < 2u) return false;
  uint16_t* sorted = (uint16_t*)(tree + 16);
  for (uint32_t s0 = 0; s0 < n; s0 += 64u) {
    const uint32_t s = s0 + (uint32_t)lane;
    const uint32_t len = s < n ? lens[s] : 0u;
#pragma unroll
    for (int L = 1; L < 16; ++L) {
      const uint64_t m = wave_ballot(len == (uint32_t)L);
      if (len == (uint32_t)L) sorted[start[L] + (uint32_t)dev_popc64(m & below)] = (uint16_t)s;
      start[L] += (uint32_t)dev_popc64(m);
    }
  }
  wave_sync();
  return true;
}

// RFC 7932 section 3.4 / 3.5: reads one prefix code over `alphabet` symbols into `tree`.
DEV bool dec_read_tree(BitRd& b, uint32_t alphabet, uint32_t* tree, uint32_t* lds) {
  const int lane = wave_lane();
  uint8_t* lens = (uint8_t*)(lds + DEC_LDS_LENS);
  uint32_t* cl_tree = lds + DEC_LDS_CLTREE;
  const uint32_t hskip = br_read(b, 2);
  for (uint32_t s = (uint32_t)lane; s < alphabet; s += 64u) lens[s] = 0;
  wave_sync();
  if (hskip == 1u) {                                    // simple code: 1 .. 4 symbols
    const uint32_t nsym = br_read(b, 2) + 1u;
    const uint32_t bits = alphabet > 1u ? log2floor(alphabet - 1u) + 1u : 0u;
    uint32_t sym[4] = {0, 0, 0, 0};
    for (uint32_t i = 0; i < nsym; ++i) {
      sym[i] = br_read(b, bits);
      if (sym[i] >= alphabet) return false;
      for (uint32_t j = 0; j < i; ++j) if (sym[j] == sym[i]) return false;
    }
    if (nsym == 1u) {
      if (lane == 0) tree[0] = 0x80000000u | sym[0];
      wave_sync();
      return true;
    }
    uint32_t l0 = 1, l1 = 1, l2 = 0, l3 = 0;
    if (nsym == 3u) { l1 = 2; l2 = 2; }
    if (nsym == 4u) {
      if (br_read(b, 1)) { l0 = 1; l1 = 2; l2 = 3; l3 = 3; } else { l0 = l1 = l2 = l3 = 2; }
    }
    if (lane == 0) {
      lens[sym[0]] = (uint8_t)l0;
      lens[sym[1]] = (uint8_t)l1;
      if (nsym > 2u) lens[sym[2]] = (uint8_t)l2;
      if (nsym > 3u) lens[sym[3]] = (uint8_t)l3;
    }
    wave_sync();
    return dec_build_tree(lens, alphabet, tree);
  }
  // complex code: the code-length code first (18 symbols in a fixed order, a fixed variable-length code)
  uint8_t* cl_lens = lens + 736;          // 18 bytes behind the largest alphabet (704)
  if (lane < 18) cl_lens[lane] = 0;
  wave_sync();
  {
    int space = 32;
    uint32_t num = 0, last = 0;
    for (uint32_t i = hskip; i < 18u && space > 0; ++i) {
      // order: 1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15
      const uint32_t idx = i < 4u ? i + 1u : i == 4u ? 0u : i == 5u ? 5u : i == 6u ? 17u : i == 7u ? 6u :
                           i == 8u ? 16u : i - 2u;
      if (b.n < 4u) br_fill(b);
      const uint32_t x = (uint32_t)b.acc & 15u;
      uint32_t v, nb;
      if ((x & 3u) == 0u) { v = 0; nb = 2; }
      else if ((x & 3u) == 2u) { v = 3; nb = 2; }
      else if ((x & 3u) == 1u) { v = 4; nb = 2; }
      else if ((x & 7u) == 3u) { v = 2; nb = 3; }
      else if (x == 7u) { v = 1; nb = 4; }
      else { v = 5; nb = 4; }
      b.acc >>= nb;
      b.n -= nb;
      if (v != 0u) {
        if (lane == 0) cl_lens[idx] = (uint8_t)v;
        space -= (int)(32u >> v);
        ++num;
        last = idx;
      }
    }
    wave_sync();
    if (num == 0u) return false;
    if (num == 1u) {
      if (lane == 0) cl_tree[0] = 0x80000000u | last;
      wave_sync();
    } else {
      if (space != 0) return false;
      // lengths 1..5 scale to the 15-bit space of dec_build_tree unchanged
      if (!dec_build_tree(cl_lens, 18u, cl_tree)) return false;
    }
  }
  uint32_t symbol = 0, prev_len = 8, repeat = 0, repeat_len = 0;
  int space = 32768;
  while (symbol < alphabet && space > 0) {
    const uint32_t c = dec_symbol(b, cl_tree);
    if (c < 16u) {
      repeat = 0;
      if (lane == 0) lens[symbol] = (uint8_t)c;
      ++symbol;
      if (c != 0u) { prev_len = c; space -= (int)(32768u >> c); }
    } else {
      const uint32_t extra = c == 16u ? 2u : 3u;
      const uint32_t new_len = c == 16u ? prev_len : 0u;
      if (repeat_len != new_len) { repeat = 0; repeat_len = new_len; }
      const uint32_t old = repeat;
      if (repeat > 0u) repeat = (repeat - 2u) << extra;
      repeat += br_read(b, extra) + 3u;
      const uint32_t delta = repeat - old;
      if (symbol + delta > alphabet) return false;
      for (uint32_t k = (uint32_t)lane; k < delta; k += 64u) lens[symbol + k] = (uint8_t)repeat_len;
      symbol += delta;
      if (repeat_len != 0u) space -= (int)(delta << (15u - repeat_len));
    }
  }
  wave_sync();
  if (space != 0) return false;
  return dec_build_tree(lens, alphabet, tree);
}

// NBLTYPES / NTREES: 1 .. 256 (section 9.2)
DEV uint32_t dec_read_count256(BitRd& b) {
  if (!br_read(b, 1)) return 1u;
  const uint32_t n = br_read(b, 3);
  return (1u << n) + 1u + br_read(b, n);
}

// block count codes (section 6.3)
DEV uint32_t dec_read_block_count(BitRd& b, const uint32_t* tree) {
  const uint32_t code = dec_symbol(b, tree);
  // base 1, 5, 9, 13, 17, 25, ...; extra bits 2 x4, 3 x4, 4 x4, 5 x4, 6, 6, 7, 8, 9, 10, 11, 12, 13, 24
  // then 241 / 305 (6 bits), 369 (7), 497 (8), 753 (9), 1265 (10), 2289 (11), 4337 (12), 8433 (13): 241 + 2^nbits,
  // and 16625 (24 bits)
  uint32_t nbits, base;
  if (code < 16u) { nbits = 2u + (code >> 2); base = 1u + (((4u << (code >> 2)) - 4u) << 2) + ((code & 3u) << nbits); }
  else if (code < 18u) { nbits = 6; base = 241u + ((code - 16u) << 6); }
  else if (code < 25u) { nbits = code - 11u; base = 241u + (1u << nbits); }
  else { nbits = 24; base = 16625u; }
  return base + br_read(b, nbits);
}

struct DecBlocks {           // one block category (section 6)
  uint32_t ntypes, type, prev_type, left;
};
DEV void dec_switch_block(BitRd& b, DecBlocks& k, const uint32_t* trees) {
  const uint32_t code = dec_symbol(b, trees);
  uint32_t t = code == 0u ? k.prev_type : code == 1u ? k.type + 1u : code - 2u;
  if (t >= k.ntypes) t -= k.ntypes;
  k.prev_type = k.type;
  k.type = t;
  k.left = dec_read_block_count(b, trees + 16u + 129u);
}

// Context map (section 7.3): `size` entries of tree indices below `ntrees`, into map[].
DEV bool dec_read_context_map(BitRd& b, uint64_t in_len, uint32_t size, uint32_t ntrees, uint8_t* map,
                              uint32_t* scratch_tree, uint32_t* lds) {
  const int lane = wave_lane();
  const uint32_t rlemax = br_read(b, 1) ? br_read(b, 4) + 1u : 0u;
  if (!dec_read_tree(b, ntrees + rlemax, scratch_tree, lds)) return false;
  uint32_t i = 0;
  while (i < size) {
    if (br_overrun(b, in_len)) return false;
    const uint32_t s = dec_symbol(b, scratch_tree);
    if (s == 0u) {
      if (lane == 0) map[i] = 0;
      ++i;
    } else if (s <= rlemax) {
      const uint32_t reps = (1u << s) + br_read(b, s);
      if (i + reps > size) return false;
      for (uint32_t k = (uint32_t)lane; k < reps; k += 64u) map[i + k] = 0;
      i += reps;
    } else {
      if (lane == 0) map[i] = (uint8_t)(s - rlemax);
      ++i;
    }
  }
  wave_sync();
  if (br_read(b, 1)) {
    // inverse move-to-front over the list 0 .. 255: entry p lives in register p >> 6 of lane p & 63
    uint32_t l0 = (uint32_t)lane, l1 = 64u + (uint32_t)lane, l2 = 128u + (uint32_t)lane, l3 = 192u + (uint32_t)lane;
    for (uint32_t k = 0; k < size; ++k) {
      const uint32_t idx = map[k];
      const uint32_t row = idx >> 6, col = idx & 63u;
      const uint32_t src = row == 0u ? l0 : row == 1u ? l1 : row == 2u ? l2 : l3;
      const uint32_t value = wave_bcast(src, (int)col);
      if (lane == 0) map[k] = (uint8_t)value;
      if (idx != 0u) {
        // positions 1 .. idx take the entry before them, position 0 takes `value`
        const uint32_t p0 = wave_shfl(l0, (lane - 1) & 63), p1 = wave_shfl(l1, (lane - 1) & 63);
        const uint32_t p2 = wave_shfl(l2, (lane - 1) & 63), p3 = wave_shfl(l3, (lane - 1) & 63);
        const uint32_t e0 = wave_bcast(l0, 63), e1 = wave_bcast(l1, 63), e2 = wave_bcast(l2, 63);
        const uint32_t n0 = lane == 0 ? value : p0;
        const uint32_t n1 = lane == 0 ? e0 : p1;
        const uint32_t n2 = lane == 0 ? e1 : p2;
        const uint32_t n3 = lane == 0 ? e2 : p3;
        if ((uint32_t)lane <= idx) l0 = n0;
        if (64u + (uint32_t)lane <= idx) l1 = n1;
        if (128u + (uint32_t)lane <= idx) l2 = n2;
        if (192u + (uint32_t)lane <= idx) l3 = n3;
      }
    }
    wave_sync();
  }
  return true;
}

// ---- dictionary words (section 8, Appendix A / B) -------------------------------------------------
// Writes transform `t` of the `wlen`-byte word at `word` to dst; returns the bytes written (< 64).
DEV uint32_t dec_dictionary_word(const DecArgs& a, const uint8_t* word, uint32_t wlen, uint32_t t, uint8_t* dst,
                                 uint32_t room) {
  const int lane = wave_lane();
  const DecTransform tr = a.transforms[t];
  uint32_t skip = 0, keep = wlen;
  if (tr.op >= 1 && tr.op <= 9) keep = wlen > tr.op ? wlen - tr.op : 0u;
  if (tr.op >= 12 && tr.op <= 20) { skip = umin((uint32_t)tr.op - 11u, wlen); keep = wlen - skip; }
  const uint32_t total = tr.prefix_len + keep + tr.suffix_len;
  if (total > room) return 0xFFFFFFFFu;
  uint32_t w = 0;
  if ((uint32_t)lane < keep) w = word[skip + (uint32_t)lane];
  if (tr.op == 10 || tr.op == 11) {
    // ToUpperCase over UTF-8 sequences: one byte (a..z: ^32), two bytes (second ^32), three (third ^5)
    uint32_t x32 = 0, x5 = 0, k = 0;
    while (k < keep) {
      const uint32_t lead = wave_bcast(w, (int)k);
      if (lead < 0xC0u) { if (lead >= 97u && lead <= 122u) x32 |= 1u << k; k += 1u; }
      else if (lead < 0xE0u) { x32 |= 1u << (k + 1u); k += 2u; }
      else { x5 |= 1u << (k + 2u); k += 3u; }
      if (tr.op == 10) break;
    }
    if ((x32 >> lane) & 1u) w ^= 32u;
    if ((x5 >> lane) & 1u) w ^= 5u;
  }
  // a lane per output byte
  uint32_t v = 0;
  const uint32_t j = (uint32_t)lane;
  const uint32_t from_word = wave_shfl(w, (int)((j - tr.prefix_len) & 63u));
  if (j < tr.prefix_len) v = a.transform_text[tr.prefix_off + j];
  else if (j < tr.prefix_len + keep) v = from_word;
  else if (j < total) v = a.transform_text[tr.suffix_off + (j - tr.prefix_len - keep)];
  if (j < total) dst[j] = (uint8_t)v;
  return total;
}

// ---- one piece ---------------------------------------------------------------------------------
DEV void decode_piece(const DecArgs& a, uint32_t piece, uint32_t* lds) {
  const int lane = wave_lane();
  const DecPiece P = a.pieces[piece];
  const DeviceTables* T = a.T;
  uint32_t* arena = a.arena + (size_t)piece * a.arena_words;
  uint8_t* cmap_l = (uint8_t*)(arena + DEC_A_CMAP_L);
  uint8_t* cmap_d = (uint8_t*)(arena + DEC_A_CMAP_D);
  uint8_t* modes = (uint8_t*)(arena + DEC_A_MODES);
  uint8_t* out = a.out + P.out_off;
  const uint8_t* lut_all = T->context_lut - 1024;      // the four context modes, 512 bytes each (section 7.1)
  BitRd b;
  br_init(b, a.input + P.in_off);
  uint32_t error = DEC_OK, finished = 0, metablocks = 0;
  uint32_t lgwin = P.lgwin;
  uint64_t pos = 0;                                    // bytes of the piece written so far
  int32_t ring[4] = {4, 11, 15, 16};                   // last distances, newest first
  if (P.flags & DEC_FLAG_HEADER) {                     // WBITS, section 9.1
    if (!br_read(b, 1)) lgwin = 16;
    else {
      const uint32_t n = br_read(b, 3);
      if (n != 0u) lgwin = 17u + n;
      else {
        const uint32_t m = br_read(b, 3);
        if (m == 0u) lgwin = 17;
        else if (m == 1u) error = DEC_ERR_UNSUPPORTED;   // large window
        else lgwin = 8u + m;
      }
    }
  }
  const uint32_t max_backward = (1u << lgwin) - 16u;

  while (!error && !finished) {
    // a piece that is not the end of the stream stops at the meta-block boundary where its input ends
    // (every shard of a plan ends byte aligned, behind the padding block of its flush)
    if (br_bitpos(b) == P.in_len * 8u && metablocks != 0u) break;
    if (br_bitpos(b) >= P.in_len * 8u) { error = DEC_ERR_INPUT; break; }
    ++metablocks;
    const uint32_t is_last = br_read(b, 1);
    if (is_last && br_read(b, 1)) { finished = 1; break; }            // ISLASTEMPTY
    const uint32_t mnib = br_read(b, 2);
    if (mnib == 3u) {                                                 // metadata: skipped
      if (br_read(b, 1)) { error = DEC_ERR_HEADER; break; }
      const uint32_t nbytes = br_read(b, 2);
      uint32_t skip = 0;
      if (nbytes) {
        skip = br_read(b, 8u * nbytes);
        if (nbytes > 1u && (skip >> (8u * (nbytes - 1u))) == 0u) { error = DEC_ERR_HEADER; break; }
        skip += 1u;
      }
      if (br_align(b) != 0u) { error = DEC_ERR_HEADER; break; }
      if ((br_bitpos(b) >> 3) + skip > P.in_len) { error = DEC_ERR_INPUT; break; }   // (never seek outside the input)
      br_seek(b, (br_bitpos(b) >> 3) + skip);
      if (is_last) finished = 1;
      continue;
    }
    const uint32_t nn = 4u + mnib;
    uint32_t mlen = br_read(b, 16);
    if (nn > 4u) mlen |= br_read(b, 4u * (nn - 4u)) << 16;
    if (nn > 4u && (mlen >> (4u * (nn - 1u))) == 0u) { error = DEC_ERR_HEADER; break; }
    mlen += 1u;
    if (pos + mlen > P.out_cap) { error = DEC_ERR_OVERRUN; break; }
    if (!is_last && br_read(b, 1)) {                                  // uncompressed
      if (br_align(b) != 0u) { error = DEC_ERR_HEADER; break; }
      const uint64_t from = br_bitpos(b) >> 3;
      if (from + mlen > P.in_len) { error = DEC_ERR_INPUT; break; }
      const uint8_t* src = b.base + from;
      for (uint32_t i = (uint32_t)lane; i < mlen; i += 64u) out[pos + i] = src[i];
      pos += mlen;
      br_seek(b, from + mlen);
      wave_sync();
      continue;
    }
    // ---- compressed meta-block header (section 9.2) ----
    DecBlocks blk[3];
    bool ok = true;
#pragma unroll                  // (blk[] indexed by constants only: it stays in registers — see k_store.h, sel3)
    for (int c = 0; c < 3; ++c) {
      if (!ok) break;
      uint32_t* bt = arena + DEC_A_BLOCK_TREES + (uint32_t)c * DEC_BT_STRIDE;
      blk[c].ntypes = dec_read_count256(b);
      blk[c].type = 0;
      blk[c].prev_type = 1;
      blk[c].left = 1u << 28;
      if (blk[c].ntypes >= 2u) {
        ok = dec_read_tree(b, blk[c].ntypes + 2u, bt, lds) && dec_read_tree(b, 26u, bt + 16u + 129u, lds);
        if (ok) blk[c].left = dec_read_block_count(b, bt + 16u + 129u);
      }
    }
    if (!ok) { error = DEC_ERR_PREFIX_CODE; break; }
    const uint32_t npostfix = br_read(b, 2);
    const uint32_t ndirect = br_read(b, 4) << npostfix;
    for (uint32_t t = 0; t < blk[0].ntypes; ++t) {
      const uint32_t m = br_read(b, 2);
      if (lane == 0) modes[t] = (uint8_t)m;
    }
    const uint32_t alphabet_d = 16u + ndirect + (48u << npostfix);
    const uint32_t stride_l = 16u + 128u, stride_i = 16u + 352u, stride_d = 16u + ((alphabet_d + 1u) >> 1);
    const uint32_t ntrees_l = dec_read_count256(b);
    uint32_t* scratch_tree = arena + DEC_A_TREES;     // the context maps' own codes: read before the trees land here
    if (ntrees_l >= 2u) {
      if (!dec_read_context_map(b, P.in_len, 64u * blk[0].ntypes, ntrees_l, cmap_l, scratch_tree, lds)) { error = DEC_ERR_CONTEXT_MAP; break; }
    } else {
      for (uint32_t i = (uint32_t)lane; i < 64u * blk[0].ntypes; i += 64u) cmap_l[i] = 0;
    }
    const uint32_t ntrees_d = dec_read_count256(b);
    if (ntrees_d >= 2u) {
      if (!dec_read_context_map(b, P.in_len, 4u * blk[2].ntypes, ntrees_d, cmap_d, scratch_tree, lds)) { error = DEC_ERR_CONTEXT_MAP; break; }
    } else {
      for (uint32_t i = (uint32_t)lane; i < 4u * blk[2].ntypes; i += 64u) cmap_d[i] = 0;
    }
    wave_sync();
    uint32_t* trees_l = arena + DEC_A_TREES;
    uint32_t* trees_i = trees_l + ntrees_l * stride_l;
    uint32_t* trees_d = trees_i + blk[1].ntypes * stride_i;
    if (DEC_A_TREES + ntrees_l * stride_l + blk[1].ntypes * stride_i + ntrees_d * stride_d > a.arena_words) {
      error = DEC_ERR_ARENA;
      break;
    }
    // (one prefix code is at most ~1.6 KB of input: BROTLI_AMD_DECODE_SLACK covers the last one of a damaged stream)
    for (uint32_t t = 0; t < ntrees_l && ok; ++t) ok = !br_overrun(b, P.in_len) && dec_read_tree(b, 256u, trees_l + t * stride_l, lds);
    for (uint32_t t = 0; t < blk[1].ntypes && ok; ++t) ok = !br_overrun(b, P.in_len) && dec_read_tree(b, 704u, trees_i + t * stride_i, lds);
    for (uint32_t t = 0; t < ntrees_d && ok; ++t) ok = !br_overrun(b, P.in_len) && dec_read_tree(b, alphabet_d, trees_d + t * stride_d, lds);
    if (!ok) { error = DEC_ERR_PREFIX_CODE; break; }
    // every cmap entry must name a tree that exists
    {
      bool bad = false;
      for (uint32_t i = (uint32_t)lane; i < 64u * blk[0].ntypes; i += 64u) bad |= cmap_l[i] >= ntrees_l;
      for (uint32_t i = (uint32_t)lane; i < 4u * blk[2].ntypes; i += 64u) bad |= cmap_d[i] >= ntrees_d;
      if (wave_ballot(bad)) { error = DEC_ERR_CONTEXT_MAP; break; }
    }

    // the per-symbol tables of this meta-block, next to the wave
    const bool cache = !(a.flags & DEC_ARG_NO_LDS_CACHE);
    uint8_t* lds_lut = (uint8_t*)(lds + DEC_LDS_LUT);
    uint8_t* lds_cmap = (uint8_t*)(lds + DEC_LDS_CMAP);
    uint32_t cached_mode = 0xFFFFFFFFu;
    if (cache) {
      for (uint32_t i = (uint32_t)lane; i < umin(ntrees_l, DEC_HDR_CACHE_L) * 16u; i += 64u)
        lds[DEC_LDS_HDR_L + i] = trees_l[(i >> 4) * stride_l + (i & 15u)];
      for (uint32_t i = (uint32_t)lane; i < umin(ntrees_d, DEC_HDR_CACHE_D) * 16u; i += 64u)
        lds[DEC_LDS_HDR_D + i] = trees_d[(i >> 4) * stride_d + (i & 15u)];
      for (uint32_t i = (uint32_t)lane; i < umin(blk[1].ntypes, DEC_HDR_CACHE_I) * 16u; i += 64u)
        lds[DEC_LDS_HDR_I + i] = trees_i[(i >> 4) * stride_i + (i & 15u)];
    }
    // (re)loads what depends on the literal block type: the LUT of its context mode, its context-map row
    auto literal_block_tables = [&]() {
      if (!cache) return;
      const uint32_t mode = modes[blk[0].type];
      if (mode != cached_mode) {
        const uint32_t* src = (const uint32_t*)(lut_all + (mode << 9));
        lds[DEC_LDS_LUT + (uint32_t)lane] = src[lane];
        lds[DEC_LDS_LUT + 64u + (uint32_t)lane] = src[64 + lane];
        cached_mode = mode;
      }
      if (lane < 16) lds[DEC_LDS_CMAP + (uint32_t)lane] = ((const uint32_t*)(cmap_l + (blk[0].type << 6)))[lane];
      wave_sync();
    };
    literal_block_tables();
    wave_sync();

    // ---- commands (section 5, 10) ----
    const uint64_t mb_end = pos + mlen;
    uint32_t p1 = 0, p2 = 0;                    // the two bytes before pos
    {
      const uint64_t sp = P.out_off + pos;      // stream position
      if (sp >= 1u) p1 = (pos >= 1u || !(P.flags & DEC_FLAG_ISOLATED)) ? a.out[sp - 1u] : 0u;
      if (sp >= 2u) p2 = (pos >= 2u || !(P.flags & DEC_FLAG_ISOLATED)) ? a.out[sp - 2u] : 0u;
    }
    const uint32_t* bt_l = arena + DEC_A_BLOCK_TREES;
    const uint32_t* bt_i = bt_l + DEC_BT_STRIDE;
    const uint32_t* bt_d = bt_i + DEC_BT_STRIDE;
    while (pos < mb_end && !error) {
      if (br_overrun(b, P.in_len)) { error = DEC_ERR_INPUT; break; }
      if (blk[1].left == 0u) dec_switch_block(b, blk[1], bt_i);
      --blk[1].left;
      const uint32_t* tree_i = trees_i + blk[1].type * stride_i;
      const uint32_t cmd = dec_symbol_at(b, cache && blk[1].type < DEC_HDR_CACHE_I ? lds + DEC_LDS_HDR_I + blk[1].type * 16u : tree_i,
                                         (const uint16_t*)(tree_i + 16));
      // insert-and-copy code -> insert code, copy code (section 5): cells of 64 symbols
      const uint32_t cell = cmd >> 6;
      uint32_t icode, ccode;
      {
        // cell:        0  1  2  3  4  5  6   7   8   9   10
        // insert base: 0  0  0  0  8  8  0   16  8   16  16
        // copy base:   0  8  0  8  0  8  16  0   16  8   16
        const uint32_t ib = (uint32_t)((0x22120110000ull >> (cell * 4u)) & 0xFull) * 8u;
        const uint32_t cb = (uint32_t)((0x21202101010ull >> (cell * 4u)) & 0xFull) * 8u;
        icode = ib + ((cmd >> 3) & 7u);
        ccode = cb + (cmd & 7u);
      }
      uint32_t insert_len, copy_len;
      {
        // insert length: codes 0..5 literal, then pairs with 1, 2, 3, 4, 5 extra bits, then 6, 7, 8, 9, 10, 12, 14, 24
        uint32_t nb, base;
        if (icode < 6u) { nb = 0; base = icode; }
        else if (icode < 16u) { nb = (icode - 4u) >> 1; base = ((2u + (icode & 1u)) << nb) + 2u; }
        else if (icode == 16u) { nb = 6; base = 130; }
        else if (icode < 21u) { nb = icode - 10u; base = (1u << nb) + 66u; }
        else if (icode == 21u) { nb = 12; base = 2114; }
        else if (icode == 22u) { nb = 14; base = 6210; }
        else { nb = 24; base = 22594; }
        insert_len = base + (nb ? br_read(b, nb) : 0u);
        if (ccode < 8u) { nb = 0; base = ccode + 2u; }
        else if (ccode < 18u) { nb = (ccode - 6u) >> 1; base = ((2u + (ccode & 1u)) << nb) + 6u; }
        else if (ccode < 23u) { nb = ccode - 12u; base = (1u << nb) + 70u; }
        else { nb = 24; base = 2118; }
        copy_len = base + (nb ? br_read(b, nb) : 0u);
      }
      if (pos + insert_len > mb_end) { error = DEC_ERR_BLOCK_LENGTH; break; }
      // literals
      for (uint32_t k = 0; k < insert_len; ++k) {
        if (br_overrun(b, P.in_len)) { error = DEC_ERR_INPUT; break; }
        if (blk[0].left == 0u) { dec_switch_block(b, blk[0], bt_l); literal_block_tables(); }
        --blk[0].left;
        uint32_t tree;
        if (cache) {
          tree = lds_cmap[lds_lut[p1] | lds_lut[256u + p2]];
        } else {
          const uint8_t* lut = lut_all + ((uint32_t)modes[blk[0].type] << 9);
          tree = cmap_l[(blk[0].type << 6) + (lut[p1] | lut[256u + p2])];
        }
        const uint32_t* tree_l = trees_l + tree * stride_l;
        // (the symbols themselves in LDS as well — 256 B per code — was measured: no gain, the loop is bound by
        // instruction issue, not by this load; profiles/r02_x_*)
        const uint32_t lit = dec_symbol_at(b, cache && tree < DEC_HDR_CACHE_L ? lds + DEC_LDS_HDR_L + tree * 16u : tree_l,
                                           (const uint16_t*)(tree_l + 16));
        if (lane == 0) out[pos] = (uint8_t)lit;
        ++pos;
        p2 = p1;
        p1 = lit;
      }
      if (error || pos >= mb_end) break;
      // distance
      uint32_t distance;
      uint32_t dcode = 0;
      if (cell >= 2u) {
        if (blk[2].left == 0u) dec_switch_block(b, blk[2], bt_d);
        --blk[2].left;
        const uint32_t dctx = copy_len > 4u ? 3u : copy_len - 2u;
        const uint32_t tree = cmap_d[(blk[2].type << 2) + dctx];
        const uint32_t* tree_d = trees_d + tree * stride_d;
        dcode = dec_symbol_at(b, cache && tree < DEC_HDR_CACHE_D ? lds + DEC_LDS_HDR_D + tree * 16u : tree_d,
                              (const uint16_t*)(tree_d + 16));
      }
      if (dcode < 16u) {
        // 0..3: the ring; 4..9: last -1 +1 -2 +2 -3 +3; 10..15: second last likewise
        int32_t d;
        // (chosen between the four values, not ring[dcode]: a computed index would keep the ring in scratch memory)
        const int32_t r0 = ring[0], r1 = ring[1], r2 = ring[2], r3 = ring[3];
        if (dcode < 4u) d = dcode == 0u ? r0 : dcode == 1u ? r1 : dcode == 2u ? r2 : r3;
        else {
          const uint32_t r = dcode - 4u, which = r >= 6u ? 1u : 0u, q = which ? r - 6u : r;
          const int32_t mag = (int32_t)(q >> 1) + 1;
          d = (which ? r1 : r0) + ((q & 1u) ? mag : -mag);
        }
        if (d <= 0) { error = DEC_ERR_DISTANCE; break; }
        distance = (uint32_t)d;
      } else if (dcode < 16u + ndirect) {
        distance = dcode - 15u;
      } else {
        const uint32_t x = dcode - ndirect - 16u;
        const uint32_t nbits = 1u + (x >> (npostfix + 1u));
        const uint32_t hcode = x >> npostfix, lcode = x & ((1u << npostfix) - 1u);
        const uint32_t offset = ((2u + (hcode & 1u)) << nbits) - 4u;
        distance = ((offset + br_read(b, nbits)) << npostfix) + lcode + ndirect + 1u;
      }
      const uint64_t stream_pos = P.out_off + pos;
      const uint32_t max_distance = stream_pos < max_backward ? (uint32_t)stream_pos : max_backward;
      if (distance <= max_distance) {
        if (dcode != 0u) { ring[3] = ring[2]; ring[2] = ring[1]; ring[1] = ring[0]; ring[0] = (int32_t)distance; }
        if (pos + copy_len > mb_end) { error = DEC_ERR_BLOCK_LENGTH; break; }
        if ((P.flags & DEC_FLAG_ISOLATED) && distance > pos) { error = DEC_ERR_DISTANCE; break; }
        wave_sync();
        uint8_t* dst = a.out + stream_pos;
        const uint8_t* src = dst - distance;
        for (uint32_t i = (uint32_t)lane; i < copy_len; i += 64u) {
          const uint32_t j = distance >= copy_len ? i : i % distance;
          dst[i] = src[j];
        }
        pos += copy_len;
      } else {
        // static dictionary (section 8)
        if (copy_len < 4u || copy_len > 24u) { error = DEC_ERR_DICTIONARY; break; }
        const uint32_t shift = T->dict_size_bits_by_length[copy_len];
        if (shift == 0u) { error = DEC_ERR_DICTIONARY; break; }
        const uint32_t word_id = distance - max_distance - 1u;
        const uint32_t index = word_id & ((1u << shift) - 1u), transform = word_id >> shift;
        if (transform >= 121u) { error = DEC_ERR_DICTIONARY; break; }
        const uint8_t* word = T->dict + T->dict_offsets_by_length[copy_len] + index * copy_len;
        const uint64_t room = mb_end - pos;
        const uint32_t n = dec_dictionary_word(a, word, copy_len, transform, out + pos,
                                               room > 64u ? 64u : (uint32_t)room);
        if (n == 0xFFFFFFFFu) { error = DEC_ERR_BLOCK_LENGTH; break; }
        pos += n;
      }
      wave_sync();
      if (pos >= 1u) p1 = out[pos - 1u];
      if (pos >= 2u) p2 = out[pos - 2u];
    }
    if (error) break;
    if (is_last) finished = 1;
  }
  // Whatever went wrong after the reader had left the input went wrong on the zeroed slack behind it: that is a
  // stream cut short (the caller may have more bytes), not a damaged one.
  if (br_bitpos(b) > P.in_len * 8u) { error = DEC_ERR_INPUT; finished = 0; }
  if (!error && finished) {                      // the last byte is padded with zeros (section 9.1 / 9.3)
    if (b.n < 8u) br_fill(b);
    if (br_align(b) != 0u) error = DEC_ERR_HEADER;
  }
  wave_sync();
  if (lane == 0) {
    DecResult r;
    r.out_bytes = pos;
    r.in_bits = br_bitpos(b);
    r.error = error;
    r.finished = finished;
    r.lgwin = lgwin;
    r.metablocks = metablocks;
    a.results[piece] = r;
  }
}

#endif  // BROTLI_AMD_CSRC_K_DECODE_H_
