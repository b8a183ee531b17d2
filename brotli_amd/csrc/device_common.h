// brotli_amd/csrc/device_common.h — small device helpers shared by the kernels.
#ifndef BROTLI_AMD_CSRC_DEVICE_COMMON_H_
#define BROTLI_AMD_CSRC_DEVICE_COMMON_H_

#include "enc_types.h"
#include "wave.h"

// gfx950 runs with unaligned global access enabled: these become single
// global_load_dword / dwordx2 / dwordx4 instructions at any byte address.
DEV uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
DEV uint64_t ld64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
DEV void st32(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
DEV void st16(uint8_t* p, uint16_t v) { __builtin_memcpy(p, &v, 2); }

DEV uint32_t log2floor(uint32_t n) { return 31u - (uint32_t)dev_clz32(n); }
DEV uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
DEV uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }

// ---- wave scans -------------------------------------------------------------
#if defined(BROTLI_AMD_SIMT_SIM) || defined(WAVE_SCAN_BPERMUTE)
DEV uint32_t wave_incl_scan(uint32_t v) {
  const int lane = wave_lane();
  uint32_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = wave_shfl(x, (lane - d) & 63);
    if (lane >= d) x += y;
  }
  return x;
}
#else
// Inclusive prefix sum over the wave in six VALU instructions with DPP operands: four shifts inside the rows of 16
// lanes (row_shr:1 / 2 / 4 / 8, zeros shifted in), then lane 15 of a row added to the row behind it (row_bcast:15,
// rows 1 and 3) and lane 31 to the upper half (row_bcast:31) — no trip through the LDS crossbar (six dependent
// ds_bpermute cost ~700 cycles of latency; k_store runs ~2000 scans per 128 KiB shard).  tools/scan_probe.hip checks it
// against the shuffle form on the device.
DEV uint32_t wave_incl_scan(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
  return v;
}
#endif

// ---- hashing ---------------------------------------------------------------
// H68: hash_longest_match64_simd_inc.h:26-32 (five bytes, 15-bit key + 8-bit
// tag); H58: hash_longest_match_simd_inc.h:18-24 (four bytes,
// bucket_bits-bit key + 8-bit tag).  tag2: our own 16-bit fingerprint of the
// first four bytes (enc_types.h).
struct KeyTag { uint32_t key, tag, tag2; };

// Bytes a hasher reads per position (HashTypeLength == StoreLookahead).
DEV uint32_t hasher_htl(int hasher_type) { return (hasher_type == 68 || hasher_type == 6 || hasher_type < 5 || hasher_type == 54) ? 8u : 4u; }

DEV KeyTag hash_pos(uint64_t x, int hasher_type, int bucket_bits) {
  KeyTag r;
  uint32_t h;
  if (hasher_type == 68) {
    h = (uint32_t)((x * (0x1FE35A7BD3579BD3ull << 24)) >> (64 - 15 - 8));
  } else if (hasher_type == 58) {
    h = ((uint32_t)x * 0x1E35A7BDu) >> (32 - bucket_bits - 8);
  } else if (hasher_type == 6) {
    // H6, hash_longest_match64_inc.h:23-29: no tag
    h = (uint32_t)((x * (0x1FE35A7BD3579BD3ull << 24)) >> (64 - 15)) << 8;
  } else {
    // H5, hash_longest_match_inc.h: 32-bit hash, no tag
    h = (((uint32_t)x * 0x1E35A7BDu) >> (32 - bucket_bits)) << 8;
  }
  r.key = h >> 8;
  r.tag = h & 0xFF;
  r.tag2 = ((uint32_t)x * 0x9E3779B1u) >> 16;
  return r;
}

// ---- command encoding (c/enc/command.h:31-143, c/enc/prefix.h:23-46) -------
DEV uint32_t insert_length_code(uint32_t n) {
  if (n < 6) return n;
  if (n < 130) { uint32_t nb = log2floor(n - 2) - 1u; return (nb << 1) + ((n - 2) >> nb) + 2; }
  if (n < 2114) return log2floor(n - 66) + 10;
  if (n < 6210) return 21u;
  if (n < 22594) return 22u;
  return 23u;
}
DEV uint32_t copy_length_code(uint32_t n) {
  if (n < 10) return n - 2;
  if (n < 134) { uint32_t nb = log2floor(n - 6) - 1u; return (nb << 1) + ((n - 6) >> nb) + 4; }
  if (n < 2118) return log2floor(n - 70) + 12;
  return 23u;
}
DEV uint32_t combine_length_codes(uint32_t ins, uint32_t cpy, bool use_last) {
  uint32_t bits64 = (cpy & 7u) | ((ins & 7u) << 3u);
  if (use_last && ins < 8u && cpy < 16u) return (cpy < 8u) ? bits64 : (bits64 | 64u);
  uint32_t offset = 2u * ((cpy >> 3u) + 3u * (ins >> 3u));
  offset = (offset << 5u) + 0x40u + ((0x520D40u >> offset) & 0xC0u);
  return offset | bits64;
}
// NPOSTFIX = NDIRECT = 0 (encode.c:616-640 at the supported qualities).
DEV void prefix_encode_distance(uint32_t distance_code, uint32_t* code, uint32_t* extra) {
  if (distance_code < 16) { *code = distance_code; *extra = 0; return; }
  uint32_t dist = 4u + (distance_code - 16u);
  uint32_t bucket = log2floor(dist) - 1u;
  uint32_t prefix = (dist >> bucket) & 1u;
  uint32_t offset = (2u + prefix) << bucket;
  *code = (bucket << 10) | (16u + 2u * (bucket - 1u) + prefix);
  *extra = dist - offset;
}
DEV Command make_command(uint32_t insertlen, uint32_t copylen, int delta, uint32_t distance_code) {
  Command c;
  uint32_t code, extra;
  prefix_encode_distance(distance_code, &code, &extra);
  c.insert_len = insertlen;
  c.copy_len = copylen | (((uint32_t)(uint8_t)(int8_t)delta) << 25);
  c.dist_extra = extra;
  c.dist_prefix = (uint16_t)code;
  c.cmd_prefix = (uint16_t)combine_length_codes(insert_length_code(insertlen),
      copy_length_code((uint32_t)((int)copylen + delta)), (code & 0x3FF) == 0);
  return c;
}
DEV Command make_insert_command(uint32_t insertlen) {
  Command c;
  c.insert_len = insertlen;
  c.copy_len = 4u << 25;
  c.dist_extra = 0;
  c.dist_prefix = 16;
  c.cmd_prefix = (uint16_t)combine_length_codes(insert_length_code(insertlen), copy_length_code(4), false);
  return c;
}
DEV uint32_t cmd_copy_len(const Command& c) { return c.copy_len & 0x1FFFFFF; }
DEV uint32_t cmd_copy_len_code(const Command& c) {
  uint32_t m = c.copy_len >> 25;
  int32_t delta = (int8_t)((uint8_t)(m | ((m & 0x40) << 1)));
  return (uint32_t)((int32_t)(c.copy_len & 0x1FFFFFF) + delta);
}

// ---- serial bit writer (LSB first, c/enc/write_bits.h:33-54) ---------------
// Used by one lane for headers, trees and raw meta-blocks.  `acc` holds the
// bits of the not yet completed 64-bit word that starts at byte `byte_pos`.
struct BitWriter {
  uint8_t* out;
  uint64_t byte_pos;  // multiple of 8 bytes relative to out
  uint64_t acc;
  uint32_t nacc;      // bits valid in acc (< 64)
};
DEV void bw_init(BitWriter& w, uint8_t* out, uint64_t bit_pos, uint32_t carry_bits_value) {
  // bit_pos < 64 expected only for the carried last_bytes_ (encode.c:1188-1190)
  w.out = out;
  w.byte_pos = 0;
  w.acc = carry_bits_value;
  w.nacc = (uint32_t)bit_pos;
}
DEV void bw_put(BitWriter& w, uint32_t nbits, uint64_t bits) {
  const uint32_t space = 64u - w.nacc;  // 1..64
  if (nbits < space) {
    w.acc |= bits << w.nacc;
    w.nacc += nbits;
  } else {
    w.acc |= bits << w.nacc;
    __builtin_memcpy(w.out + w.byte_pos, &w.acc, 8);
    w.byte_pos += 8;
    w.acc = space < 64u ? (bits >> space) : 0;
    w.nacc = nbits - space;
  }
}
DEV uint64_t bw_bitpos(const BitWriter& w) { return w.byte_pos * 8 + w.nacc; }
DEV void bw_align_byte(BitWriter& w) {
  uint32_t pad = (8 - (w.nacc & 7)) & 7;
  if (pad) bw_put(w, pad, 0);
}
// Writes out whole bytes of the accumulator; leaves < 8 bits pending.
DEV void bw_flush_bytes(BitWriter& w) {
  while (w.nacc >= 8) {
    w.out[w.byte_pos] = (uint8_t)w.acc;
    w.byte_pos += 1;
    w.acc >>= 8;
    w.nacc -= 8;
  }
}

#endif  // BROTLI_AMD_CSRC_DEVICE_COMMON_H_
