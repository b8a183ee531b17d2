// brotli_amd/csrc/k_fast.h — quality 1: the two-pass fragment compressor
// (reference: c/enc/compress_fragment_two_pass.c, the fast
// tree writer c/enc/brotli_bit_stream.c:404-573, driven by
// BrotliEncoderCompressStreamFast c/enc/encode.c:1425-1547).
//
// The reference already cuts a quality-1 stream into independent fragments
// (one per 1 << lgwin bytes of a CompressStream call: own zeroed hash table,
// positions relative to the fragment), so the *unpartitioned* stream parallelises:
//
//   k_fast_parse   one wave per fragment, its 128 KiB blocks in order (they share
//                  the table).  The scan for the next match — one table probe per
//                  position, step growing with the distance from the last match —
//                  visits positions that do not depend on what the probes find, so
//                  64 lanes take the next 64 probe positions of the scan at once
//                  and the first hit (ballot) decides how many of them happened.
//   k_fast_store   one wave per block: ShouldCompress, literal / command
//                  histograms, prefix codes, and the command stream written as a
//                  position-independent bit string into the block's scratch.
//   k_fast_sizes / k_fast_scan / k_fast_emit
//                  the only sequential datum left is the bit position a fragment
//                  starts at, and only its residue mod 8 matters (raw meta-blocks
//                  pad to a byte): sizes for the 8 residues, a scan, then every
//                  block shifts its bit string (or its raw bytes) into place.
#ifndef BROTLI_AMD_CSRC_K_FAST_H_
#define BROTLI_AMD_CSRC_K_FAST_H_

#include "device_common.h"
#include "k_build.h"
#include "k_store.h"

#define FAST_INPUT_SLACK 64u   // readable bytes behind the job input (include/brotli_amd_hip.h)

// ---- workspace addressing (regular functions of the input offset) -------------------
DEV uint32_t* fast_cmds(const FastArgs& a, const FastBlock& B) {
  return (uint32_t*)(a.ws + a.cmds_base + 4ull * B.in_off);
}
DEV uint8_t* fast_lits(const FastArgs& a, const FastBlock& B) { return a.ws + a.lits_base + B.in_off; }
DEV uint32_t* fast_lsum(const FastArgs& a, const FastBlock& B, uint32_t bidx) {
  return (uint32_t*)(a.ws + a.lsum_base + 4ull * (B.in_off + bidx));
}
DEV uint8_t* fast_scratch(const FastArgs& a, const FastBlock& B, uint32_t bidx) {
  return a.ws + a.scr_base + ((2ull * B.in_off + 3ull) & ~3ull) + 1024ull * bidx;
}
DEV uint32_t fast_scratch_cap(const FastBlock& B) { return 2u * B.len + 1016u; }

// ---- hashing / matching (compress_fragment_two_pass.c:31-52) ------------------------
template <int MM>
DEV uint32_t fast_hash(uint64_t x, uint32_t shift) {
  return (uint32_t)(((x << ((8 - MM) * 8)) * 0x1E35A7BDull) >> shift);
}
template <int MM>
DEV bool fast_is_match(uint64_t x, uint64_t y) {
  return MM == 4 ? (uint32_t)x == (uint32_t)y : ((x ^ y) << 16) == 0;
}

// Two-pass command words (:106-214): low byte = code of the 128-symbol working
// alphabet, upper 24 bits = extra-bit value.
DEV uint32_t fast_insert_word(uint32_t insertlen) {
  if (insertlen < 6) return insertlen;
  if (insertlen < 130) {
    const uint32_t tail = insertlen - 2, nbits = log2floor(tail) - 1u, prefix = tail >> nbits;
    return ((nbits << 1) + prefix + 2) | ((tail - (prefix << nbits)) << 8);
  }
  if (insertlen < 2114) {
    const uint32_t tail = insertlen - 66, nbits = log2floor(tail);
    return (nbits + 10) | ((tail - (1u << nbits)) << 8);
  }
  if (insertlen < 6210) return 21u | ((insertlen - 2114) << 8);
  if (insertlen < 22594) return 22u | ((insertlen - 6210) << 8);
  return 23u | ((insertlen - 22594) << 8);
}
DEV uint32_t fast_copy_word(uint32_t copylen) {
  if (copylen < 10) return copylen + 38;
  if (copylen < 134) {
    const uint32_t tail = copylen - 6, nbits = log2floor(tail) - 1u, prefix = tail >> nbits;
    return ((nbits << 1) + prefix + 44) | ((tail - (prefix << nbits)) << 8);
  }
  if (copylen < 2118) {
    const uint32_t tail = copylen - 70, nbits = log2floor(tail);
    return (nbits + 52) | ((tail - (1u << nbits)) << 8);
  }
  return 63u | ((copylen - 2118) << 8);
}
// Copy with the previous command's distance: one word, or two (the second is the
// "distance = last" word 64).  Returns the count.
DEV uint32_t fast_copy_last_words(uint32_t copylen, uint32_t* w0) {
  if (copylen < 12) { *w0 = copylen + 20; return 1; }
  if (copylen < 72) {
    const uint32_t tail = copylen - 8, nbits = log2floor(tail) - 1u, prefix = tail >> nbits;
    *w0 = ((nbits << 1) + prefix + 28) | ((tail - (prefix << nbits)) << 8);
    return 1;
  }
  if (copylen < 136) {
    const uint32_t tail = copylen - 8;
    *w0 = ((tail >> 5) + 54) | ((tail & 31u) << 8);
  } else if (copylen < 2120) {
    const uint32_t tail = copylen - 72, nbits = log2floor(tail);
    *w0 = (nbits + 52) | ((tail - (1u << nbits)) << 8);
  } else {
    *w0 = 63u | ((copylen - 2120) << 8);
  }
  return 2;
}
DEV uint32_t fast_distance_word(uint32_t distance) {
  const uint32_t d = distance + 3, nbits = log2floor(d) - 1u, prefix = (d >> nbits) & 1u;
  const uint32_t offset = (2u + prefix) << nbits;
  return (2u * (nbits - 1u) + prefix + 80u) | ((d - offset) << 8);
}
// Extra bits of a working-alphabet code (kNumExtraBits, :463-473).
DEV uint32_t fast_num_extra(uint32_t code) {
  if (code < 24) return k_ins_extra[code];
  if (code < 40) return k_copy_extra[code - 24];
  if (code < 64) return k_copy_extra[code - 40];
  if (code < 80) return 0;
  return (code - 80u) / 2u + 1u;
}

// ---- wave helpers ---------------------------------------------------------------------
// Common prefix of a[0, limit) and b[0, limit) (FindMatchLengthWithLimit), whole wave:
// 512 bytes per round, first differing lane by ballot.
DEV uint32_t fast_match_length(const uint8_t* a, const uint8_t* b, uint32_t limit) {
  const int lane = wave_lane();
  uint32_t done = 0;
  for (;;) {
    const uint32_t off = done + 8u * (uint32_t)lane;
    uint32_t m = 0;             // matching bytes of this lane's chunk
    bool stop = true;           // chunk not completely equal (or cut by the limit)
    if (off < limit) {
      const uint32_t n = umin(8u, limit - off);
      if (n == 8) {
        const uint64_t d = ld64(a + off) ^ ld64(b + off);
        m = d ? (uint32_t)(dev_ctz64(d) >> 3) : 8u;
        stop = d != 0;
      } else {
        while (m < n && a[off + m] == b[off + m]) ++m;
      }
    }
    const uint64_t sm = wave_ballot(stop);
    if (sm) {
      const int f = dev_ctz64(sm);
      return done + 8u * (uint32_t)f + wave_bcast(m, f);
    }
    done += 512;
  }
}

// Same-hash ranking of the lanes [0, n) of a step (see fast_create_commands): exact
// answer from wave_equal_neighbours, but only after an LDS scoreboard says two lanes
// may share a hash at all — on text they almost never do, and for a lone wave the
// n-iteration readlane loop is a visible part of a step.
#define FAST_SB_SLOTS 8192u
DEV void fast_rank_equal(uint32_t h, int n, uint8_t* sb, int* prev, int* next) {
  const int lane = wave_lane();
  uint8_t* p = sb + (h & (FAST_SB_SLOTS - 1u));
  if (lane < n) *p = (uint8_t)lane;
  wave_sync();
  const bool maybe = lane < n && *p != (uint8_t)lane;
  wave_sync();
  *prev = -1;
  *next = 64;
  if (wave_ballot(maybe) != 0) wave_equal_neighbours(h, n, prev, next);
}

DEV void fast_copy_literals(uint8_t* dst, const uint8_t* src, uint32_t n) {
  const int lane = wave_lane();
  if (n >= 1024) {
    // long literal runs (incompressible data): 4 KiB per step, four 16-byte loads in
    // flight per lane before the stores
    const uint32_t n4k = n & ~4095u, n16 = n & ~15u;
    uint32_t j = 16u * (uint32_t)lane;
    for (; j < n4k; j += 4096) {
      uint64_t v[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) { v[2 * q] = ld64(src + j + 1024u * q); v[2 * q + 1] = ld64(src + j + 1024u * q + 8); }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        __builtin_memcpy(dst + j + 1024u * q, &v[2 * q], 8);
        __builtin_memcpy(dst + j + 1024u * q + 8, &v[2 * q + 1], 8);
      }
    }
    for (; j < n16; j += 1024) {
      const uint64_t lo = ld64(src + j), hi = ld64(src + j + 8);
      __builtin_memcpy(dst + j, &lo, 8);
      __builtin_memcpy(dst + j + 8, &hi, 8);
    }
    for (uint32_t k = n16 + (uint32_t)lane; k < n; k += 64) dst[k] = src[k];
  } else {
    for (uint32_t j = (uint32_t)lane; j < n; j += 64) dst[j] = src[j];
  }
}

// The hash-table refresh after a copy (:354-386, :410-442): NE ordered (hash,
// position) stores inside the copy just made, then the probe at `ip`.  Lane j < NE
// holds store j, lane NE the probe; a later store to the same slot wins, and the
// probe sees the latest of them (or memory).  Returns the probe's candidate.
// `first`: the variant used after the first match of a scan, which with 4-byte
// hashes keys the third store with the first store's bytes again (:362-363).
template <int MM>
DEV uint32_t fast_refresh(const uint8_t* base, uint32_t ip, uint32_t* table, uint32_t shift, bool first,
                          uint8_t* sb) {
  const int lane = wave_lane();
  const int NE = MM == 4 ? 3 : 5;
  uint32_t h = 0xFFFFFFFFu, val = 0, tv = 0;
  if (lane <= NE) {
    uint32_t p = ip - (uint32_t)NE + (uint32_t)lane;   // position whose bytes are hashed
    val = p;
    if (MM == 4 && first && lane == 2) p = ip - 3;       // hashed bytes of store 0, value ip - 1
    h = fast_hash<MM>(ld64(base + p), shift);
    if (lane == NE) tv = table[h];
  }
  int prev, next;
  fast_rank_equal(h, NE + 1, sb, &prev, &next);
  const uint32_t pv = wave_shfl(val, prev < 0 ? 0 : prev);
  const uint32_t cand = prev >= 0 ? pv : tv;
  if (lane <= NE && next > NE) table[h] = val;
  wave_sync();
  return wave_bcast(cand, NE);
}

// ---- CreateCommands (:234-459) of one block, one wave ----------------------------------
// Positions are offsets from the fragment's first byte (`base`).
template <int MM>
DEV void fast_create_commands(const uint8_t* base, uint32_t in0, uint32_t block_size, uint32_t left,
                              uint32_t* table, uint32_t shift, uint32_t* cmds, uint8_t* lits,
                              uint32_t* ncmds_out, uint32_t* nlits_out, uint8_t* sb) {
  const int lane = wave_lane();
  uint32_t ncmds = 0, nlits = 0;
  const uint32_t ip_end = in0 + block_size;
  uint32_t next_emit = in0;
  if (block_size >= 16) {
    const uint32_t la = block_size - MM, lb = left - 16u;
    const uint32_t ip_limit = in0 + (la < lb ? la : lb);
    int32_t last_distance = -1;
    uint32_t ip = in0 + 1;
    uint32_t skip = 32;
    // Probes per step: on compressible data a scan ends within a few probes, and a
    // lone wave pays for every instruction of the step (the same-hash ranking below is
    // a loop over the participating lanes), so a scan starts 16 wide and doubles while
    // nothing is found; incompressible data runs 64 wide almost always.
    int width = 16;
    for (;;) {
      // -- the next `width` probes of the scan: probe t looks at ip + sum of (s >> 5) for
      //    s in [skip, skip + t), a closed form in skip + t
      const uint32_t s = skip + (uint32_t)lane, q = s >> 5, r = s & 31u;
      const uint32_t q0 = skip >> 5, r0 = skip & 31u;
      const uint32_t pos = ip + (16u * q * (q - 1u) + q * r) - (16u * q0 * (q0 - 1u) + q0 * r0);
      const uint32_t nxt = pos + q;
      const bool active = lane < width;
      const bool valid = active && nxt <= ip_limit;   // monotonic in the lane
      uint64_t x = 0, yl = 0;
      uint32_t h = 0xFFFFFFFFu, tv = 0;
      if (valid) {
        x = ld64(base + pos);
        if (last_distance > 0) yl = ld64(base + pos - (uint32_t)last_distance);
        h = fast_hash<MM>(x, shift);
        tv = table[h];
      }
      const int nv = dev_popc64(wave_ballot(valid));
      int prev, next;
      fast_rank_equal(h, nv, sb, &prev, &next);
      const uint32_t pv = wave_shfl(pos, prev < 0 ? 0 : prev);
      const uint32_t cand = prev >= 0 ? pv : tv;     // what table[h] holds when this probe runs
      bool hit_l = false, hit_t = false;
      if (valid) {
        hit_l = last_distance > 0 && fast_is_match<MM>(x, yl);
        hit_t = fast_is_match<MM>(x, ld64(base + cand)) && pos - cand <= FAST_MAX_DISTANCE;
      }
      const uint64_t stops = wave_ballot(active && (!valid || hit_l || hit_t));
      const int f = stops ? dev_ctz64(stops) : 64;
      const bool f_hit = f < nv;                      // lane f is a hit (valid lanes stop only on hits)
      const int last_commit = f_hit ? f : (f == 64 ? width - 1 : f - 1);   // probes that ran and wrote the table
      if (valid && lane <= last_commit && next > last_commit) table[h] = pos;
      wave_sync();
      if (f == 64) {
        skip += (uint32_t)width;
        ip = wave_bcast(nxt, width - 1);
        if (width < 64) width *= 2;
        continue;
      }
      if (!f_hit) break;                              // next_ip > ip_limit: emit_remainder
      // -- a match at lane f
      ip = wave_bcast(pos, f);
      const bool by_last = wave_bcast((uint32_t)hit_l, f) != 0;
      uint32_t candidate = by_last ? ip - (uint32_t)last_distance : wave_bcast(cand, f);
      {
        const uint32_t matched = MM + fast_match_length(base + candidate + MM, base + ip + MM, ip_end - ip - MM);
        const uint32_t distance = ip - candidate;
        const uint32_t insert = ip - next_emit;
        uint32_t w0;
        const uint32_t nw = fast_copy_last_words(matched, &w0);
        const bool same = (int32_t)distance == last_distance;
        if (lane == 0) {
          uint32_t* c = cmds + ncmds;
          c[0] = fast_insert_word(insert);
          c[1] = same ? 64u : fast_distance_word(distance);
          c[2] = w0;
          if (nw == 2) c[3] = 64u;
        }
        fast_copy_literals(lits + nlits, base + next_emit, insert);
        ncmds += 2u + nw;
        nlits += insert;
        last_distance = (int32_t)distance;
        ip += matched;
        next_emit = ip;
      }
      if (ip >= ip_limit) break;
      candidate = fast_refresh<MM>(base, ip, table, shift, true, sb);
      bool out_of_input = false;
      for (;;) {
        // matches that start exactly where the previous copy ended (:388-443)
        if (ip - candidate > FAST_MAX_DISTANCE) break;
        uint64_t xa = 0, xb = 0;
        if (lane == 0) { xa = ld64(base + ip); xb = ld64(base + candidate); }
        if (!wave_bcast((uint32_t)fast_is_match<MM>(xa, xb), 0)) break;
        const uint32_t matched = MM + fast_match_length(base + candidate + MM, base + ip + MM, ip_end - ip - MM);
        last_distance = (int32_t)(ip - candidate);
        if (lane == 0) {
          cmds[ncmds] = fast_copy_word(matched);
          cmds[ncmds + 1] = fast_distance_word((uint32_t)last_distance);
        }
        ncmds += 2;
        ip += matched;
        next_emit = ip;
        if (ip >= ip_limit) { out_of_input = true; break; }
        candidate = fast_refresh<MM>(base, ip, table, shift, false, sb);
      }
      if (out_of_input) break;
      ++ip;
      skip = 32;
      width = 16;
    }
  }
  // emit_remainder
  if (next_emit < ip_end) {
    const uint32_t insert = ip_end - next_emit;
    if (lane == 0) cmds[ncmds] = fast_insert_word(insert);
    fast_copy_literals(lits + nlits, base + next_emit, insert);
    ncmds += 1;
    nlits += insert;
  }
  *ncmds_out = ncmds;
  *nlits_out = nlits;
}

// One fragment: zero the table (GetHashTable, encode.c:156-189), then its blocks.
DEV void fast_parse_fragment(const FastArgs& a, uint32_t f, uint32_t* table, uint8_t* sb) {
  const int lane = wave_lane();
  const FastFrag F = a.frags[f];
  if (F.nblocks == 0) return;
  {
    uint64_t* t2 = (uint64_t*)table;
    const uint32_t n2 = (1u << F.table_bits) / 2u;
    for (uint32_t j = (uint32_t)lane; j < n2; j += 64) t2[j] = 0;
  }
  wave_sync();
  const uint8_t* base = a.input + F.in_off;
  const uint32_t shift = 64u - F.table_bits;
  for (uint32_t k = 0; k < F.nblocks; ++k) {
    const uint32_t bidx = F.first_block + k;
    const FastBlock B = a.blocks[bidx];
    uint32_t ncmds, nlits;
    if (F.table_bits <= 15)
      fast_create_commands<4>(base, B.off_in_frag, B.len, B.left, table, shift, fast_cmds(a, B), fast_lits(a, B), &ncmds, &nlits, sb);
    else
      fast_create_commands<6>(base, B.off_in_frag, B.len, B.left, table, shift, fast_cmds(a, B), fast_lits(a, B), &ncmds, &nlits, sb);
    if (lane == 0) {
      a.bstate[bidx].ncmds = ncmds;
      a.bstate[bidx].nlits = nlits;
      a.bstate[bidx].error = 0;
    }
    wave_sync();
  }
}

// Order in which BuildAndStoreCommandPrefixCode (:56-104) lines the 64 command codes
// of the working alphabet up for canonical code assignment.
static __device__ const uint8_t k_fast_order[64] = {
    24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47,
    0, 1, 2, 3, 4, 5, 6, 7, 48, 49, 50, 51, 52, 53, 54, 55, 8, 9, 10, 11, 12, 13, 14, 15,
    56, 57, 58, 59, 60, 61, 62, 63, 16, 17, 18, 19, 20, 21, 22, 23};

// LDS of one k_fast_store wave (dwords).  The prefix-code builder's work area (k_prefix.h) and
// the output window of the command stream are never in use at the same time.
#define FAST_WIN_DW 1024u
#define FS_HIST 0u                       // u32[256] literal histogram
#define FS_CHIST 256u                    // u32[128] command histogram
#define FS_LDEPTH 384u                   // u8[256]
#define FS_LBITS (FS_LDEPTH + 64u)       // u16[256]
#define FS_CDEPTH (FS_LBITS + 128u)      // u8[128]
#define FS_CBITS (FS_CDEPTH + 32u)       // u16[128]
#define FS_TMP8 (FS_CBITS + 64u)         // u8[64] + u16[64]: codes of the permuted command alphabet
#define FS_START (FS_TMP8 + 48u)         // u32[65]
#define FS_BASE (FS_START + 65u)         // u32[64]
#define FS_FLAG (FS_BASE + 64u)          // u32[3]
#define FS_WIN (FS_FLAG + 3u)            // u32[FAST_WIN_DW + 4] | the prefix-code work area
#define FS_PFX FS_WIN
#define FAST_STORE_LDS_WORDS (FS_WIN + (PFX_LDS_WORDS > FAST_WIN_DW + 4u ? PFX_LDS_WORDS : FAST_WIN_DW + 4u))

// One block after the parse: ShouldCompress (:524-544), then StoreCommands (:461-522)
// into the block's scratch as a bit string that starts at bit 0.
DEV void fast_store_block(const FastArgs& a, uint32_t bidx, uint32_t* lds) {
  const int lane = wave_lane();
  const FastBlock B = a.blocks[bidx];
  FastBlockState* S = &a.bstate[bidx];
  const uint32_t ncmds = S->ncmds, nlits = S->nlits, len = B.len;
  const uint8_t* in = a.input + B.in_off;
  const uint32_t* cmds = fast_cmds(a, B);
  const uint8_t* lits = fast_lits(a, B);
  uint32_t* lsum = fast_lsum(a, B, bidx);
  uint8_t* scr = fast_scratch(a, B, bidx);
  const uint64_t cap_bits = 8ull * fast_scratch_cap(B);

  uint32_t* hist = lds + FS_HIST;
  uint32_t* chist = lds + FS_CHIST;
  uint32_t* P = lds + FS_PFX;
  uint8_t* ldepth = (uint8_t*)(lds + FS_LDEPTH);
  uint16_t* lbits = (uint16_t*)(lds + FS_LBITS);
  uint8_t* cdepth = (uint8_t*)(lds + FS_CDEPTH);
  uint16_t* cbits_ = (uint16_t*)(lds + FS_CBITS);
  uint32_t* s_start = lds + FS_START;
  uint32_t* s_base = lds + FS_BASE;
  uint32_t* flag = lds + FS_FLAG;
  uint32_t* W = lds + FS_WIN;

  for (uint32_t j = (uint32_t)lane; j < 384u; j += 64) lds[j] = 0;   // both histograms
  wave_sync();
  // ---- ShouldCompress
  {
    const double corpus = (double)len;
    bool compress = (double)nlits < 0.98 * corpus;
    if (!compress) {
      for (uint32_t i = 43u * (uint32_t)lane; i < len; i += 43u * 64u) lds_atomic_add(&hist[in[i]], 1u);
      wave_sync();
      if (lane == 0) {
        const double max_total_bit_cost = corpus * 8 * 0.98 / 43;
        flag[0] = bits_entropy2(hist, nullptr, 256, a.T->log2_lut) < max_total_bit_cost ? 1u : 0u;
      }
      wave_sync();
      compress = flag[0] != 0;
      wave_sync();
      for (uint32_t j = (uint32_t)lane; j < 256u; j += 64) hist[j] = 0;
      wave_sync();
    }
    if (!compress) {
      if (lane == 0) S->bits = FAST_RAW;
      return;
    }
  }
  // ---- histograms
  for (uint32_t k = (uint32_t)lane; k < nlits; k += 64) lds_atomic_add(&hist[lits[k]], 1u);
  for (uint32_t k = (uint32_t)lane; k < ncmds; k += 64) lds_atomic_add(&chist[cmds[k] & 0xFFu], 1u);
  // ldepth, lbits, cdepth, cbits are contiguous
  for (uint32_t j = (uint32_t)lane; j < FS_START - FS_LDEPTH; j += 64) (lds + FS_LDEPTH)[j] = 0;
  wave_sync();
  // ---- header and the three prefix codes, the whole wave on each (k_prefix.h) ----
  uint32_t* bitbuf = P + PFX_BITBUF;       // (<= 256 used symbols per code here: the sort keys stay below it)
  uint32_t hbits = 0;
  for (uint32_t j = (uint32_t)lane; j < 128u; j += 64) bitbuf[j] = 0;
  wave_sync();
  {
    // BrotliStoreMetaBlockHeader (:216-232) + "no block splits, no contexts" (:586-588):
    // ISLAST 0, MNIBBLES, MLEN - 1, ISUNCOMPRESSED 0, thirteen zero bits
    const uint32_t nibbles = len <= (1u << 16) ? 4u : (len <= (1u << 20) ? 5u : 6u);
    uint32_t nb = 0, val = 0;
    if (lane == 0) { nb = 3; val = (nibbles - 4u) << 1; }
    else if (lane == 1) { nb = nibbles * 4u; val = len - 1u; }
    else if (lane == 2) { nb = 14; val = 0; }
    pfx_put_lanes(bitbuf, hbits, nb, val);
  }
  pfx_build_and_append<true>(hist, 256u, 8u, P, ldepth, lbits, bitbuf, hbits, false);
  if (lane == 0) { chist[1] += 1; chist[2] += 1; chist[64] += 1; chist[84] += 1; }
  wave_sync();
  {
    // BuildAndStoreCommandPrefixCode (:56-104): lengths of the two halves of the working
    // alphabet, codes assigned with the first half lined up in k_fast_order
    uint8_t* len8 = (uint8_t*)(P + PFX_LEN);
    uint8_t* t8 = (uint8_t*)(lds + FS_TMP8);
    uint16_t* t16 = (uint16_t*)(lds + FS_TMP8 + 16u);
    pfx_code_lengths<false>(chist, 64u, 15u, P);
    if (lane < 64) cdepth[lane] = len8[lane];
    wave_sync();
    pfx_code_lengths<false>(chist + 64, 64u, 14u, P);
    if (lane < 64) cdepth[64 + lane] = len8[lane];
    wave_sync();
    pfx_assign_codes(64u, P, cdepth + 64, cbits_ + 64);          // (len8 still holds the second half)
    if (lane < 64) len8[lane] = cdepth[k_fast_order[lane]];
    wave_sync();
    pfx_assign_codes(64u, P, t8, t16);
    if (lane < 64) cbits_[k_fast_order[lane]] = t16[lane];
    wave_sync();
    // the 704-symbol insert-and-copy alphabet the working alphabet stands for (:79-98)
    for (uint32_t j = (uint32_t)lane; j < 704u / 4u; j += 64) ((uint32_t*)len8)[j] = 0;
    wave_sync();
    if (lane < 40) {
      const uint32_t i = (uint32_t)lane & 7u, blk = (uint32_t)lane >> 3;      // cdepth[24 + 8 * blk + i]
      const uint32_t at = blk == 4u ? 384u : 64u * blk;
      len8[at + i] = cdepth[24 + lane];
    }
    wave_sync();                                   // (slot 128 is written again below, and that write stands)
    if (lane >= 40 && lane < 64) {
      const uint32_t i = ((uint32_t)lane - 40u) & 7u, blk = ((uint32_t)lane - 40u) >> 3;   // cdepth[8 * blk + i]
      const uint32_t at = blk == 0u ? 128u : blk == 1u ? 256u : 448u;
      len8[at + 8u * i] = cdepth[8u * blk + i];
    }
    wave_sync();
    pfx_store_complex<false>(704u, P, bitbuf, hbits);
    if (lane < 64) len8[lane] = cdepth[64 + lane];
    wave_sync();
    pfx_store_complex<false>(64u, P, bitbuf, hbits);
  }
  const uint64_t bit_cmds = hbits;
  for (uint32_t j = (uint32_t)lane; j < (hbits + 31u) / 32u; j += 64) st32(scr + 4u * j, bitbuf[j]);
  wave_sync();

  // ---- (a) running sum of literal bits: lsum[k] = bits of literals [0, k)
  {
    uint32_t carry = 0;
    for (uint32_t k0 = 0; k0 < nlits; k0 += 64) {
      const uint32_t k = k0 + (uint32_t)lane;
      const uint32_t nb = k < nlits ? ldepth[lits[k]] : 0u;
      const uint32_t incl = wave_incl_scan(nb);
      if (k < nlits) lsum[k] = carry + incl - nb;
      carry += wave_bcast(incl, 63);
    }
    if (lane == 0) lsum[nlits] = carry;
    wave_sync();
  }
  // ---- (b) 64 command words per step into an LDS window (see k_store.h phase 3)
  uint32_t* out32 = (uint32_t*)scr;
  uint64_t wbit = bit_cmds & ~(uint64_t)31;
  for (uint32_t j = (uint32_t)lane; j < FAST_WIN_DW + 4u; j += 64) W[j] = 0;
  wave_sync();
  if (lane == 0) {
    const uint32_t part = (uint32_t)(bit_cmds & 31u);
    W[0] = part ? (out32[wbit >> 5] & ((1u << part) - 1u)) : 0u;
  }
  wave_sync();
  uint32_t lit_base = 0, cbits = 0;
  bool overflow = false;
  for (uint32_t base = 0; base < ncmds; base += 64) {
    const uint32_t i = base + (uint32_t)lane;
    const bool valid = i < ncmds;
    const uint32_t word = valid ? cmds[i] : 0u;
    const uint32_t code = word & 0xFFu, extra = word >> 8;
    const uint32_t ins = (valid && code < 24u) ? k_ins_base[code] + extra : 0u;
    const uint32_t ins_incl = wave_incl_scan(ins);
    const uint32_t my_lit = lit_base + ins_incl - ins;
    const uint32_t cn = valid ? cdepth[code] : 0u;
    const uint32_t xn = valid ? fast_num_extra(code) : 0u;
    const uint64_t cv = (uint64_t)cbits_[code] | ((uint64_t)extra << cn);
    const uint32_t own = cn + xn;
    const uint32_t own_incl = wave_incl_scan(own);
    const uint32_t ls = valid ? lsum[my_lit] : 0u;
    const uint64_t p0 = bit_cmds + cbits + (own_incl - own) + ls;
    const uint32_t total_ins = wave_bcast(ins_incl, 63);
    const uint32_t total_own = wave_bcast(own_incl, 63);
    const uint64_t span_end = bit_cmds + cbits + total_own + lsum[lit_base + total_ins];
    if (span_end + 64u > cap_bits) { overflow = true; break; }
    const bool in_window = span_end - wbit <= (uint64_t)FAST_WIN_DW * 32u;
    s_start[lane] = my_lit;
    s_base[lane] = (uint32_t)(p0 + own - bit_cmds) - ls;
    if (lane == 63) s_start[64] = lit_base + total_ins;
    wave_sync();
    if (!in_window) {
      // a span longer than the window is written with dword atomics: the partial
      // dword goes back to memory, the dwords behind it start from zero
      if (lane == 0) { out32[wbit >> 5] = W[0]; W[0] = 0; }
      const uint64_t d0 = (wbit >> 5) + 1, d1 = (span_end + 31) >> 5;
      for (uint64_t j = d0 + (uint32_t)lane; j <= d1; j += 64) out32[j] = 0;
      wave_mem_barrier();
    }
    if (valid) {
      if (in_window) lds_or_bits(W, (uint32_t)(p0 - wbit), own, cv);
      else or_bits(out32, p0, own, cv);
    }
    for (uint32_t L = lit_base + (uint32_t)lane; L < lit_base + total_ins; L += 64) {
      uint32_t lo = 0, hi = 63;
      while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (s_start[mid] <= L) lo = mid; else hi = mid - 1;
      }
      const uint64_t pos = bit_cmds + s_base[lo] + lsum[L];
      const uint32_t v = lits[L];
      if (in_window) lds_or_bits(W, (uint32_t)(pos - wbit), ldepth[v], lbits[v]);
      else or_bits(out32, pos, ldepth[v], lbits[v]);
    }
    wave_sync();
    if (in_window) {
      const uint32_t ndw = (uint32_t)((span_end - wbit) >> 5);
      const uint32_t carry = W[ndw];
      wave_sync();
      uint32_t* dst = out32 + (wbit >> 5);
      for (uint32_t j = (uint32_t)lane; j <= ndw; j += 64) {
        if (j < ndw) dst[j] = W[j];
        W[j] = 0;
      }
      wave_sync();
      if (lane == 0) W[0] = carry;
      wbit += (uint64_t)ndw * 32u;
    } else {
      wave_mem_barrier();
      wbit = span_end & ~(uint64_t)31;
      if (lane == 0) {
        const uint32_t part = (uint32_t)(span_end & 31u);
        W[0] = part ? (glb_atomic_or(out32 + (wbit >> 5), 0u) & ((1u << part) - 1u)) : 0u;
      }
    }
    wave_sync();
    cbits += total_own;
    lit_base += total_ins;
  }
  if (overflow) {
    if (lane == 0) { S->error = 1; S->bits = FAST_RAW; glb_atomic_or((uint32_t*)&a.result[1], 1u); }
    return;
  }
  if (lane == 0) {
    out32[wbit >> 5] = W[0];
    S->bits = (uint32_t)(bit_cmds + cbits + lsum[nlits]);
  }
}

// ---- placement -----------------------------------------------------------------------------
DEV uint32_t fast_header_bits(uint32_t len) {   // BrotliStoreMetaBlockHeader, :216-232
  return 4u + 4u * (len <= (1u << 16) ? 4u : (len <= (1u << 20) ? 5u : 6u));
}
DEV uint64_t fast_raw_header(uint32_t len) {    // ISLAST 0, MNIBBLES, MLEN - 1, ISUNCOMPRESSED 1
  const uint32_t nib = len <= (1u << 16) ? 4u : (len <= (1u << 20) ? 5u : 6u);
  return ((uint64_t)(nib - 4u) << 1) | ((uint64_t)(len - 1u) << 3) | (1ull << (3u + 4u * nib));
}
// Bit position after block `bidx` when it starts at `pos`.
DEV uint64_t fast_advance(const FastArgs& a, uint32_t bidx, uint64_t pos) {
  const uint32_t bits = a.bstate[bidx].bits;
  if (bits != FAST_RAW) return pos + bits;
  const uint32_t len = a.blocks[bidx].len;
  return ((pos + fast_header_bits(len) + 7u) & ~(uint64_t)7u) + 8ull * len;   // EmitUncompressedMetaBlock :554-562
}

// One thread per (fragment, start residue): bits the fragment occupies, and whether
// BrotliCompressFragmentTwoPass replaces it by one raw meta-block (:622-627).
DEV void fast_fragment_sizes(const FastArgs& a, uint32_t f, uint32_t r) {
  const FastFrag F = a.frags[f];
  uint64_t pos = r;
  for (uint32_t k = 0; k < F.nblocks; ++k) pos = fast_advance(a, F.first_block + k, pos);
  bool rewrite = false;
  if (pos - r > 31u + 8ull * F.len) {
    rewrite = true;
    pos = ((r + fast_header_bits(F.len) + 7u) & ~7u) + 8ull * F.len;
  }
  a.fstate[f].bits[r] = pos - r;
  if (rewrite) glb_atomic_or(&a.fstate[f].rewrite_mask, 1u << r);
}

// One wave: the start bit of every fragment.  Lane l owns a contiguous range of
// fragments; a range maps a start residue to (bits, end residue), ranges compose.
DEV void fast_scan_fragments(const FastArgs& a) {
  const int lane = wave_lane();
  const uint32_t per = (a.nfrags + 63u) / 64u;
  const uint32_t f0 = umin(a.nfrags, per * (uint32_t)lane), f1 = umin(a.nfrags, f0 + per);
  uint64_t comp[8];
  for (int r = 0; r < 8; ++r) comp[r] = 0;
  for (uint32_t f = f0; f < f1; ++f) {
    uint64_t row[8];
    for (int r = 0; r < 8; ++r) row[r] = a.fstate[f].bits[r];
    for (int r = 0; r < 8; ++r) comp[r] += row[(r + comp[r]) & 7u];
  }
  uint64_t start = a.carry_bits;      // running position, uniform
  uint64_t my_start = 0;
  for (int l = 0; l < 64; ++l) {
    if (lane == l) my_start = start;
    uint64_t sel = 0;
    for (int r = 0; r < 8; ++r) if ((int)(start & 7u) == r) sel = comp[r];
    start += wave_bcast64(sel, l);
  }
  uint64_t pos = my_start;
  for (uint32_t f = f0; f < f1; ++f) {
    a.fstate[f].start = pos;
    pos += a.fstate[f].bits[pos & 7u];
  }
  if (lane == 0) {
    uint32_t* out32 = (uint32_t*)a.out;
    uint64_t total = start;
    bool bad = (total + 16u + 7u) / 8u > a.out_cap;
    if (!bad) {
      if (a.carry_bits) glb_atomic_or(out32, a.carry_value & ((1u << a.carry_bits) - 1u));
      if (a.is_last) {
        or_bits(out32, total, 2, 3);            // ISLAST, ISLASTEMPTY (:629-633)
        total = (total + 2u + 7u) & ~(uint64_t)7u;
      }
    }
    a.result[0] = total;
    if (bad) glb_atomic_or((uint32_t*)&a.result[1], 2u);
  }
}

// Copies `nbits` bits of `src` (bit 0 = bit 0 of src[0]) to bit position dst_bit of the
// zero-initialised output.  Interior dwords are plain stores, the two edge dwords
// are OR-ed in (neighbouring pieces share them).  `src_avail` = readable bytes at src.
DEV void fast_copy_bits(uint32_t* out32, uint64_t dst_bit, const uint8_t* src, uint64_t nbits,
                        uint64_t src_avail, uint32_t tid, uint32_t nthreads) {
  if (nbits == 0) return;
  const uint64_t d0 = dst_bit >> 5, d1 = (dst_bit + nbits - 1) >> 5;
  const uint32_t sh = (uint32_t)(dst_bit & 31u);
  for (uint64_t j = d0 + tid; j <= d1; j += nthreads) {
    // source bit that lands on bit 0 of output dword j (negative only for j == d0)
    const int64_t sb = (int64_t)(32ull * j) - (int64_t)dst_bit;
    uint32_t v;
    uint64_t have;                  // source bits available from max(sb, 0)
    if (sb < 0) {
      const uint32_t w0 = src_avail >= 4 ? ld32(src) : (uint32_t)(src[0] | (src_avail > 1 ? src[1] << 8 : 0) | (src_avail > 2 ? src[2] << 16 : 0));
      v = w0 << sh;
      have = nbits;
      if (have < 32u - sh) v &= (1u << ((uint32_t)have + sh)) - 1u;
    } else {
      const uint64_t dw = (uint64_t)sb >> 5;
      const uint32_t s2 = (uint32_t)sb & 31u;
      uint64_t w;
      if (4 * dw + 8 <= src_avail) {
        w = ld64(src + 4 * dw);
      } else {
        w = 0;
        for (uint32_t q = 0; q < 8; ++q) if (4 * dw + q < src_avail) w |= (uint64_t)src[4 * dw + q] << (8 * q);
      }
      v = (uint32_t)(w >> s2);
      have = nbits - (uint64_t)sb;
      if (have < 32) v &= (1u << (uint32_t)have) - 1u;
    }
    if (j == d0 || j == d1) { if (v) glb_atomic_or(out32 + j, v); }
    else out32[j] = v;
  }
}

// One workgroup per block: its bits (or raw bytes) go to their place in the output.
DEV void fast_emit_block(const FastArgs& a, uint32_t bidx, uint32_t tid, uint32_t nthreads) {
  if (a.result[1]) return;
  const FastBlock B = a.blocks[bidx];
  const FastFrag F = a.frags[B.frag];
  const uint64_t fstart = a.fstate[B.frag].start;
  const bool rewritten = (a.fstate[B.frag].rewrite_mask >> (fstart & 7u)) & 1u;
  uint32_t* out32 = (uint32_t*)a.out;
  const uint8_t* in = a.input + B.in_off;
  const uint64_t in_avail = (uint64_t)B.len + FAST_INPUT_SLACK;
  if (rewritten) {
    const uint32_t hb = fast_header_bits(F.len);
    if (B.off_in_frag == 0 && tid == 0) or_bits(out32, fstart, hb, fast_raw_header(F.len));
    const uint64_t data = ((fstart + hb + 7u) & ~(uint64_t)7u) + 8ull * B.off_in_frag;
    fast_copy_bits(out32, data, in, 8ull * B.len, in_avail, tid, nthreads);
    return;
  }
  uint64_t pos = fstart;
  for (uint32_t b = F.first_block; b < bidx; ++b) pos = fast_advance(a, b, pos);
  const uint32_t bits = a.bstate[bidx].bits;
  if (bits == FAST_RAW) {
    const uint32_t hb = fast_header_bits(B.len);
    if (tid == 0) or_bits(out32, pos, hb, fast_raw_header(B.len));
    fast_copy_bits(out32, (pos + hb + 7u) & ~(uint64_t)7u, in, 8ull * B.len, in_avail, tid, nthreads);
  } else {
    fast_copy_bits(out32, pos, fast_scratch(a, B, bidx), bits, fast_scratch_cap(B) + 8u, tid, nthreads);
  }
}

#endif  // BROTLI_AMD_CSRC_K_FAST_H_
