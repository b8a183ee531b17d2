// brotli_amd/csrc/k_prefix.h — prefix codes built by a whole wavefront.
//
// What has to come out (bit-exact): the code lengths BrotliCreateHuffmanTree assigns
// (c/enc/entropy_encode.c:68-147: leaves ordered by (count, symbol descending), two-queue
// merge that prefers a leaf on ties, counts raised to 1, 2, 4 ... until the tree fits the
// length limit), the canonical codes of BrotliConvertBitDepthsToSymbols (:454-497), and the
// serialised code of BuildAndStoreHuffmanTree / BrotliStoreHuffmanTree
// (c/enc/brotli_bit_stream.c:242-397) with the run-length symbols of BrotliWriteHuffmanTree
// (entropy_encode.c:160-239, 372-452).  RFC 7932 sections 3.4 / 3.5 fix the format; the
// reference fixes the tie-breaking — the way to get there is this file's own:
//
//   * leaves are compacted with ballots and ranked by a bitonic sort of 64-bit keys
//     (count << 10 | 1023 - symbol) in LDS, all compare-exchanges ascending, the padding to a
//     power of two virtual;
//   * the two-queue merge — the one inherently serial piece, m - 1 steps — keeps the counts of
//     the waiting inner nodes and the parent of every finished node in ONE array (a slot holds
//     the count until the node is consumed, its parent afterwards);
//   * depths come from pointer jumping over (parent, distance) pairs packed in a word, log2
//     rounds for the whole tree instead of a stack walk;
//   * canonical codes: per-length counters in LDS, the rank of a symbol among the symbols of
//     its length from a 4-bit match-any over the wave;
//   * the code-length sequence is run-length coded by ballots: run starts compacted, every run
//     sizes its own output (literal + base-4 / base-8 repeat digits), a wave scan places it,
//     and the bits are OR-ed into an LDS buffer after a second scan over their lengths.
//
// One code at a time per wave; `P` is PFX_LDS_WORDS dwords of LDS.
#ifndef BROTLI_AMD_CSRC_K_PREFIX_H_
#define BROTLI_AMD_CSRC_K_PREFIX_H_

#include "device_common.h"

#define PFX_MAX_SYMS 704u
// LDS map, dwords.  A: sort keys, then {counts, symbols}; C: count-or-parent per node, then
// (parent, distance) pairs; then the code lengths.  The serialisation arrays live in A (free
// by then) above the 64 dwords a nested 18-symbol build touches.
#define PFX_A 0u               // u64 keys[704] | u32 cnt[704] @ 0, u16 sym[704] @ 704
#define PFX_SYM 704u
#define PFX_C 1408u            // u32 node[1408]
#define PFX_LEN 2816u          // u8 len[704]
#define PFX_LDS_WORDS 2992u
#define PFX_OUT 64u            // u8 out[704]: code-length symbols 0 .. 17
#define PFX_EXTRA 240u         // u8 extra[704]
#define PFX_STARTS 416u        // u16 starts[706]
#define PFX_BITBUF 832u        // u32 bitbuf[128]: the serialised code
#define PFX_SMALL 960u         // 80 dwords: per-length counters / next codes / the code-length code

// ---- pieces --------------------------------------------------------------------------------
// Ascending sort of keys[0 .. m) (m <= 1024): bitonic network in the form where every
// compare-exchange is ascending (the first step of a merge pairs i with its mirror image in the
// block), so elements beyond m behave like +infinity without being stored.
DEV void pfx_sort(uint64_t* keys, uint32_t m) {
  const uint32_t lane = (uint32_t)wave_lane();
  uint32_t M = 2;
  while (M < m) M <<= 1;
  for (uint32_t k = 2; k <= M; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      const bool mirror = j == (k >> 1);
      for (uint32_t t = lane; t < (M >> 1); t += 64u) {
        // the t-th pair of this step: i has bit j clear
        const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u));
        const uint32_t p = mirror ? (i ^ (k - 1u)) : (i | j);
        if (p < m) {
          const uint64_t a = keys[i], b = keys[p];
          if (a > b) { keys[i] = b; keys[p] = a; }
        }
      }
      wave_sync();
    }
  }
}

// The order SortHuffmanTreeItems (c/enc/entropy_encode.h:82-115) leaves m items in when its
// comparator looks at the count alone (the quality-1 writer, brotli_bit_stream.c:399-402).
// Every pass of that sort — gap 132, 57, 23, 10, 4, 1, or one plain insertion pass below 13
// items — is a STABLE sort of each residue class modulo the gap, so the result is reproduced
// without moving a single element serially: per pass, every item counts the members of its class
// that must precede it (smaller count, or equal count and earlier place) and that is its new
// place in the class.  keys: count << 10 | payload, compared by count only.
DEV void pfx_sort_by_count_in_reference_order(uint64_t* keys, uint32_t m) {
  const uint32_t lane = (uint32_t)wave_lane();
  uint32_t gaps[6] = {132u, 57u, 23u, 10u, 4u, 1u};
  const int first = m < 13u ? 5 : m < 57u ? 2 : 0;
  for (int gi = first; gi < 6; ++gi) {
    const uint32_t g = gaps[gi];
    if (g >= m) continue;
    uint64_t mine[4];
    uint32_t place[4];
#pragma unroll
    for (uint32_t r = 0; r < 4u; ++r) {          // (m <= 256)
      const uint32_t x = r * 64u + lane;
      mine[r] = 0; place[r] = 0;
      if (x < m) {
        mine[r] = keys[x];
        const uint32_t c = (uint32_t)(mine[r] >> 10);
        uint32_t before = 0;
        for (uint32_t y = x % g; y < m; y += g) {
          const uint32_t cy = (uint32_t)(keys[y] >> 10);
          before += (cy < c || (cy == c && y < x)) ? 1u : 0u;
        }
        place[r] = x % g + g * before;
      }
    }
    wave_sync();
#pragma unroll
    for (uint32_t r = 0; r < 4u; ++r) if (r * 64u + lane < m) keys[place[r]] = mine[r];
    wave_sync();
  }
}

// Lanes (of those with `act`) whose 4-bit value equals this lane's.
DEV uint64_t pfx_match4(bool act, uint32_t v) {
  uint64_t same = wave_ballot(act);
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const bool bit = (v >> b) & 1u;
    const uint64_t mk = wave_ballot(act && bit);
    same &= bit ? mk : ~mk;
  }
  return act ? same : 0ull;
}

// Code lengths of `histo[0 .. n)` (n <= 704) under the length limit, into len[] (LDS, PFX_LEN);
// returns the number of symbols with a non-zero count.  `histo` may be global memory or LDS
// outside A / C.  Q1: the quality-1 writer's leaf order (count only, n <= 256).
template <bool Q1>
DEV uint32_t pfx_code_lengths(const uint32_t* histo, uint32_t n, uint32_t limit, uint32_t* P) {
  const uint32_t lane = (uint32_t)wave_lane();
  uint64_t* keys = (uint64_t*)(P + PFX_A);
  uint32_t* cnt = P + PFX_A;
  uint16_t* sym = (uint16_t*)(P + PFX_SYM);
  uint32_t* node = P + PFX_C;
  uint8_t* len = (uint8_t*)(P + PFX_LEN);
  for (uint32_t floor_count = 1;; floor_count <<= 1) {
    wave_sync();
    for (uint32_t i = lane; i < (n + 3u) / 4u; i += 64u) ((uint32_t*)len)[i] = 0;
    // leaves, compacted (Q1: in the order the reference lines them up before its sort, highest
    // symbol first)
    uint32_t m = 0, total = 0;
    if (Q1) {
      for (uint32_t base = 0; base < n; base += 64u) {
        const uint32_t i = base + lane;
        total += (uint32_t)dev_popc64(wave_ballot(i < n && histo[i] != 0));
      }
    }
    for (uint32_t base = 0; base < n; base += 64u) {
      const uint32_t i = base + lane;
      const uint32_t c = i < n ? histo[i] : 0u;
      const uint64_t mk = wave_ballot(c != 0);
      const uint32_t at = m + (uint32_t)dev_popc64(mk & ((1ull << lane) - 1ull));
      if (c != 0) keys[Q1 ? total - 1u - at : at] = ((uint64_t)umax(c, floor_count) << 10) | (uint64_t)(1023u - i);
      m += (uint32_t)dev_popc64(mk);
    }
    wave_sync();
    if (m == 0) return 0;
    if (m == 1) {
      if (lane == 0) len[1023u - (uint32_t)(keys[0] & 1023u)] = 1;
      wave_sync();
      return 1;
    }
    if (Q1) pfx_sort_by_count_in_reference_order(keys, m); else pfx_sort(keys, m);
    {
      // keys -> counts + symbols in the same place: through registers
      uint64_t k[PFX_MAX_SYMS / 64u];
#pragma unroll
      for (uint32_t r = 0; r < PFX_MAX_SYMS / 64u; ++r) {
        const uint32_t i = r * 64u + lane;
        k[r] = i < m ? keys[i] : 0ull;
      }
      wave_sync();
#pragma unroll
      for (uint32_t r = 0; r < PFX_MAX_SYMS / 64u; ++r) {
        const uint32_t i = r * 64u + lane;
        if (i < m) { cnt[i] = (uint32_t)(k[r] >> 10); sym[i] = (uint16_t)(1023u - (uint32_t)(k[r] & 1023u)); }
      }
      wave_sync();
    }
    // Two queues: leaves i .. m in sorted order, inner nodes j .. k in the order they were made.
    // node[v] of a waiting inner node is its count; once v is consumed it is v's parent.
    {
      uint32_t i = 0, j = 0;
      for (uint32_t k = 0; k + 1u < m; ++k) {
        uint32_t pick[2], sum = 0;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const uint32_t lc = i < m ? cnt[i] : 0xFFFFFFFFu;
          const uint32_t ic = j < k ? node[m + j] : 0xFFFFFFFFu;
          if (lc <= ic) { pick[s] = i; sum += lc; ++i; } else { pick[s] = m + j; sum += ic; ++j; }
        }
        wave_sync();                       // (every lane has read the counts it needs)
        if (lane == 0) { node[pick[0]] = m + k; node[pick[1]] = m + k; node[m + k] = sum; }
        wave_sync();
      }
    }
    // depth of every node: (parent, distance to it) pairs, doubled until every parent is the root
    const uint32_t nn = 2u * m - 1u, root = nn - 1u;
    for (uint32_t v = lane; v < nn; v += 64u) node[v] = v == root ? (root << 16) : ((node[v] << 16) | 1u);
    wave_sync();
    for (uint32_t round = 0; round < 12u; ++round) {
      bool open = false;
      for (uint32_t v = lane; v < nn; v += 64u) {
        const uint32_t e = node[v], pe = node[e >> 16];
        // (pe is either the pair of this round or of the last one: both describe a real ancestor)
        node[v] = (pe & 0xFFFF0000u) | ((e & 0xFFFFu) + (pe & 0xFFFFu));
        open = open || (pe >> 16) != root;
      }
      wave_sync();
      if (!wave_ballot(open)) break;
    }
    uint32_t deepest = 0;
    for (uint32_t i = lane; i < m; i += 64u) deepest = umax(deepest, node[i] & 0xFFFFu);
    if (wave_max_u32(deepest) > limit) continue;      // raise the floor under the counts, again
    for (uint32_t i = lane; i < m; i += 64u) len[sym[i]] = (uint8_t)(node[i] & 0xFFFFu);
    wave_sync();
    return m;
  }
}

// Canonical codes of len[0 .. n) (bit-reversed for the LSB-first stream): bits16[] / depth8[]
// are the caller's arrays (global memory or LDS outside PFX_SMALL).
DEV void pfx_assign_codes(uint32_t n, uint32_t* P, uint8_t* depth8, uint16_t* bits16) {
  const uint32_t lane = (uint32_t)wave_lane();
  const uint8_t* len = (const uint8_t*)(P + PFX_LEN);
  uint32_t* per_len = P + PFX_SMALL;        // [16] symbols of each length, then the running rank base
  uint32_t* next = P + PFX_SMALL + 16u;     // [16] first code of each length
  wave_sync();
  if (lane < 16u) per_len[lane] = 0;
  wave_sync();
  for (uint32_t i = lane; i < n; i += 64u) { const uint32_t d = len[i]; if (d != 0) lds_atomic_add(&per_len[d], 1u); }
  wave_sync();
  if (lane == 0) {
    uint32_t code = 0;
    next[0] = 0;
    for (uint32_t d = 1; d < 16u; ++d) { code = (code + per_len[d - 1u]) << 1; next[d] = code; }
    for (uint32_t d = 0; d < 16u; ++d) per_len[d] = 0;
  }
  wave_sync();
  for (uint32_t base = 0; base < n; base += 64u) {
    const uint32_t i = base + lane;
    const uint32_t d = i < n ? len[i] : 0u;
    const uint64_t same = pfx_match4(d != 0, d);
    const uint32_t below = (uint32_t)dev_popc64(same & ((1ull << lane) - 1ull));
    uint32_t seen = 0;
    if (d != 0) seen = per_len[d];
    wave_sync();
    if (d != 0 && below + 1u == (uint32_t)dev_popc64(same)) per_len[d] = seen + below + 1u;
    wave_sync();
    if (i < n) {
      depth8[i] = (uint8_t)d;
      if (d != 0) bits16[i] = (uint16_t)(dev_bitrev32(next[d] + seen + below) >> (32u - d));
    }
  }
  wave_sync();
}

// Appends (value, nbits <= 32) of every lane, in lane order, to the LDS bit buffer.
DEV void pfx_put_lanes(uint32_t* bitbuf, uint32_t& bitpos, uint32_t nbits, uint32_t value) {
  const uint32_t incl = wave_incl_scan(nbits);
  const uint32_t at = bitpos + incl - nbits;
  if (nbits != 0) {
    const uint64_t v = (uint64_t)(value & (nbits >= 32u ? 0xFFFFFFFFu : ((1u << nbits) - 1u))) << (at & 31u);
    lds_atomic_or(&bitbuf[at >> 5], (uint32_t)v);
    if ((uint32_t)(v >> 32) != 0) lds_atomic_or(&bitbuf[(at >> 5) + 1u], (uint32_t)(v >> 32));
  }
  bitpos += wave_bcast(incl, 63);
}

// Repeat digits after the first three repetitions: base (1 << shift) with the "bijective"
// carry of RFC 7932 3.5 (a further symbol 16 / 17 multiplies what has been repeated so far).
DEV uint32_t pfx_digits(uint32_t reps, uint32_t shift) {
  uint32_t nd = 1;
  for (reps >>= shift; reps != 0; reps = (reps - 1u) >> shift) ++nd;
  return nd;
}

// The complex form (RFC 7932 3.5) of the code whose lengths sit in len[0 .. n): header with the
// code-length code, then the run-length coded length sequence, appended to bitbuf.
// Q1 (BrotliBuildAndStoreHuffmanTreeFast, brotli_bit_stream.c:516-571): a fixed code-length code
// (entropy_encode_static.h: symbols 0 .. 12, 16, 17 four bits, 13 and 14 five), runs always coded,
// no split of a run of seven.
template <bool Q1>
DEV void pfx_store_complex(uint32_t n, uint32_t* P, uint32_t* bitbuf, uint32_t& bitpos) {
  const uint32_t lane = (uint32_t)wave_lane();
  const uint8_t* len = (const uint8_t*)(P + PFX_LEN);
  uint8_t* out = (uint8_t*)(P + PFX_OUT);
  uint8_t* extra = (uint8_t*)(P + PFX_EXTRA);
  uint16_t* starts = (uint16_t*)(P + PFX_STARTS);
  uint32_t* small = P + PFX_SMALL;
  wave_sync();
  // the sequence ends behind the last used symbol
  uint32_t used = 0;
  for (uint32_t base = 0; base < n; base += 64u) {
    const uint32_t i = base + lane;
    const uint64_t mk = wave_ballot(i < n && len[i] != 0);
    if (mk) used = base + 64u - (uint32_t)__builtin_clzll(mk);
  }
  // maximal runs of equal lengths; decide per class (zero / non-zero) whether runs pay
  // (entropy_encode.c:332-370: only sequences longer than 50 are looked at)
  bool rle_zero = Q1, rle_other = Q1;
  if (!Q1 && n > 50u) {
    uint32_t nruns = 0;
    for (uint32_t base = 0; base < used; base += 64u) {
      const uint32_t i = base + lane;
      const bool st = i < used && (i == 0 || len[i] != len[i - 1u]);
      const uint64_t mk = wave_ballot(st);
      if (st) starts[nruns + (uint32_t)dev_popc64(mk & ((1ull << lane) - 1ull))] = (uint16_t)i;
      nruns += (uint32_t)dev_popc64(mk);
    }
    if (lane == 0) starts[nruns] = (uint16_t)used;
    wave_sync();
    uint32_t tz = 0, cz = 0, to = 0, co = 0;
    for (uint32_t r = lane; r < nruns; r += 64u) {
      const uint32_t s = starts[r], reps = (uint32_t)starts[r + 1u] - s;
      if (len[s] == 0) { if (reps >= 3u) { tz += reps; ++cz; } }
      else if (reps >= 4u) { to += reps; ++co; }
    }
    tz = wave_bcast(wave_incl_scan(tz), 63); cz = wave_bcast(wave_incl_scan(cz), 63);
    to = wave_bcast(wave_incl_scan(to), 63); co = wave_bcast(wave_incl_scan(co), 63);
    rle_zero = tz > (cz + 1u) * 2u;
    rle_other = to > (co + 1u) * 2u;
    wave_sync();
  }
  // coding runs: a class without run-length coding is written length by length
  uint32_t nruns = 0;
  for (uint32_t base = 0; base < used; base += 64u) {
    const uint32_t i = base + lane;
    bool st = false;
    if (i < used) {
      const uint32_t d = len[i];
      st = i == 0 || d != len[i - 1u] || !(d == 0 ? rle_zero : rle_other);
    }
    const uint64_t mk = wave_ballot(st);
    if (st) starts[nruns + (uint32_t)dev_popc64(mk & ((1ull << lane) - 1ull))] = (uint16_t)i;
    nruns += (uint32_t)dev_popc64(mk);
  }
  if (lane == 0) starts[nruns] = (uint16_t)used;
  if (lane < 18u) small[32u + lane] = 0;            // histogram of the code-length symbols
  wave_sync();
  // every run sizes and writes its own symbols
  uint32_t total = 0;
  uint32_t last_nz = 8;                              // the length a symbol 16 repeats (initially 8)
  for (uint32_t base = 0; base < nruns; base += 64u) {
    const uint32_t r = base + lane;
    const bool act = r < nruns;
    const uint32_t s = act ? starts[r] : 0u;
    uint32_t reps = act ? (uint32_t)starts[r + 1u] - s : 0u;
    const uint32_t d = act ? len[s] : 0u;
    // the non-zero length before this run: the nearest non-zero run of this row, else the carry
    const uint64_t nzm = wave_ballot(act && d != 0);
    const uint64_t lower = nzm & ((1ull << lane) - 1ull);
    const int src = lower ? 63 - __builtin_clzll(lower) : 0;
    const uint32_t prev_in_row = wave_shfl(d, src);
    const uint32_t prev = lower ? prev_in_row : last_nz;
    uint32_t lit = 0, nd = 0, rest = 0;              // literal symbols, repeat digits, what the digits encode
    if (act) {
      if (d == 0) {
        if (reps == 11u) { lit = 1; reps = 10; }
        if (reps < 3u) { lit += reps; } else { rest = reps - 3u; nd = pfx_digits(rest, 3); }
      } else {
        if (prev != d) { lit = 1; --reps; }
        if (!Q1 && reps == 7u) { ++lit; --reps; }
        if (reps < 3u) { lit += reps; } else { rest = reps - 3u; nd = pfx_digits(rest, 2); }
      }
    }
    const uint32_t cnt = lit + nd;
    const uint32_t incl = wave_incl_scan(cnt);
    uint32_t at = total + incl - cnt;
    if (act) {
      for (uint32_t q = 0; q < lit; ++q) { out[at + q] = (uint8_t)d; extra[at + q] = 0; }
      if (lit != 0) lds_atomic_add(&small[32u + d], lit);
      at += lit;
      if (nd != 0) {
        const uint32_t shift = d == 0 ? 3u : 2u, code = d == 0 ? 17u : 16u;
        // least significant digit last
        uint32_t v = rest;
        for (uint32_t q = nd; q-- != 0;) {
          out[at + q] = (uint8_t)code;
          extra[at + q] = (uint8_t)(v & ((1u << shift) - 1u));
          v >>= shift;
          if (q != 0) --v;
        }
        lds_atomic_add(&small[32u + code], nd);
      }
    }
    total += wave_bcast(incl, 63);
    if (nzm) last_nz = wave_bcast(d, 63 - __builtin_clzll(nzm));
  }
  wave_sync();
  uint8_t* cl_len = (uint8_t*)(P + PFX_SMALL + 52u);     // [18]
  uint16_t* cl_bits = (uint16_t*)(P + PFX_SMALL + 57u);  // [18]
  uint32_t zero_len_code = 0xFFFFFFFFu;                   // a lone code-length symbol costs no bits
  if (Q1) {
    if (lane < 18u) {
      const uint32_t code = lane <= 12u ? lane : lane >= 16u ? lane - 3u : lane == 15u ? 0u : lane + 17u;
      const uint32_t nb = (lane == 13u || lane == 14u) ? 5u : 4u;
      cl_len[lane] = (uint8_t)nb;
      cl_bits[lane] = (uint16_t)(dev_bitrev32(code) >> (32u - nb));
    }
    wave_sync();
    // StoreStaticCodeLengthCode: 40 fixed bits
    pfx_put_lanes(bitbuf, bitpos, lane == 0 ? 32u : lane == 1u ? 8u : 0u, lane == 0 ? 0x55555554u : 0xFFu);
  } else {
  // the code-length code: 18 symbols, lengths of at most 5 bits
  uint32_t distinct = 0, only = 0;
  {
    const uint64_t mk = wave_ballot(lane < 18u && small[32u + lane] != 0);
    distinct = (uint32_t)dev_popc64(mk);
    only = mk ? (uint32_t)dev_ctz64(mk) : 0u;
  }
  // (the nested build works below dword 64 of A and C; out / extra / starts sit above)
  uint32_t* histo18 = P + PFX_SMALL + 32u;
  pfx_code_lengths<false>(histo18, 18u, 5u, P);
  pfx_assign_codes(18u, P, cl_len, cl_bits);
  // header: HSKIP, then the lengths of the code-length code in the order of RFC 7932 3.5
  // (1, 2, 3, 4, 0, 5, 17, 6, 16, then 7 .. 15), each with its fixed code
  const uint32_t ord = lane < 9u ? (uint32_t)((0x10344A020C41ull >> (5u * lane)) & 31u) : lane - 2u;
  const uint32_t l_ord = lane < 18u ? cl_len[ord] : 0u;
  uint32_t keep = 18;
  if (distinct > 1u) {
    const uint64_t nz = wave_ballot(lane < 18u && l_ord != 0);
    keep = nz ? 64u - (uint32_t)__builtin_clzll(nz) : 0u;
  }
  const uint32_t l0 = wave_bcast(l_ord, 0), l1 = wave_bcast(l_ord, 1), l2 = wave_bcast(l_ord, 2);
  const uint32_t skip = (l0 == 0 && l1 == 0) ? (l2 == 0 ? 3u : 2u) : 0u;
  {
    // symbols of the fixed code for code lengths 0 .. 5: values / lengths (RFC 7932 3.5)
    const uint32_t fv = (0xF12370u >> (4u * l_ord)) & 0xFu;     // 0, 7, 3, 2, 1, 15
    const uint32_t fl = (0x422342u >> (4u * l_ord)) & 0xFu;     // 2, 4, 3, 2, 2, 4
    const bool on = lane >= skip && lane < keep;
    // HSKIP rides in front of the first stored length
    const bool first = lane == skip;
    uint32_t nb = on ? fl : 0u, val = on ? fv : 0u;
    if (first && on) { val = (val << 2) | skip; nb += 2u; }
    if (skip >= keep) { if (lane == 0) { nb = 2; val = skip; } }
    pfx_put_lanes(bitbuf, bitpos, nb, val);
  }
  if (distinct == 1u) zero_len_code = only;
  }
  for (uint32_t base = 0; base < total; base += 64u) {
    const uint32_t e = base + lane;
    uint32_t nb = 0, val = 0;
    if (e < total) {
      const uint32_t v = out[e];
      nb = v == zero_len_code ? 0u : cl_len[v];
      val = cl_bits[v];
      if (v == 16u) { val |= (uint32_t)extra[e] << nb; nb += 2u; }
      else if (v == 17u) { val |= (uint32_t)extra[e] << nb; nb += 3u; }
    }
    pfx_put_lanes(bitbuf, bitpos, nb, val);
  }
  wave_sync();
}

// One histogram -> lengths + codes (depth8 / bits16: the caller's arrays) and the description of
// the code appended to bitbuf at bitpos: NSYM = 1 / simple / complex form as the reference chooses
// (BuildAndStoreHuffmanTree, brotli_bit_stream.c:349-397; Q1: BrotliBuildAndStoreHuffmanTreeFast,
// :404-573, length limit 14).  `clear`: the buffer is this code's own and is zeroed here, once the
// sort — whose keys overlay it for more than 416 used symbols — is through.
template <bool Q1>
DEV void pfx_build_and_append(const uint32_t* histo, uint32_t n, uint32_t max_bits, uint32_t* P,
                              uint8_t* depth8, uint16_t* bits16, uint32_t* bitbuf, uint32_t& bitpos, bool clear) {
  const uint32_t lane = (uint32_t)wave_lane();
  wave_sync();
  // the first four used symbols, and whether there are more
  uint32_t count = 0, s4[4] = {0, 0, 0, 0};
  for (uint32_t base = 0; base < n && count <= 4u; base += 64u) {
    const uint32_t i = base + lane;
    uint64_t mk = wave_ballot(i < n && histo[i] != 0);
    while (mk != 0 && count <= 4u) {
      const uint32_t b = (uint32_t)dev_ctz64(mk);
      if (count < 4u) s4[count] = base + b;
      ++count;
      mk &= mk - 1ull;
    }
  }
  if (count <= 1u) {
    // one symbol: simple code with NSYM = 1; it costs no bits in the data
    if (clear) { for (uint32_t i = lane; i < 128u; i += 64u) bitbuf[i] = 0; }
    wave_sync();
    if (lane == 0) { depth8[s4[0]] = 0; bits16[s4[0]] = 0; }
    pfx_put_lanes(bitbuf, bitpos, lane == 0 ? 4u + max_bits : 0u, 1u | (s4[0] << 4));
    wave_sync();
    return;
  }
  pfx_code_lengths<Q1>(histo, n, Q1 ? 14u : 15u, P);
  pfx_assign_codes(n, P, depth8, bits16);
  if (clear) { for (uint32_t i = lane; i < 128u; i += 64u) bitbuf[i] = 0; }
  wave_sync();
  if (count <= 4u) {
    // simple code: the symbols in order of their lengths (the order a selection sort that
    // swaps on "shorter" leaves them in, brotli_bit_stream.c:258-266), NSYM - 1, tree select
    const uint8_t* len = (const uint8_t*)(P + PFX_LEN);
    for (uint32_t i = 0; i < count; ++i)
      for (uint32_t j = i + 1u; j < count; ++j)
        if (len[s4[j]] < len[s4[i]]) { const uint32_t t = s4[j]; s4[j] = s4[i]; s4[i] = t; }
    uint32_t nb = 0, val = 0;
    if (lane == 0) { nb = 4; val = 1u | ((count - 1u) << 2); }
    else if (lane <= count) { nb = max_bits; val = lane == 1u ? s4[0] : lane == 2u ? s4[1] : lane == 3u ? s4[2] : s4[3]; }
    else if (lane == 5u && count == 4u) { nb = 1; val = len[s4[0]] == 1 ? 1u : 0u; }
    pfx_put_lanes(bitbuf, bitpos, nb, val);
  } else {
    pfx_store_complex<Q1>(n, P, bitbuf, bitpos);
  }
  wave_sync();
}

// The store kernel's job: one code into its own 512-byte buffer `buf`; returns its bits.
// (Q1: the count-only builder — quality 2's BrotliStoreMetaBlockFast uses it for all three codes.)
template <bool Q1 = false>
DEV uint32_t pfx_build_and_store(const uint32_t* histo, uint32_t n, uint32_t alphabet_size, uint32_t* P,
                                 uint8_t* depth8, uint16_t* bits16, uint8_t* buf) {
  const uint32_t lane = (uint32_t)wave_lane();
  uint32_t* bitbuf = P + PFX_BITBUF;
  uint32_t max_bits = 0, bitpos = 0;
  for (uint32_t c = alphabet_size - 1u; c != 0; c >>= 1) ++max_bits;
  pfx_build_and_append<Q1>(histo, n, max_bits, P, depth8, bits16, bitbuf, bitpos, true);
  for (uint32_t i = lane; i < (bitpos + 31u) / 32u; i += 64u) st32(buf + 4u * i, bitbuf[i]);
  wave_sync();
  return bitpos;
}

#endif  // BROTLI_AMD_CSRC_K_PREFIX_H_
