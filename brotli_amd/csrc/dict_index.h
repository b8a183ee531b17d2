/* brotli_amd/csrc/dict_index.h — host side of BrotliEncoderPrepareDictionary for raw (LZ77 prefix)
 * dictionaries: the index the device lookup (k_dict.h) walks.  Plain C, included by encode_abi.c.
 *
 * Reference behaviour being reproduced (c/enc/compound_dictionary.c:13-173): every position i with
 * i + 7 < size is hashed over 40 bits of the eight bytes there into 2^bucket_bits keys
 * (bucket_bits = 17, one more for every doubling of the dictionary past 2 MiB, at most 22); a key
 * remembers its NEWEST 32 positions, newest first.  (The reference also shrinks a "slot"'s limit
 * when a 16-bit offset would overflow; with its parameters a slot holds at most 1024 * 32 items, so
 * that branch is dead and is not restated.)
 *
 * Built here as CSR by a counting pass and one backward sweep: starts[key] .. starts[key + 1] index
 * into items[]; positions are appended from the last one down, a key that has its 32 is full. */
#ifndef BROTLI_AMD_CSRC_DICT_INDEX_H_
#define BROTLI_AMD_CSRC_DICT_INDEX_H_

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define DICT_INDEX_MAGIC 0xA3D1C7E5u
#define DICT_INDEX_LIMIT 32u
#define DICT_INDEX_MAX_RAW 0x7FFFFFFFu   /* SHARED_BROTLI_MAX_RAW_DICT_SIZE, c/include/brotli/shared_dictionary.h */

typedef struct DictIndex {
  uint32_t magic;
  uint32_t bucket_bits;
  uint32_t source_size;
  uint32_t num_items;
  const uint8_t* source;     /* the caller's bytes: referenced, not copied (encode.h:331-335) */
  uint32_t* starts;          /* [(1 << bucket_bits) + 1] */
  uint32_t* items;           /* [num_items] */
  void* (*alloc)(void* opaque, size_t size);
  void (*free_)(void* opaque, void* p);
  void* opaque;
} DictIndex;

static void* dict_index_alloc(DictIndex* d, size_t n) { return d->alloc ? d->alloc(d->opaque, n) : malloc(n); }
static void dict_index_free(DictIndex* d, void* p) { if (!p) return; if (d->free_) d->free_(d->opaque, p); else free(p); }

static uint32_t dict_index_key(const uint8_t* p, uint32_t bucket_bits) {
  uint64_t x;
  memcpy(&x, p, 8);   /* little endian host */
  return (uint32_t)(((x & 0xFFFFFFFFFFull) * 0x1FE35A7BD3579BD3ull) >> (64u - bucket_bits));
}

/* Returns 0 on allocation failure; `d` must carry alloc / free_ / opaque already. */
static int dict_index_build(DictIndex* d, const uint8_t* source, size_t size) {
  uint32_t bucket_bits = 17, nkeys, i, k, run;
  size_t volume = (size_t)16 << bucket_bits;
  uint8_t* fill;
  while (volume < size && bucket_bits < 22) { ++bucket_bits; volume <<= 1; }
  nkeys = 1u << bucket_bits;
  d->magic = DICT_INDEX_MAGIC;
  d->bucket_bits = bucket_bits;
  d->source_size = (uint32_t)size;
  d->source = source;
  d->starts = (uint32_t*)dict_index_alloc(d, ((size_t)nkeys + 1) * 4);
  fill = (uint8_t*)dict_index_alloc(d, nkeys);
  if (!d->starts || !fill) { dict_index_free(d, d->starts); dict_index_free(d, fill); d->starts = NULL; return 0; }
  memset(d->starts, 0, ((size_t)nkeys + 1) * 4);
  for (i = 0; (size_t)i + 7 < size; ++i) {
    uint32_t* c = &d->starts[dict_index_key(source + i, bucket_bits) + 1];
    if (*c < DICT_INDEX_LIMIT) ++*c;
  }
  for (k = 0, run = 0; k < nkeys; ++k) { const uint32_t c = d->starts[k + 1]; d->starts[k] = run; run += c; }
  d->starts[nkeys] = run;
  d->num_items = run;
  d->items = (uint32_t*)dict_index_alloc(d, ((size_t)run + 1) * 4);
  if (!d->items) { dict_index_free(d, d->starts); dict_index_free(d, fill); d->starts = NULL; return 0; }
  memset(fill, 0, nkeys);
  for (i = size >= 8 ? (uint32_t)(size - 7) : 0; i-- > 0;) {
    const uint32_t key = dict_index_key(source + i, bucket_bits);
    if (fill[key] < DICT_INDEX_LIMIT) d->items[d->starts[key] + fill[key]++] = i;
  }
  dict_index_free(d, fill);
  return 1;
}

static void dict_index_release(DictIndex* d) {
  dict_index_free(d, d->starts);
  dict_index_free(d, d->items);
  d->starts = d->items = NULL;
  d->magic = 0;
}

#endif  /* BROTLI_AMD_CSRC_DICT_INDEX_H_ */
