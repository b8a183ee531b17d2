/* tools/gen_tables.c — one-shot generator for brotli_amd/data/brotli_tables.bin.
 *
 * Run only in the build container (needs /root/reference).  It is linked with
 * the reference's own translation units and dumps the *format data* the
 * encoder needs as one binary blob; no reference source text is copied:
 *   - RFC 7932 Appendix A static dictionary (122 784 B) and its per-length
 *     layout (c/common/dictionary.h:18-41, data in c/common/dictionary.bin),
 *   - RFC 7932 section 7.1 literal context LUT (c/common/context.c:9,
 *     2048 B),
 *   - the encoder's static-dictionary hash (kStaticDictionaryHashWords /
 *     kStaticDictionaryHashLengths, c/enc/dictionary_hash.h:25-38); this is a
 *     tuned artefact (c/enc/dictionary_hash.c:24-129 needs a 1688-byte
 *     "frozen" bitmap), so it cannot be re-derived from the dictionary alone.
 * It also checks that kBrotliLog2Table (c/enc/fast_log.c:14) equals
 * (double)(float)log2(i), which is how the product rebuilds that table.
 *
 * Blob layout (little endian):
 *   magic "BRTB" u32 version=1
 *   u8  context_lut[2048]
 *   u8  size_bits_by_length[32]
 *   u32 offsets_by_length[32]
 *   u32 dict_size; u8 dict[dict_size]        (padded to 4)
 *   u16 hash_words[32768]
 *   u8  hash_lengths[32768]
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../reference/c/common/context.h"
#include "../../reference/c/common/dictionary.h"
#include "../../reference/c/enc/dictionary_hash.h"
#include "../../reference/c/enc/fast_log.h"

int main(int argc, char** argv) {
  const BrotliDictionary* d = BrotliGetDictionary();
  FILE* f;
  uint32_t u;
  int i, bad = 0;
  if (argc < 2) return 2;
  for (i = 0; i < 256; ++i) {
    double mine = i == 0 ? 0.0 : (double)(float)log2((double)i);
    if (mine != kBrotliLog2Table[i]) { ++bad; printf("log2 table mismatch at %d\n", i); }
  }
  if (bad) return 1;
  f = fopen(argv[1], "wb");
  if (!f) return 3;
  fwrite("BRTB", 1, 4, f);
  u = 1; fwrite(&u, 4, 1, f);
  fwrite(_kBrotliContextLookupTable, 1, 2048, f);
  fwrite(d->size_bits_by_length, 1, 32, f);
  fwrite(d->offsets_by_length, 4, 32, f);
  u = (uint32_t)d->data_size; fwrite(&u, 4, 1, f);
  fwrite(d->data, 1, d->data_size, f);
  for (i = (int)d->data_size; i & 3; ++i) fputc(0, f);
  fwrite(kStaticDictionaryHashWords, 2, 32768, f);
  fwrite(kStaticDictionaryHashLengths, 1, 32768, f);
  fclose(f);
  printf("ok dict_size=%u\n", (unsigned)d->data_size);
  return 0;
}
