/* tools/gen_transforms.c — one-shot generator for brotli_amd/data/brotli_transforms.bin.
 *
 * Run only in the build container (needs /root/reference); linked with the reference's
 * c/common/transform.c, it dumps the *format data* of RFC 7932 Appendix B — the 121 word
 * transforms (prefix, elementary operation, suffix) the decoder (k_decode.h) applies to static
 * dictionary words — in a form of our own; no reference source text is copied.
 *
 * Blob (little endian):  magic "BRTT", u32 version = 1, u32 n (121), u32 text_size,
 *   n records { u16 prefix_off; u8 prefix_len; u8 op; u16 suffix_off; u8 suffix_len; u8 param },
 *   u8 text[text_size] (the affix strings, not terminated), padded to 4.
 * op: 0 identity, 1..9 omit last n, 10 uppercase first, 11 uppercase all, 12..20 omit first n - 11
 * (c/common/transform.h:18-44).
 *
 *   gcc -O2 -I/root/reference/c/include tools/gen_transforms.c /root/reference/c/common/transform.c \
 *       -o /tmp/gen_transforms && /tmp/gen_transforms brotli_amd/data/brotli_transforms.bin
 */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../reference/c/common/transform.h"

int main(int argc, char** argv) {
  const BrotliTransforms* t = BrotliGetTransforms();
  uint8_t text[4096];
  uint8_t rec[256][8];
  uint32_t n = t->num_transforms, text_size = 0, i, u;
  FILE* f;
  if (argc < 2 || n > 256) return 2;
  for (i = 0; i < n; ++i) {
    const uint8_t* pre = BROTLI_TRANSFORM_PREFIX(t, i);
    const uint8_t* suf = BROTLI_TRANSFORM_SUFFIX(t, i);
    const uint8_t op = BROTLI_TRANSFORM_TYPE(t, i);
    uint16_t po = (uint16_t)text_size, so;
    if (op > 20) return 4;   /* the RFC's set has no shift transforms */
    memcpy(text + text_size, pre + 1, pre[0]); text_size += pre[0];
    so = (uint16_t)text_size;
    memcpy(text + text_size, suf + 1, suf[0]); text_size += suf[0];
    memcpy(&rec[i][0], &po, 2); rec[i][2] = pre[0]; rec[i][3] = op;
    memcpy(&rec[i][4], &so, 2); rec[i][6] = suf[0]; rec[i][7] = 0;
  }
  f = fopen(argv[1], "wb");
  if (!f) return 3;
  fwrite("BRTT", 1, 4, f);
  u = 1; fwrite(&u, 4, 1, f);
  fwrite(&n, 4, 1, f);
  fwrite(&text_size, 4, 1, f);
  fwrite(rec, 8, n, f);
  fwrite(text, 1, text_size, f);
  for (i = text_size; i & 3; ++i) fputc(0, f);
  fclose(f);
  printf("ok n=%u text=%u\n", n, text_size);
  return 0;
}
