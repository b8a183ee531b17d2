/* oracle/plan_bench.c — TEST / BENCH INFRASTRUCTURE ONLY.
 *
 * CPU baseline driver: encodes a file with the reference encoder
 * (oracle/_ref/libbrotli_ref.so, loaded with dlopen) using the same partition
 * plan as the GPU run — one independent encoder instance per shard
 * (BROTLI_PARAM_STREAM_OFFSET contract, c/include/brotli/encode.h:231-246) —
 * on T POSIX threads, and prints one JSON line with the wall time.
 *
 *   plan_bench <libbrotli_ref.so> <input file> <quality> <lgwin> <shard_size>
 *              <threads> [size_hint]
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef void* (*create_fn)(void*, void*, void*);
typedef void (*destroy_fn)(void*);
typedef int (*setparam_fn)(void*, int, uint32_t);
typedef int (*stream_fn)(void*, int, size_t*, const uint8_t**, size_t*, uint8_t**, size_t*);
typedef int (*more_fn)(void*);

static create_fn Create;
static destroy_fn Destroy;
static setparam_fn SetParameter;
static stream_fn CompressStream;
static more_fn HasMoreOutput;

static const uint8_t* g_in;
static size_t g_len, g_shard, g_nshards;
static int g_quality, g_lgwin;
static uint32_t g_hint;
static volatile size_t g_next;
static uint64_t* g_sizes;

static void* worker(void* arg) {
  uint8_t* out = NULL;
  size_t cap = 0;
  (void)arg;
  for (;;) {
    size_t k = __sync_fetch_and_add(&g_next, 1);
    size_t off, n, avail_in, avail_out, total = 0;
    const uint8_t* next_in;
    uint8_t* next_out;
    void* st;
    int op;
    if (k >= g_nshards) break;
    off = k * g_shard;
    n = g_len - off < g_shard ? g_len - off : g_shard;
    if (cap < 2 * n + 1024) { cap = 2 * n + 1024; out = (uint8_t*)realloc(out, cap); }
    st = Create(NULL, NULL, NULL);
    SetParameter(st, 1 /* QUALITY */, (uint32_t)g_quality);
    SetParameter(st, 2 /* LGWIN */, (uint32_t)g_lgwin);
    SetParameter(st, 5 /* SIZE_HINT */, g_hint);
    if (off) SetParameter(st, 9 /* STREAM_OFFSET */, off > (1u << 30) ? (1u << 30) : (uint32_t)off);
    avail_in = n; next_in = g_in + off; avail_out = cap; next_out = out;
    op = (off + n == g_len) ? 2 /* FINISH */ : 1 /* FLUSH */;
    do {
      if (!CompressStream(st, op, &avail_in, &next_in, &avail_out, &next_out, &total)) {
        fprintf(stderr, "CompressStream failed on shard %zu\n", k);
        exit(2);
      }
    } while (avail_in || HasMoreOutput(st));
    g_sizes[k] = total;
    Destroy(st);
  }
  free(out);
  return NULL;
}

int main(int argc, char** argv) {
  void* lib;
  FILE* f;
  uint8_t* buf;
  pthread_t* th;
  int threads, i;
  struct timespec t0, t1;
  uint64_t out_total = 0;
  size_t k;
  double dt;
  if (argc < 7) { fprintf(stderr, "usage: see source\n"); return 1; }
  lib = dlopen(argv[1], RTLD_NOW);
  if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
  Create = (create_fn)dlsym(lib, "BrotliEncoderCreateInstance");
  Destroy = (destroy_fn)dlsym(lib, "BrotliEncoderDestroyInstance");
  SetParameter = (setparam_fn)dlsym(lib, "BrotliEncoderSetParameter");
  CompressStream = (stream_fn)dlsym(lib, "BrotliEncoderCompressStream");
  HasMoreOutput = (more_fn)dlsym(lib, "BrotliEncoderHasMoreOutput");
  if (!Create || !Destroy || !SetParameter || !CompressStream || !HasMoreOutput) return 1;
  f = fopen(argv[2], "rb");
  if (!f) { perror("input"); return 1; }
  fseek(f, 0, SEEK_END);
  g_len = (size_t)ftell(f);
  fseek(f, 0, SEEK_SET);
  buf = (uint8_t*)malloc(g_len + 16);
  if (fread(buf, 1, g_len, f) != g_len) return 1;
  fclose(f);
  g_in = buf;
  g_quality = atoi(argv[3]);
  g_lgwin = atoi(argv[4]);
  g_shard = (size_t)strtoull(argv[5], NULL, 10);
  threads = atoi(argv[6]);
  if (g_shard == 0 || g_shard > g_len) g_shard = g_len;
  g_hint = argc > 7 ? (uint32_t)strtoul(argv[7], NULL, 10)
                    : (g_len >= (1u << 30) ? (1u << 30) : (uint32_t)g_len);
  g_nshards = (g_len + g_shard - 1) / g_shard;
  g_sizes = (uint64_t*)calloc(g_nshards, 8);
  th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (i = 0; i < threads; ++i) pthread_create(&th[i], NULL, worker, NULL);
  for (i = 0; i < threads; ++i) pthread_join(th[i], NULL);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  dt = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  for (k = 0; k < g_nshards; ++k) out_total += g_sizes[k];
  printf("{\"bytes\": %zu, \"shards\": %zu, \"threads\": %d, \"seconds\": %.6f, \"MBps\": %.2f, "
         "\"out_bytes\": %llu}\n", g_len, g_nshards, threads, dt, (double)g_len / 1e6 / dt,
         (unsigned long long)out_total);
  return 0;
}
