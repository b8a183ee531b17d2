/* oracle/plan_bench.c — TEST / BENCH INFRASTRUCTURE ONLY.
 *
 * CPU baseline driver: encodes a file with the reference encoder
 * (oracle/_ref/libbrotli_ref.so, loaded with dlopen) using the same partition
 * plan as the GPU run — one independent encoder instance per shard
 * (BROTLI_PARAM_STREAM_OFFSET contract, c/include/brotli/encode.h:231-246) —
 * on T POSIX threads (optionally pinned to a CPU list, BASELINE.md §3.3: one
 * socket), `reps` times, and prints one JSON line: every wall time, the median,
 * and the sha256 of the concatenated output (the bytes the GPU run must equal).
 *
 *   plan_bench <libbrotli_ref.so> <input file> <quality> <lgwin> <shard_size>
 *              <threads> [size_hint [reps [cpu,cpu,...]]]
 *
 * What is tried to make the baseline as fast as the box allows — environment:
 *   PLAN_BENCH_ALLOC=pool   every worker hands the reference's alloc_func / free_func hooks
 *                           (c/include/brotli/encode.h:289-307) a private pool that keeps freed blocks by size: an
 *                           instance per shard allocates the same ~2.7 MiB (hash table, ring buffer, command buffer)
 *                           every time, which glibc serves with mmap + page faults + munmap — and every munmap of a
 *                           64-thread process interrupts the other 63 cores (TLB shootdown);
 *   PLAN_BENCH_ALLOC=arena  glibc tuned instead (M_MMAP_THRESHOLD / M_TRIM_THRESHOLD raised: blocks stay in the
 *                           per-thread arenas);
 *   PLAN_BENCH_PROCS=1      the workers are forked processes (one address space each: no shared mmap lock, no
 *                           cross-core TLB shootdowns), sizes and bytes through shared memory;
 *   the main thread pins itself to the CPU list before it allocates and reads the input, so the input's pages are
 *   first touched on the workers' socket (NUMA-local).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <malloc.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

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
static volatile size_t* g_next;   /* (shared memory: the forked workers take shards from the same counter) */
static uint64_t* g_sizes;
static uint8_t** g_outs;      /* per-shard compressed bytes of the last repetition */
static uint8_t* g_shm_out;    /* PLAN_BENCH_PROCS: shard k's bytes at k * g_shm_stride */
static size_t g_shm_stride;
static int g_cpus[1024], g_ncpus;
static int g_pool, g_procs;

/* ---- PLAN_BENCH_ALLOC=pool: a worker's private allocator behind the reference's hooks ------- */
typedef struct PoolBlock { size_t size; struct PoolBlock* next; } PoolBlock;
typedef struct { PoolBlock* free_list; } Pool;
static void* pool_alloc(void* opaque, size_t size) {
  Pool* P = (Pool*)opaque;
  PoolBlock** pp = &P->free_list;
  PoolBlock* b;
  for (; *pp; pp = &(*pp)->next)
    if ((*pp)->size == size) { b = *pp; *pp = b->next; return (void*)(b + 1); }
  b = (PoolBlock*)malloc(sizeof(PoolBlock) + size);
  if (!b) return NULL;
  b->size = size;
  return (void*)(b + 1);
}
static void pool_free(void* opaque, void* ptr) {
  Pool* P = (Pool*)opaque;
  PoolBlock* b;
  if (!ptr) return;
  b = (PoolBlock*)ptr - 1;
  b->next = P->free_list;
  P->free_list = b;
}
static void pool_release(Pool* P) {
  while (P->free_list) { PoolBlock* b = P->free_list; P->free_list = b->next; free(b); }
}

/* ---- sha256 (FIPS 180-4), for the whole-output parity check --------------- */
typedef struct { uint32_t h[8]; uint8_t buf[64]; uint64_t n; } Sha;
static const uint32_t K256[64] = {
  0x428a2f98,0x71374491,0xb5c0fbcf,0xe9b5dba5,0x3956c25b,0x59f111f1,0x923f82a4,0xab1c5ed5,
  0xd807aa98,0x12835b01,0x243185be,0x550c7dc3,0x72be5d74,0x80deb1fe,0x9bdc06a7,0xc19bf174,
  0xe49b69c1,0xefbe4786,0x0fc19dc6,0x240ca1cc,0x2de92c6f,0x4a7484aa,0x5cb0a9dc,0x76f988da,
  0x983e5152,0xa831c66d,0xb00327c8,0xbf597fc7,0xc6e00bf3,0xd5a79147,0x06ca6351,0x14292967,
  0x27b70a85,0x2e1b2138,0x4d2c6dfc,0x53380d13,0x650a7354,0x766a0abb,0x81c2c92e,0x92722c85,
  0xa2bfe8a1,0xa81a664b,0xc24b8b70,0xc76c51a3,0xd192e819,0xd6990624,0xf40e3585,0x106aa070,
  0x19a4c116,0x1e376c08,0x2748774c,0x34b0bcb5,0x391c0cb3,0x4ed8aa4a,0x5b9cca4f,0x682e6ff3,
  0x748f82ee,0x78a5636f,0x84c87814,0x8cc70208,0x90befffa,0xa4506ceb,0xbef9a3f7,0xc67178f2};
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void sha_block(Sha* s, const uint8_t* p) {
  uint32_t w[64], a[8], t1, t2;
  int i;
  for (i = 0; i < 16; ++i) w[i] = (uint32_t)p[4*i] << 24 | (uint32_t)p[4*i+1] << 16 | (uint32_t)p[4*i+2] << 8 | p[4*i+3];
  for (i = 16; i < 64; ++i)
    w[i] = w[i-16] + (ROR(w[i-15],7) ^ ROR(w[i-15],18) ^ (w[i-15] >> 3)) + w[i-7] +
           (ROR(w[i-2],17) ^ ROR(w[i-2],19) ^ (w[i-2] >> 10));
  memcpy(a, s->h, 32);
  for (i = 0; i < 64; ++i) {
    t1 = a[7] + (ROR(a[4],6) ^ ROR(a[4],11) ^ ROR(a[4],25)) + ((a[4] & a[5]) ^ (~a[4] & a[6])) + K256[i] + w[i];
    t2 = (ROR(a[0],2) ^ ROR(a[0],13) ^ ROR(a[0],22)) + ((a[0] & a[1]) ^ (a[0] & a[2]) ^ (a[1] & a[2]));
    a[7] = a[6]; a[6] = a[5]; a[5] = a[4]; a[4] = a[3] + t1; a[3] = a[2]; a[2] = a[1]; a[1] = a[0]; a[0] = t1 + t2;
  }
  for (i = 0; i < 8; ++i) s->h[i] += a[i];
}
static void sha_init(Sha* s) {
  static const uint32_t h0[8] = {0x6a09e667,0xbb67ae85,0x3c6ef372,0xa54ff53a,0x510e527f,0x9b05688c,0x1f83d9ab,0x5be0cd19};
  memcpy(s->h, h0, 32);
  s->n = 0;
}
static void sha_update(Sha* s, const uint8_t* p, size_t n) {
  size_t fill = (size_t)(s->n & 63);
  s->n += n;
  if (fill) {
    size_t take = 64 - fill < n ? 64 - fill : n;
    memcpy(s->buf + fill, p, take);
    p += take; n -= take;
    if (fill + take < 64) return;
    sha_block(s, s->buf);
  }
  for (; n >= 64; p += 64, n -= 64) sha_block(s, p);
  memcpy(s->buf, p, n);
}
static void sha_final(Sha* s, char hex[65]) {
  uint64_t bits = s->n * 8;
  uint8_t pad[72] = {0x80};
  size_t fill = (size_t)(s->n & 63), npad = (fill < 56 ? 56 : 120) - fill;
  int i;
  for (i = 0; i < 8; ++i) pad[npad + i] = (uint8_t)(bits >> (56 - 8 * i));
  sha_update(s, pad, npad + 8);
  for (i = 0; i < 8; ++i) sprintf(hex + 8 * i, "%08x", s->h[i]);
}

static void* worker(void* arg) {
  uint8_t* out = NULL;
  size_t cap = 0;
  const long tid = (long)arg;
  Pool pool = {NULL};
  if (g_ncpus) {
    cpu_set_t set;
    CPU_ZERO(&set);
    CPU_SET(g_cpus[tid % g_ncpus], &set);
    pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
  }
  for (;;) {
    size_t k = __sync_fetch_and_add(g_next, 1);
    size_t off, n, avail_in, avail_out, total = 0;
    const uint8_t* next_in;
    uint8_t* next_out;
    void* st;
    int op;
    if (k >= g_nshards) break;
    off = k * g_shard;
    n = g_len - off < g_shard ? g_len - off : g_shard;
    if (cap < 2 * n + 1024) { cap = 2 * n + 1024; out = (uint8_t*)realloc(out, cap); }
    st = g_pool ? Create((void*)pool_alloc, (void*)pool_free, &pool) : Create(NULL, NULL, NULL);
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
    /* keep the bytes for the output hash (outside the hot loop of the encoder) */
    if (g_shm_out) memcpy(g_shm_out + k * g_shm_stride, out, total);
    else {
      g_outs[k] = (uint8_t*)realloc(g_outs[k], total ? total : 1);
      memcpy(g_outs[k], out, total);
    }
    Destroy(st);
  }
  free(out);
  pool_release(&pool);
  return NULL;
}

static int cmp_double(const void* a, const void* b) {
  const double x = *(const double*)a, y = *(const double*)b;
  return x < y ? -1 : x > y;
}

int main(int argc, char** argv) {
  void* lib;
  FILE* f;
  uint8_t* buf;
  pthread_t* th;
  int threads, i, reps, r;
  struct timespec t0, t1;
  uint64_t out_total = 0;
  size_t k;
  double times[64], sorted[64], median;
  char hex[65];
  Sha sha;
  const char* alloc_mode = getenv("PLAN_BENCH_ALLOC");
  if (argc < 7) { fprintf(stderr, "usage: see source\n"); return 1; }
  g_pool = alloc_mode && strcmp(alloc_mode, "pool") == 0;
  g_procs = getenv("PLAN_BENCH_PROCS") && atoi(getenv("PLAN_BENCH_PROCS")) != 0;
  if (alloc_mode && strcmp(alloc_mode, "arena") == 0) {
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_TOP_PAD, 64 << 20);
  }
  if (argc > 9) {
    /* the CPU list first: the main thread moves there before it touches the input's pages */
    char* c = argv[9];
    cpu_set_t set;
    CPU_ZERO(&set);
    while (*c && g_ncpus < 1024) {
      g_cpus[g_ncpus] = (int)strtol(c, &c, 10);
      CPU_SET(g_cpus[g_ncpus], &set);
      ++g_ncpus;
      if (*c == ',') ++c;
    }
    if (g_ncpus) sched_setaffinity(0, sizeof(set), &set);
  }
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
  if (threads < 1) threads = 1;
  if (g_shard == 0 || g_shard > g_len) g_shard = g_len;
  g_hint = argc > 7 && atol(argv[7]) > 0 ? (uint32_t)strtoul(argv[7], NULL, 10)
                    : (g_len >= (1u << 30) ? (1u << 30) : (uint32_t)g_len);
  reps = argc > 8 ? atoi(argv[8]) : 1;
  if (reps < 1) reps = 1;
  if (reps > 64) reps = 64;
  g_nshards = (g_len + g_shard - 1) / g_shard;
  if (g_procs) {
    /* counter, sizes and output bytes where forked workers and the parent both see them */
    g_shm_stride = 2 * g_shard + 1024;
    g_next = (volatile size_t*)mmap(NULL, 4096 + g_nshards * 8, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    g_shm_out = (uint8_t*)mmap(NULL, g_nshards * g_shm_stride, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (g_next == MAP_FAILED || g_shm_out == MAP_FAILED) { perror("mmap"); return 1; }
    g_sizes = (uint64_t*)((uint8_t*)g_next + 4096);
  } else {
    g_next = (volatile size_t*)calloc(1, sizeof(size_t));
    g_sizes = (uint64_t*)calloc(g_nshards, 8);
  }
  g_outs = (uint8_t**)calloc(g_nshards, sizeof(uint8_t*));
  th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
  for (r = 0; r < reps; ++r) {
    *g_next = 0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    if (g_procs) {
      for (i = 0; i < threads; ++i) {
        const pid_t pid = fork();
        if (pid < 0) { perror("fork"); return 1; }
        if (pid == 0) { worker((void*)(long)i); _exit(0); }
      }
      for (i = 0; i < threads; ++i) {
        int st = 0;
        if (wait(&st) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) { fprintf(stderr, "a worker process failed\n"); return 2; }
      }
    } else {
      for (i = 0; i < threads; ++i) pthread_create(&th[i], NULL, worker, (void*)(long)i);
      for (i = 0; i < threads; ++i) pthread_join(th[i], NULL);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    times[r] = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  }
  memcpy(sorted, times, sizeof(double) * (size_t)reps);
  qsort(sorted, (size_t)reps, sizeof(double), cmp_double);
  median = (reps & 1) ? sorted[reps / 2] : 0.5 * (sorted[reps / 2 - 1] + sorted[reps / 2]);
  sha_init(&sha);
  for (k = 0; k < g_nshards; ++k) {
    out_total += g_sizes[k];
    sha_update(&sha, g_shm_out ? g_shm_out + k * g_shm_stride : g_outs[k], g_sizes[k]);
  }
  sha_final(&sha, hex);
  printf("{\"bytes\": %zu, \"shards\": %zu, \"threads\": %d, \"pinned_cpus\": %d, \"alloc\": \"%s\", \"workers\": \"%s\", "
         "\"reps\": %d, \"seconds\": %.6f, \"MBps\": %.2f, \"seconds_all\": [", g_len, g_nshards, threads, g_ncpus,
         alloc_mode ? alloc_mode : "malloc", g_procs ? "processes" : "threads",
         reps, median, (double)g_len / 1e6 / median);
  for (r = 0; r < reps; ++r) printf("%s%.6f", r ? ", " : "", times[r]);
  printf("], \"out_bytes\": %llu, \"sha256\": \"%s\"}\n", (unsigned long long)out_total, hex);
  return 0;
}
