// brotli_amd/csrc/hip_layer.hip — the HIP C-ABI layer (include/brotli_amd_hip.h):
// context / workspace management, kernel launches on the context's stream,
// HIP-event timing.  Written for gfx950 only.
#include <hip/hip_runtime.h>

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <atomic>
#include <algorithm>
#include <chrono>
#include <string>
#include <thread>
#include <mutex>
#include <vector>

#include "../../include/brotli_amd_hip.h"
#include "host_plan.h"
#include "kernels.h"

// The BROTLI_AMD_* experiment knobs of this layer, read when a context is created (and by brotli_amd_refresh_env, for
// tools and tests that flip one between jobs) instead of with getenv() on the launch path of every job: a library that
// promises re-entrancy must not race a caller's setenv() there.
struct EnvKnobs {
  bool plain_copy = false, ix_debug = false, index_only = false, tile_log = false;
  int copy_threads = 0;          // 0: the default (4)
  int indexed = -1;              // -1: not set
  uint32_t ix_bpw = 0;           // 0: the planner's choice
  int tile_kb = -1, tile_warm = -1, cgroups = -1, sweep_groups = 0, decode_variant = -1;
  long wide_kb = 0;              // 0: not set
  int wide = -1;
  int wide_k = 0;
  int tile_log_level = 0;
};
static EnvKnobs g_env;
static std::mutex g_env_mutex;
static int env_int(const char* name, int unset) { const char* e = getenv(name); return e ? atoi(e) : unset; }
static void env_read() {
  std::lock_guard<std::mutex> lock(g_env_mutex);
  EnvKnobs k;
  k.plain_copy = getenv("BROTLI_AMD_PLAIN_COPY") != nullptr;
  k.ix_debug = getenv("BROTLI_AMD_IX_DEBUG") != nullptr;
  k.index_only = getenv("BROTLI_AMD_INDEX_ONLY") != nullptr;
  k.tile_log = getenv("BROTLI_AMD_TILE_LOG") != nullptr;
  k.tile_log_level = env_int("BROTLI_AMD_TILE_LOG", 0);
  k.copy_threads = env_int("BROTLI_AMD_COPY_THREADS", 0);
  k.indexed = env_int("BROTLI_AMD_INDEXED", -1);
  k.ix_bpw = (uint32_t)env_int("BROTLI_AMD_IX_BPW", 0);
  k.tile_kb = env_int("BROTLI_AMD_TILE_KB", -1);
  k.tile_warm = env_int("BROTLI_AMD_TILE_WARM", -1);
  k.cgroups = env_int("BROTLI_AMD_CGROUPS", -1);
  k.sweep_groups = env_int("BROTLI_AMD_SWEEP_GROUPS", 0);
  k.decode_variant = env_int("BROTLI_AMD_DECODE_VARIANT", -1);
  if (const char* e = getenv("BROTLI_AMD_WIDE_KB")) k.wide_kb = atol(e);
  k.wide = env_int("BROTLI_AMD_WIDE", -1);
  k.wide_k = env_int("BROTLI_AMD_WIDE_K", 0);
  g_env = k;
  // Callers that compress many buffers at once (a server: one encoder instance per thread, each with a context and
  // a HIP stream of its own) are limited by the runtime's hardware queues, not by the device: the HIP runtime maps all
  // streams of a process onto GPU_MAX_HW_QUEUES (default 4) queues, and a job of a few MiB is a string of small,
  // latency-bound kernels — sixteen 4 MiB calls in flight gave 0.65 GB/s with 4 queues and 2.2 GB/s with 32
  // (profiles/r06_s_*).  Asked for here, before the first HIP call of the process where this library makes it; a value
  // the user set stays, BROTLI_AMD_HW_QUEUES=0 leaves the variable alone.  (A process that initialised HIP earlier —
  // torch — has made its choice already.)
  {
    const int q = env_int("BROTLI_AMD_HW_QUEUES", 32);
    if (q > 0) { char v[16]; snprintf(v, sizeof(v), "%d", q); setenv("GPU_MAX_HW_QUEUES", v, 0); }
  }
}

struct BrotliAmdCtx {
  int device = 0;
  int num_cus = 256;
  hipStream_t stream = nullptr;
  HostTables ht;
  uint8_t* d_lut = nullptr;
  uint8_t* d_dict = nullptr;
  uint16_t* d_hash_words = nullptr;
  uint8_t* d_hash_lengths = nullptr;
  double* d_log2 = nullptr;
  uint32_t log2_n = 0;
  DeviceTables* d_T = nullptr;
  uint8_t* d_ws = nullptr;
  uint64_t ws_cap = 0;
  // Hash tables live in their own allocation.
  // Hash tables live in allocations of their own, at most TABLE_CHUNK bytes each (one
  // 128 GiB hipMalloc for 4096 quality-9 tables fails where eight 16 GiB ones succeed).
  std::vector<uint8_t*> d_table_chunks;
  uint64_t chunk_shards = 0, chunk_shard_bytes = 0;
  uint64_t ix_region_bytes = 0;     // indexed job being planned: bytes of one shard's index region

  ShardDesc* d_shards = nullptr;
  ShardState* d_states = nullptr;
  uint64_t* d_scan = nullptr;       // nshards + 1 output offsets
  uint32_t* d_counters = nullptr;   // [0] shards not done, [1] shards in error, [2..4] tiled jobs (k_tile.h)
  uint64_t shard_cap = 0;
  // batches of an indexed job: the index kernels of batch b + 1 run beside the chain of batch b and the
  // build / store of batch b - 1 (run_batches)
  TileDesc* d_tiles = nullptr;      // tiled jobs (JOB_FLAG_TILED): the tile table and the tiles' records
  TileRec* d_trecs = nullptr;
  uint64_t tile_cap = 0;
  // a tiled stream (JOB_FLAG_STREAMT): index chunks, the meta-blocks as shards, their bit offsets
  ShardDesc* d_chunks = nullptr;
  ShardDesc* d_mdesc = nullptr;
  ShardState* d_mstate = nullptr;
  uint64_t* d_moff = nullptr;
  bool tail_fix = false;              // the last stream job: stream_tail_fix (host_plan.h) applies to its output
  bool from_host = false;             // brotli_amd_encode_device is being called by brotli_amd_encode_host
  uint64_t tail_bit = 0, tail_total_bits = 0;
  uint64_t chunk_cap = 0, mb_cap = 0;
  uint8_t* d_stage_in = nullptr;    // encode_host staging
  uint8_t* d_stage_out = nullptr;
  uint64_t stage_in_cap = 0, stage_out_cap = 0;
  // quality 1 (k_fast.h)
  FastFrag* d_ffrags = nullptr;
  FastBlock* d_fblocks = nullptr;
  FastBlockState* d_fbstate = nullptr;
  FastFragState* d_ffstate = nullptr;
  uint64_t* d_fresult = nullptr;
  uint64_t ffrag_cap = 0, fblock_cap = 0;
  hipEvent_t ev[8] = {};
  hipEvent_t ev_ix = nullptr, ev_ixb = nullptr;
  // attached dictionaries of the jobs run on this context (k_dict.h; brotli_amd_ctx_set_dictionary)
  std::vector<void*> dict_allocs;
  CompoundDict* d_cd = nullptr;
  // decoder (k_decode.h): word transforms, per-piece arenas, piece descriptors / results
  HostTransforms htr;
  std::string tables_path;
  DecTransform* d_transforms = nullptr;
  uint8_t* d_transform_text = nullptr;
  uint32_t* d_dec_arena = nullptr;
  uint64_t dec_arena_cap = 0;        // dwords
  DecPiece* d_dec_pieces = nullptr;
  DecResult* d_dec_results = nullptr;
  uint64_t dec_piece_cap = 0;
  // host <-> device copies of encode_host: copy lanes, each a stream and two pinned chunks
  struct CopyLane { hipStream_t s = nullptr; uint8_t* pin[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr}; };
  std::vector<CopyLane> lanes;
  std::string err;
};

// A large copy between a pageable host buffer and device memory: the runtime's own path stages
// through one pinned buffer on the calling thread (about 5 GB/s on the MI355X host, whatever the
// number of callers); here T threads each move every T-th chunk through pinned chunks of their own
// — the host-side memcpy of one chunk overlaps the DMA of the other.  Blocking.
static const uint64_t PIN_CHUNK = 8ull << 20;
static bool big_copy(BrotliAmdCtx* c, uint8_t* dst, const uint8_t* src, uint64_t len, bool to_device) {
  if (len == 0) return true;
  if (len < 2 * PIN_CHUNK || g_env.plain_copy)
    return hipMemcpy(dst, src, len, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost) == hipSuccess;
  if (c->lanes.empty()) {
    int T = 4;
    if (g_env.copy_threads >= 1 && g_env.copy_threads <= 16) T = g_env.copy_threads;
    std::vector<BrotliAmdCtx::CopyLane> lanes((size_t)T);
    bool ok = true;
    for (auto& l : lanes) {
      ok = ok && hipStreamCreateWithFlags(&l.s, hipStreamNonBlocking) == hipSuccess;
      for (int k = 0; k < 2; ++k) {
        ok = ok && hipHostMalloc((void**)&l.pin[k], PIN_CHUNK, hipHostMallocDefault) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&l.ev[k], hipEventDisableTiming) == hipSuccess;
      }
    }
    if (!ok) {
      for (auto& l : lanes) {
        for (int k = 0; k < 2; ++k) { if (l.pin[k]) (void)hipHostFree(l.pin[k]); if (l.ev[k]) (void)hipEventDestroy(l.ev[k]); }
        if (l.s) (void)hipStreamDestroy(l.s);
      }
      (void)hipGetLastError();
      return hipMemcpy(dst, src, len, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost) == hipSuccess;
    }
    c->lanes.swap(lanes);
  }
  const uint64_t T = c->lanes.size(), nchunks = (len + PIN_CHUNK - 1) / PIN_CHUNK;
  std::atomic<int> failed{0};
  std::vector<std::thread> th;
  for (uint64_t j = 0; j < T; ++j) {
    th.emplace_back([&, j]() {
      BrotliAmdCtx::CopyLane& l = c->lanes[j];
      if (hipSetDevice(c->device) != hipSuccess) { failed = 1; return; }
      uint64_t it = 0, prev_off = 0, prev_m = 0;
      for (uint64_t i = j; i < nchunks; i += T, ++it) {
        const uint64_t off = i * PIN_CHUNK, m = off + PIN_CHUNK <= len ? PIN_CHUNK : len - off;
        const int k = (int)(it & 1u);
        if (to_device) {
          if (it >= 2 && hipEventSynchronize(l.ev[k]) != hipSuccess) failed = 1;
          memcpy(l.pin[k], src + off, m);
          if (hipMemcpyAsync(dst + off, l.pin[k], m, hipMemcpyHostToDevice, l.s) != hipSuccess) failed = 1;
          if (hipEventRecord(l.ev[k], l.s) != hipSuccess) failed = 1;
        } else {
          if (hipMemcpyAsync(l.pin[k], src + off, m, hipMemcpyDeviceToHost, l.s) != hipSuccess) failed = 1;
          if (hipEventRecord(l.ev[k], l.s) != hipSuccess) failed = 1;
          if (it >= 1) {
            if (hipEventSynchronize(l.ev[k ^ 1]) != hipSuccess) failed = 1;
            memcpy(dst + prev_off, l.pin[k ^ 1], prev_m);
          }
          prev_off = off; prev_m = m;
        }
      }
      if (hipStreamSynchronize(l.s) != hipSuccess) failed = 1;
      if (!to_device && it >= 1) memcpy(dst + prev_off, l.pin[(it - 1) & 1u], prev_m);
    });
  }
  for (auto& t : th) t.join();
  return failed == 0;
}

namespace {

bool fail(BrotliAmdCtx* c, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  c->err = buf;
  return false;
}

#define HIP_OK(c, expr)                                                        \
  do {                                                                         \
    hipError_t e_ = (expr);                                                    \
    if (e_ != hipSuccess) {                                                    \
      fail((c), "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return false;                                                            \
    }                                                                          \
  } while (0)

template <class T>
bool dev_upload(BrotliAmdCtx* c, T** dst, const void* src, size_t bytes) {
  HIP_OK(c, hipMalloc((void**)dst, bytes ? bytes : 1));
  HIP_OK(c, hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
  return true;
}

bool ensure_log2(BrotliAmdCtx* c, uint32_t n) {
  if (n <= c->log2_n) return true;
  std::vector<double> lut;
  host_log2_lut(n, &lut);
  if (c->d_log2) HIP_OK(c, hipFree(c->d_log2));
  c->d_log2 = nullptr;
  HIP_OK(c, hipMalloc((void**)&c->d_log2, (size_t)n * sizeof(double)));
  HIP_OK(c, hipMemcpy(c->d_log2, lut.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice));
  c->log2_n = n;
  DeviceTables T;
  memset(&T, 0, sizeof(T));
  T.context_lut = c->d_lut + (2 << 9);  // CONTEXT_UTF8, c/common/context.h:104
  T.dict = c->d_dict;
  T.dict_hash_words = c->d_hash_words;
  T.dict_hash_lengths = c->d_hash_lengths;
  T.log2_lut = c->d_log2;
  memcpy(T.dict_offsets_by_length, c->ht.offsets_by_length, sizeof(T.dict_offsets_by_length));
  memcpy(T.dict_size_bits_by_length, c->ht.size_bits_by_length, sizeof(T.dict_size_bits_by_length));
  if (!c->d_T) HIP_OK(c, hipMalloc((void**)&c->d_T, sizeof(DeviceTables)));
  HIP_OK(c, hipMemcpy(c->d_T, &T, sizeof(T), hipMemcpyHostToDevice));
  return true;
}

// Points every shard at its table (own allocation, 128 B * 2^bucket_bits per
// shard; cleared by k_init at the start of every job).
bool prepare_tables(BrotliAmdCtx* c, JobPlan* plan) {
  const uint64_t TABLE_CHUNK = 16ull << 30;
  const bool indexed = (plan->J.flags & JOB_FLAG_INDEXED) != 0;
  const uint64_t tbytes = indexed ? c->ix_region_bytes : (uint64_t)plan->J.rec_bytes << plan->J.bucket_bits;
  const uint64_t nbytes = ((plan->J.flags & (JOB_FLAG_DEEP | JOB_FLAG_QUICK)) == JOB_FLAG_DEEP) ? ((uint64_t)2 << plan->J.bucket_bits) : 0;
  const uint64_t per = tbytes + nbytes;            // records, then the counters of the same shard
                                                   // (indexed job: the shard's index region instead)
  const uint64_t n = plan->shards.size();
  uint64_t per_chunk = TABLE_CHUNK / per;
  if (per_chunk < 1) per_chunk = 1;
  if (per_chunk > n) per_chunk = n;
  const uint64_t nchunks = (n + per_chunk - 1) / per_chunk;
  const bool fits = c->chunk_shard_bytes == per && c->chunk_shards >= per_chunk &&
                    c->d_table_chunks.size() >= nchunks;
  if (!fits) {
    for (uint8_t* p : c->d_table_chunks) if (p) HIP_OK(c, hipFree(p));
    c->d_table_chunks.clear();
    c->chunk_shards = per_chunk;
    c->chunk_shard_bytes = per;
    for (uint64_t k = 0; k < nchunks; ++k) {
      uint8_t* p = nullptr;
      HIP_OK(c, hipMalloc((void**)&p, per_chunk * per));
      c->d_table_chunks.push_back(p);
    }
  }
  for (uint64_t k = 0; k < n; ++k) {
    uint8_t* base = c->d_table_chunks[k / c->chunk_shards] + (k % c->chunk_shards) * per;
    plan->shards[k].table_off = (uint64_t)(base - c->d_ws);            // ws + off (mod 2^64)
    plan->shards[k].num_off = (uint64_t)(base + tbytes - c->d_ws);
    if (indexed) plan->shards[k].ix_off = plan->shards[k].table_off;
  }
  return true;
}

bool ensure_ws(BrotliAmdCtx* c, uint64_t ws_bytes, uint64_t nshards) {
  if (ws_bytes > c->ws_cap) {
    if (c->d_ws) HIP_OK(c, hipFree(c->d_ws));
    c->d_ws = nullptr;
    c->ws_cap = 0;
    HIP_OK(c, hipMalloc((void**)&c->d_ws, ws_bytes));
    c->ws_cap = ws_bytes;
  }
  if (nshards > c->shard_cap) {
    if (c->d_shards) HIP_OK(c, hipFree(c->d_shards));
    if (c->d_states) HIP_OK(c, hipFree(c->d_states));
    if (c->d_scan) HIP_OK(c, hipFree(c->d_scan));
    c->d_shards = nullptr; c->d_states = nullptr; c->d_scan = nullptr;
    c->shard_cap = 0;
    HIP_OK(c, hipMalloc((void**)&c->d_shards, nshards * sizeof(ShardDesc)));
    HIP_OK(c, hipMalloc((void**)&c->d_states, nshards * sizeof(ShardState)));
    HIP_OK(c, hipMalloc((void**)&c->d_scan, (nshards + 1) * sizeof(uint64_t)));
    c->shard_cap = nshards;
  }
  if (!c->d_counters) HIP_OK(c, hipMalloc((void**)&c->d_counters, 16 * sizeof(uint32_t)));
  return true;
}

bool ensure_tiles(BrotliAmdCtx* c, uint64_t ntiles) {
  if (ntiles > c->tile_cap) {
    if (c->d_tiles) HIP_OK(c, hipFree(c->d_tiles));
    if (c->d_trecs) HIP_OK(c, hipFree(c->d_trecs));
    c->d_tiles = nullptr; c->d_trecs = nullptr; c->tile_cap = 0;
    HIP_OK(c, hipMalloc((void**)&c->d_tiles, ntiles * sizeof(TileDesc)));
    HIP_OK(c, hipMalloc((void**)&c->d_trecs, ntiles * sizeof(TileRec)));
    c->tile_cap = ntiles;
  }
  return true;
}

// Copies the chunks of a compound dictionary (bytes + index, k_dict.h) to the device; `allocs` owns the memory.
int upload_dictionary(BrotliAmdCtx* c, const BrotliAmdDictChunk* chunks, uint32_t nchunks,
                      std::vector<void*>* allocs, CompoundDict** d_cd) {
  if (nchunks > DICT_MAX_CHUNKS) { fail(c, "more than 15 dictionary chunks"); return BROTLI_AMD_UNSUPPORTED; }
  if (hipStreamSynchronize(c->stream) != hipSuccess) { fail(c, "stream sync failed"); return BROTLI_AMD_ERROR; }
  for (void* p : *allocs) (void)hipFree(p);
  allocs->clear();
  *d_cd = nullptr;
  if (nchunks == 0) return BROTLI_AMD_OK;
  CompoundDict cd;
  memset(&cd, 0, sizeof(cd));
  auto upload = [&](const void* src, uint64_t bytes, uint64_t slack) -> void* {
    void* d = nullptr;
    if (hipMalloc(&d, bytes + slack + 16) != hipSuccess) return nullptr;
    allocs->push_back(d);
    if (bytes && hipMemcpy(d, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    if (slack && hipMemset((uint8_t*)d + bytes, 0, slack) != hipSuccess) return nullptr;
    return d;
  };
  uint64_t total = 0;
  for (uint32_t k = 0; k < nchunks; ++k) {
    const BrotliAmdDictChunk& h = chunks[k];
    if (h.bucket_bits < 17 || h.bucket_bits > 22 || total + h.source_size > 0x7FFFFFFFull) {
      fail(c, "bad dictionary chunk");
      return BROTLI_AMD_UNSUPPORTED;
    }
    const uint64_t nkeys = 1ull << h.bucket_bits;
    DictChunk& g = cd.chunks[k];
    g.source = (const uint8_t*)upload(h.source, h.source_size, DICT_SOURCE_SLACK);
    g.starts = (const uint32_t*)upload(h.starts, (nkeys + 1) * 4, 0);
    g.items = (const uint32_t*)upload(h.items, (uint64_t)h.starts[nkeys] * 4, 0);
    if (!g.source || !g.starts || !g.items) { fail(c, "dictionary upload failed"); return BROTLI_AMD_ERROR; }
    g.source_size = h.source_size;
    g.bucket_bits = h.bucket_bits;
    g.offset = (uint32_t)total;
    total += h.source_size;
  }
  cd.num_chunks = nchunks;
  cd.total_size = (uint32_t)total;
  *d_cd = (CompoundDict*)upload(&cd, sizeof(cd), 0);
  if (!*d_cd) { fail(c, "dictionary upload failed"); return BROTLI_AMD_ERROR; }
  return BROTLI_AMD_OK;
}

int plan_from_params(BrotliAmdCtx* c, uint64_t len, const BrotliAmdJobParams* p, JobPlan* plan) {
  if (len == 0) { fail(c, "empty job"); return BROTLI_AMD_UNSUPPORTED; }
  if (!plan_job(len, p->quality, p->lgwin, p->size_hint, p->shard_size, p->stream_base,
                p->is_last != 0, plan, /*tables_in_ws=*/false, (int)((p->flags >> BROTLI_AMD_FLAG_LGBLOCK_SHIFT) & 31u))) {
    fail(c, "parameters outside the GPU path (quality %d lgwin %d)", p->quality, p->lgwin);
    return BROTLI_AMD_UNSUPPORTED;
  }
  uint32_t too_long = 0;
  // with dictionaries attached to the context every shard looks them up after each search: at quality 5 that lives in
  // the hash-table kernel with four shards per wave (k_parse4.h, compound_lookup16: round 6 — the one-shard-per-wave
  // k_parse did 0.4 GB/s) for shards that fit the window, not in the indexed parse; at the other qualities in the
  // one-shard-per-wave kernels (k_parse_deep / k_parse_quick)
  const uint32_t api_flags = p->flags | (c->d_cd ? (uint32_t)BROTLI_AMD_FLAG_NO_INDEX : 0u);
  if (!plan_choose_kernels(plan, api_flags, c->num_cus, &too_long)) {
    fail(c, "quality %d needs shards of at most %u bytes", plan->J.quality, too_long);
    return BROTLI_AMD_UNSUPPORTED;
  }
  if (c->d_cd) plan->J.flags &= ~(uint32_t)JOB_FLAG_DUO;      // (the scout groups of k_parse4 do not look dictionaries up)
  // Quality 5 on shards that fit the window: the position index + table-free chain.
  c->ix_region_bytes = 0;
  if ((plan->J.flags & JOB_FLAG_QUAD) && !(api_flags & BROTLI_AMD_FLAG_NO_INDEX)) {
    if (g_env.indexed != 0) {
      c->ix_region_bytes = plan_add_index(plan, /*ix_in_ws=*/false);
      {                                                             // experiment knob: 1, 2, 4, 8
        const uint32_t v = g_env.ix_bpw;
        if (v == 1 || v == 2 || v == 4 || v == 8) plan->J.ix_bpw = v;
      }
      plan->J.flags &= ~(uint32_t)JOB_FLAG_DUO;
      // Shards longer than a tile: their chain runs tile by tile, all tiles at once (k_chain.h, k_tile.h).
      // BROTLI_AMD_TILE_KB: KiB per tile (default 64 = one input block; 0 = off: the plain chain, one 16-lane group per
      // shard), BROTLI_AMD_TILE_WARM: bytes of warm-up before a tile.  Measured on 1 GiB of text in 1 MiB shards
      // (profiles/r03_n_tile_size.txt): parse 40.1 ms with 64 KiB tiles, 43.4 with 128 KiB; 512 … 2048 bytes of warm-up
      // within 0.3 ms of each other.
      uint32_t tile_kb = 64, tile_warm = 2048;
      if (g_env.tile_kb >= 0) tile_kb = (uint32_t)g_env.tile_kb;
      if (g_env.tile_warm >= 0) tile_warm = (uint32_t)g_env.tile_warm;
      if (tile_kb != 0 && !(api_flags & BROTLI_AMD_FLAG_FORCE_SLOW)) plan_add_tiles(plan, tile_kb, tile_warm);
      // shards per wave of k_chain: one 16-lane group per shard, as many waves as stay resident
      const uint64_t resident = (uint64_t)c->num_cus * 4u * CHAIN_WAVES;
      const uint64_t ns = (plan->J.flags & JOB_FLAG_TILED) ? plan->tiles.size() : plan->shards.size();
      // measured (profiles/r02_b/c): the step is bound by instruction issue, so four shards per
      // wave win as soon as there are enough shards to give every SIMD a wave
      (void)resident;
      uint32_t v = ns >= 1024 ? 4u : ns >= 512 ? 2u : 1u;
      if (g_env.cgroups >= 0) v = (uint32_t)g_env.cgroups;
      plan->J.flags &= ~(3u << JOB_FLAG_GROUPS_SHIFT);
      if (v == 1 || v == 2) plan->J.flags |= v << JOB_FLAG_GROUPS_SHIFT;
    }
  }
  return BROTLI_AMD_OK;
}

// Every extern "C" entry runs on the context's device and leaves the caller's current device
// as it found it (a host application may be driving another GPU from the same thread).
struct DeviceScope {
  int prev = -1;
  bool ok = false;
  explicit DeviceScope(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = hipSetDevice(device) == hipSuccess;
  }
  ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};

enum { STAGE_PARSE = 1, STAGE_BUILD = 2, STAGE_STORE = 4, STAGE_ALL = 7 };

// What run_build_store (kernels.h) launches through.
struct HipRun {
  hipStream_t stream;
  void operator()(void (*fn)(JobArgs), const JobArgs& a, uint32_t grid, uint32_t block) const {
    hipLaunchKernelGGL(fn, dim3(grid), dim3(block), 0, stream, a);
  }
};
// Meta-blocks of `longest` bytes at most: one wave each (k_build / k_store: returns 0), or spread over K waves
// (k_wide.h: returns K)?  One wave needs ~14 ms per MiB of text for the two kernels together; the twelve launches of
// the many-wave kernels pay from a few hundred KiB on.  K = a wave per 32 KiB (a part of 4096 commands is about
// that much text), 64 at most.  BROTLI_AMD_WIDE=0 / 1 overrides the choice, BROTLI_AMD_WIDE_KB moves the bound,
// BROTLI_AMD_WIDE_K sets K (experiments).
uint32_t use_wide(const JobParams& J, uint64_t longest) {
  bool wide = J.quality >= 5;
  uint64_t bound = 512u << 10;
  if (g_env.wide_kb > 0 && g_env.wide_kb <= (1 << 20)) bound = (uint64_t)g_env.wide_kb << 10;
  wide = wide && longest >= bound;
  if (g_env.wide >= 0) wide = g_env.wide != 0;
  if (!wide) return 0u;
  uint64_t k = (longest + (32u << 10) - 1u) / (32u << 10);
  if (g_env.wide_k >= 1) k = (uint64_t)g_env.wide_k;
  return k < 1u ? 1u : k > WIDE_K_MAX ? WIDE_K_MAX : (uint32_t)k;
}

// Runs the job's rounds on the context stream.  On return (synchronised) the
// shard states describe the outputs sitting in the workspace.
bool run_rounds(BrotliAmdCtx* c, JobPlan& plan, const uint8_t* d_in, int stages,
                BrotliAmdJobInfo* info, std::vector<ShardState>* states_out) {
  const uint32_t nshards = (uint32_t)plan.shards.size();
  if (!ensure_log2(c, plan.J.log2_lut_size)) return false;
  if (!ensure_ws(c, plan.ws_bytes, nshards)) return false;
  HIP_OK(c, hipEventRecord(c->ev[0], c->stream));
  if (!prepare_tables(c, &plan)) return false;
  HIP_OK(c, hipMemcpyAsync(c->d_shards, plan.shards.data(), nshards * sizeof(ShardDesc),
                           hipMemcpyHostToDevice, c->stream));
  JobArgs a;
  a.J = plan.J;
  a.shards = c->d_shards;
  a.states = c->d_states;
  a.T = c->d_T;
  a.input = d_in;
  a.ws = c->d_ws;
  a.nshards = nshards;
  a.counters = c->d_counters;
  a.cd = c->d_cd;
  const bool tiled = (plan.J.flags & JOB_FLAG_TILED) != 0;
  uint64_t longest_shard = 0;
  for (const ShardDesc& D : plan.shards) if (D.len > longest_shard) longest_shard = D.len;
  const uint32_t wide = use_wide(plan.J, longest_shard < plan.J.max_metablock_size ? longest_shard : plan.J.max_metablock_size);
  const uint32_t ntiles = (uint32_t)plan.tiles.size();
  uint32_t tile_sweeps = 0, tile_bad = 0;
  if (tiled) {
    if (!ensure_tiles(c, ntiles)) return false;
    HIP_OK(c, hipMemcpyAsync(c->d_tiles, plan.tiles.data(), ntiles * sizeof(TileDesc), hipMemcpyHostToDevice, c->stream));
    HIP_OK(c, hipMemsetAsync(c->d_trecs, 0, ntiles * sizeof(TileRec), c->stream));
    a.tiles = c->d_tiles;
    a.trecs = c->d_trecs;
    a.ntiles = ntiles;
  }
  // Table init: enough 256-thread blocks per shard to stream the 128-byte
  // records at HBM rate without flooding the dispatcher.
  uint32_t ibs = 4096u / (nshards < 4096u ? nshards : 4096u);
  if (ibs < 1) ibs = 1;
  if (ibs > 64) ibs = 64;
  a.init_blocks_per_shard = ibs;

  const uint32_t gpw = ((plan.J.flags >> JOB_FLAG_GROUPS_SHIFT) & 3u) ? ((plan.J.flags >> JOB_FLAG_GROUPS_SHIFT) & 3u) : 4u;
  float ms_parse = 0, ms_build = 0, ms_store = 0;
  const bool indexed = (plan.J.flags & JOB_FLAG_INDEXED) != 0;
  if (indexed) a.init_blocks_per_shard = ibs = 1;
  hipLaunchKernelGGL(k_init, dim3(nshards * ibs), dim3(256), 0, c->stream, a);
  HIP_OK(c, hipEventRecord(c->ev[1], c->stream));
  float ms_index = 0;
  // (Batches of whole shards with the index of one batch beside the chain of the one before were measured and
  //  dropped: 95.8 ms per step became 101.7 ... 144.5, profiles/r03_h_batches.txt — the chain's waves and the index
  //  kernels' waves take each other's LDS and issue slots.)
  if (indexed) {
    // the data-parallel half of the parse, once per job (k_index.h)
    hipLaunchKernelGGL(k_ix_count, dim3(nshards * plan.J.ix_slices), dim3(64), 0, c->stream, a);
    hipLaunchKernelGGL(k_ix_scan, dim3(nshards), dim3(64), 0, c->stream, a);
    hipLaunchKernelGGL(k_ix_scatter, dim3(nshards * plan.J.ix_slices), dim3(64),
                       (IX_CHUNK + (3u << plan.J.ix_nb_log2)) * 4u, c->stream, a);
    HIP_OK(c, hipEventRecord(c->ev_ixb, c->stream));
    hipLaunchKernelGGL(k_ix_bucket, dim3(ix_bucket_grid(plan.J, (uint32_t)nshards)), dim3(64), 0, c->stream, a);
    hipLaunchKernelGGL(k_ix_big, dim3(IX_BIG_GRID), dim3(64), 0, c->stream, a);
    if (g_env.ix_debug) {     // diagnostics: the lists' header (records per XCD, cursors, buckets placed twice)
      uint32_t hdr[18];
      (void)hipStreamSynchronize(c->stream);
      (void)hipMemcpy(hdr, c->d_ws + plan.J.big_off, sizeof(hdr), hipMemcpyDeviceToHost);
      fprintf(stderr, "IXDEBUG shards %u giant %u cap %llu off %llu: big blocks per XCD %u %u %u %u %u %u %u %u; cursors %u %u; buckets placed twice: small %u, big %u\n",
              (unsigned)nshards, plan.J.ix_giant, (unsigned long long)plan.J.big_cap, (unsigned long long)plan.J.big_off,
              hdr[0], hdr[1], hdr[2], hdr[3], hdr[4], hdr[5], hdr[6], hdr[7], hdr[8], hdr[9], hdr[16], hdr[17]);
    }
#if defined(IX_PROFILE)   // (experiment builds only: wave-cycles per phase of k_ix_bucket, summed over the waves)
    {
      uint64_t prof[16];
      (void)hipStreamSynchronize(c->stream);
      (void)hipMemcpyFromSymbol(prof, HIP_SYMBOL(ix_prof), sizeof(prof));
      double tot = 0;
      for (int k = 0; k < 8; ++k) tot += (double)prof[k];
      fprintf(stderr, "IXPROF");
      for (int k = 0; k < 8; ++k) fprintf(stderr, " p%d=%.1f%%", k, 100.0 * (double)prof[k] / (tot > 0 ? tot : 1));
      fprintf(stderr, " total=%.3g wave-cycles\n", tot);
      memset(prof, 0, sizeof(prof));
      (void)hipMemcpyToSymbol(HIP_SYMBOL(ix_prof), prof, sizeof(prof));
    }
#endif
    HIP_OK(c, hipEventRecord(c->ev_ix, c->stream));
    if (g_env.index_only) {   // timing experiments: stop after the index kernels
      HIP_OK(c, hipStreamSynchronize(c->stream));
      if (info) {
        HIP_OK(c, hipEventElapsedTime(&ms_index, c->ev[1], c->ev_ix)); info->ms_index = ms_index;
        HIP_OK(c, hipEventElapsedTime(&info->ms_ix_bucket, c->ev_ixb, c->ev_ix));
      }
      if (states_out) {
        states_out->resize(nshards);
        HIP_OK(c, hipMemcpy(states_out->data(), c->d_states, nshards * sizeof(ShardState), hipMemcpyDeviceToHost));
      }
      return true;
    }
  }
  uint32_t rounds = 0;
  for (;;) {
    HIP_OK(c, hipMemsetAsync(c->d_counters, 0, 16 * sizeof(uint32_t), c->stream));
    HIP_OK(c, hipEventRecord(c->ev[2], c->stream));
    if (plan.J.flags & JOB_FLAG_QUICK) {
      hipLaunchKernelGGL(k_parse_quick, dim3(nshards), dim3(64), 0, c->stream, a);
    } else if (plan.J.flags & JOB_FLAG_DEEP) {
      if (plan.J.block_bits <= 6) hipLaunchKernelGGL(k_parse_deep<1>, dim3(nshards), dim3(64), 0, c->stream, a);
      else if (plan.J.block_bits == 7) hipLaunchKernelGGL(k_parse_deep<2>, dim3(nshards), dim3(64), 0, c->stream, a);
      else hipLaunchKernelGGL(k_parse_deep<4>, dim3(nshards), dim3(64), 0, c->stream, a);
    } else if (tiled && rounds != 0) {
      // a shard that left the tiled path and holds more than one meta-block (incompressible data: a cut every
      // max_literals, encode.c:1141-1166): its later rounds are the plain chain's
      JobArgs p = a;
      p.J.flags &= ~(uint32_t)(JOB_FLAG_TILED | JOB_FLAG_SWEEP);
      hipLaunchKernelGGL(k_chain, dim3((nshards + gpw - 1) / gpw), dim3(64), gpw * C_GROUP_LDS_WORDS * 4u, c->stream, p);
      hipLaunchKernelGGL(k_cmd_encode, dim3(nshards * CE_SPLIT), dim3(64), 0, c->stream, p);
    } else if (tiled) {
      // the tiles' parses, then verify / events / sweep until nothing is pending (k_tile.h)
      const dim3 cgrid((ntiles + gpw - 1) / gpw);
      const uint32_t clds = gpw * C_GROUP_LDS_WORDS * 4u;
      const bool tlog = g_env.tile_log;
      double t_prev = 0;                           // (this job's own: contexts on other threads log independently)
      auto lap = [&](const char* what) {           // (diagnostics: wall time per stage, with a sync each)
        if (!tlog) return;
        (void)hipStreamSynchronize(c->stream);
        timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
        const double now = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
        if (what) fprintf(stderr, "  tile stage %-14s %8.3f ms\n", what, now - t_prev);
        t_prev = now;
      };
      lap(nullptr);
      hipLaunchKernelGGL(k_chain_tiles, cgrid, dim3(64), clds, c->stream, a);
      lap("first parse");
      // shards whose tile 0 ended with the static dictionary's gate open (real English): their other tiles once more,
      // the gate taken as open for good (k_tile.h); for everything else the second launch finds nothing to do
      hipLaunchKernelGGL(k_tile_restart, dim3(nshards), dim3(64), 0, c->stream, a);
      hipLaunchKernelGGL(k_tile_restart_clear, dim3(ntiles), dim3(64), 0, c->stream, a);
      hipLaunchKernelGGL(k_chain_tiles, cgrid, dim3(64), clds, c->stream, a);
      lap("second parse");
      bool settled = false;
      uint32_t tc[16];
      for (int pass = 0; pass < 12 && !settled; ++pass) {
        HIP_OK(c, hipMemsetAsync(c->d_counters + TILE_CNT_START, 0, 2 * sizeof(uint32_t), c->stream));
        HIP_OK(c, hipMemsetAsync(c->d_counters + TILE_CNT_RESTART, 0, sizeof(uint32_t), c->stream));
        if (pass != 0 && tc[TILE_CNT_RESTART] != 0) {
          // tiles the gate walk sent back (k_tile.h: the static dictionary's gate may close in them, or has closed in
          // front of them): parsed again from scratch, every search exact against the bitmap as it stands
          JobArgs l = a;
          l.J.flags |= JOB_FLAG_VIEWALL;
          hipLaunchKernelGGL(k_tile_restart_clear, dim3(ntiles), dim3(64), 0, c->stream, l);
          hipLaunchKernelGGL(k_chain_tiles, cgrid, dim3(64), clds, c->stream, l);
          lap("tiles again");
        }
        {
          JobArgs e = a;
          if (pass != 0) e.J.flags |= JOB_FLAG_SWEEP;     // (first pass: cross-tile successors only, k_tile.h)
          hipLaunchKernelGGL(k_tile_events, dim3(nshards * plan.J.ix_slices), dim3(64), 0, c->stream, e);
        }
        hipLaunchKernelGGL(k_tile_verify, dim3(nshards), dim3(64), 0, c->stream, a);
        HIP_OK(c, hipMemcpyAsync(tc, c->d_counters, sizeof(tc), hipMemcpyDeviceToHost, c->stream));
        HIP_OK(c, hipStreamSynchronize(c->stream));
        lap("events+verify");
        tile_bad = tc[TILE_CNT_BAD];
        if (tlog) fprintf(stderr, "tile pass %d: start events %u, changed skip bits %u, shards off the tiled path %u\n",
                          pass, tc[TILE_CNT_START], tc[TILE_CNT_FLIPS], tc[TILE_CNT_BAD]);
        if (tc[TILE_CNT_START] == 0 && tc[TILE_CNT_FLIPS] == 0 && tc[TILE_CNT_RESTART] == 0) { settled = true; break; }
        JobArgs b = a;
        b.J.flags |= JOB_FLAG_SWEEP;
        // a sweep is a few hundred dependent steps per tile around its events: few tiles per wave, so that a tile
        // does not wait for the steps of three others (BROTLI_AMD_SWEEP_GROUPS: 1, 2 or 4 tiles per wave)
        uint32_t sg = 2;        // (measured, profiles/r03_e: 1 GiB text in 1 MiB shards: sweep 23 / 17 ms with 1 / 2 tiles per wave)
        { const int v = g_env.sweep_groups; if (v == 1 || v == 2 || v == 4) sg = (uint32_t)v; }
        b.J.flags &= ~(3u << JOB_FLAG_GROUPS_SHIFT);
        if (sg != 4) b.J.flags |= sg << JOB_FLAG_GROUPS_SHIFT;
        hipLaunchKernelGGL(k_chain_sweep, dim3((ntiles + sg - 1) / sg), dim3(64), sg * C_GROUP_LDS_WORDS * 4u, c->stream, b);
        ++tile_sweeps;
        lap("sweep");
      }
      if (!settled) {
        // give up on the tiles: every tiled shard goes the plain way
        hipLaunchKernelGGL(k_tile_giveup, dim3(nshards), dim3(64), 0, c->stream, a);
        tile_bad = nshards;
      }
      hipLaunchKernelGGL(k_tile_finish, dim3(ntiles), dim3(64), 0, c->stream, a);
      lap("finish");
      if (tile_bad != 0) {
        if (tlog) {
          std::vector<TileRec> tr(ntiles);
          HIP_OK(c, hipMemcpy(tr.data(), c->d_trecs, ntiles * sizeof(TileRec), hipMemcpyDeviceToHost));
          uint32_t why[32] = {0};
          for (const ShardDesc& D : plan.shards) {
            if (D.ntiles <= 1) continue;
            uint32_t f = tr[D.tile_base].flags;
            if (!(f & TILE_BAD)) continue;
            for (uint32_t t = 1; t < D.ntiles; ++t) if (tr[D.tile_base + t].flags & TILE_BAD) f |= tr[D.tile_base + t].flags;
            for (int b = 8; b < 32; ++b) if (f & (1u << b)) ++why[b];
          }
          fprintf(stderr, "  shards off the tiled path: wrap %u error %u no-mb %u not-run %u no-cmd %u gate %u cut %u events %u\n",
                  why[8], why[9], why[10], why[11], why[12], why[13], why[14], why[15]);
        }
        hipLaunchKernelGGL(k_tile_fallback, dim3(nshards), dim3(64), 0, c->stream, a);
        JobArgs p = a;
        p.J.flags &= ~(uint32_t)(JOB_FLAG_TILED | JOB_FLAG_SWEEP);
        hipLaunchKernelGGL(k_chain, dim3((nshards + gpw - 1) / gpw), dim3(64), clds, c->stream, p);
        hipLaunchKernelGGL(k_cmd_encode, dim3(nshards * CE_SPLIT), dim3(64), 0, c->stream, p);
        lap("plain chain");
      }
    } else if (indexed)
      hipLaunchKernelGGL(k_chain, dim3((nshards + gpw - 1) / gpw), dim3(64), gpw * C_GROUP_LDS_WORDS * 4u, c->stream, a);
    else if (plan.J.flags & JOB_FLAG_QUAD)
      hipLaunchKernelGGL(k_parse4, dim3((nshards + gpw - 1) / gpw), dim3(64), 0, c->stream, a);
    else
      hipLaunchKernelGGL(k_parse, dim3(nshards), dim3(64), 0, c->stream, a);
    if (indexed && !tiled) hipLaunchKernelGGL(k_cmd_encode, dim3(nshards * CE_SPLIT), dim3(64), 0, c->stream, a);
    HIP_OK(c, hipEventRecord(c->ev[3], c->stream));
    if ((stages & (STAGE_BUILD | STAGE_STORE)) == (STAGE_BUILD | STAGE_STORE)) {
      HipRun R{c->stream};
      hipError_t ev_err = hipSuccess;
      run_build_store(R, a, nshards, wide, [&]() { ev_err = hipEventRecord(c->ev[4], c->stream); });
      HIP_OK(c, ev_err);
    } else {
      if (stages & STAGE_BUILD) hipLaunchKernelGGL(k_build, dim3(nshards), dim3(64), 0, c->stream, a);
      HIP_OK(c, hipEventRecord(c->ev[4], c->stream));
    }
    HIP_OK(c, hipEventRecord(c->ev[5], c->stream));
    uint32_t counters[16];
    HIP_OK(c, hipMemcpyAsync(counters, c->d_counters, sizeof(counters), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(c, hipStreamSynchronize(c->stream));
    HIP_OK(c, hipGetLastError());
    float t;
    HIP_OK(c, hipEventElapsedTime(&t, c->ev[2], c->ev[3])); ms_parse += t;
    HIP_OK(c, hipEventElapsedTime(&t, c->ev[3], c->ev[4])); ms_build += t;
    HIP_OK(c, hipEventElapsedTime(&t, c->ev[4], c->ev[5])); ms_store += t;
    ++rounds;
    if (counters[1]) return fail(c, "%u shard(s) reported a device fault", counters[1]);
    if (counters[0] == 0 || stages != STAGE_ALL) break;
  }
  if (info) {
    float t;
    HIP_OK(c, hipEventElapsedTime(&t, c->ev[0], c->ev[1]));
    info->ms_init = t;
    info->ms_ix_bucket = 0;
    if (indexed) {
      HIP_OK(c, hipEventElapsedTime(&ms_index, c->ev[1], c->ev_ix));
      HIP_OK(c, hipEventElapsedTime(&info->ms_ix_bucket, c->ev_ixb, c->ev_ix));
    }
    info->ms_index = ms_index;
    info->ms_parse = ms_parse;
    info->ms_build = ms_build;
    info->ms_store = ms_store;
    info->rounds = rounds;
    info->reserved = tile_sweeps | (tile_bad << 8);     // tiled jobs: sweeps of the chain, shards that went the plain way
    info->nshards = nshards;
    info->ws_bytes = plan.ws_bytes;
  }
  if (states_out) {
    states_out->resize(nshards);
    HIP_OK(c, hipMemcpy(states_out->data(), c->d_states, nshards * sizeof(ShardState),
                        hipMemcpyDeviceToHost));
  }
  return true;
}


// One unpartitioned quality-5 stream longer than the window on the tiled path (JOB_FLAG_STREAMT, k_tile.h; the
// simulator's sim_encode_stream is the same sequence).  rc: BROTLI_AMD_OK, BROTLI_AMD_SERIAL (the stream left the
// tiled path: nothing written), BROTLI_AMD_OVERFLOW.  info->reserved = sweeps | meta-blocks << 8 | reasons << 16.
bool run_stream_job(BrotliAmdCtx* c, uint64_t len, const BrotliAmdJobParams* p, const uint8_t* d_in, uint8_t* d_out,
                    uint64_t out_cap, uint64_t* out_size, BrotliAmdJobInfo* info, int* rc) {
  JobPlan plan;
  uint32_t warm = 2048;
  if (g_env.tile_warm >= 0) warm = (uint32_t)g_env.tile_warm;
  uint64_t region = 0;
  if (!plan_stream(len, p->lgwin, p->size_hint, warm, /*ix_in_ws=*/false, &plan, &region)) { *rc = BROTLI_AMD_SERIAL; return true; }
  if (p->flags & BROTLI_AMD_FLAG_NO_LITERAL_CONTEXT) plan.J.flags |= JOB_FLAG_NO_LITCTX;
  if (p->flags & BROTLI_AMD_FLAG_NO_HEADER) plan.J.flags |= JOB_FLAG_NO_HEADER;
  if (p->flags & BROTLI_AMD_FLAG_TAIL_FINISH) plan.J.flags |= JOB_FLAG_TAILFIN;
  const uint32_t ntiles = (uint32_t)plan.tiles.size(), nchunks = plan.J.nchunks, mcap = plan.mcap;
  {
    // ≈ 85 bytes of device memory per input byte (index chunks with their look-back, two command slots per tile,
    // the meta-blocks' workspaces): a stream that does not fit beside what else lives on the device takes the serial
    // path, as it did before there were tiles — never an error of the stock call
    size_t free_b = 0, total_b = 0;
    const uint64_t need = plan.ws_bytes + (uint64_t)nchunks * region + (uint64_t)ntiles * (sizeof(TileDesc) + sizeof(TileRec));
    uint64_t cached = c->ws_cap;
    cached += (uint64_t)c->d_table_chunks.size() * c->chunk_shards * c->chunk_shard_bytes;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > (uint64_t)free_b + cached - ((uint64_t)free_b + cached) / 16u) {
      c->err = "tiled stream: not enough device memory";
      *rc = BROTLI_AMD_SERIAL;
      return true;
    }
  }
  auto room = [&]() -> bool {
    if (!ensure_log2(c, plan.J.log2_lut_size)) return false;
    if (!ensure_ws(c, plan.ws_bytes, 1)) return false;
    if (!ensure_tiles(c, ntiles)) return false;
    // the chunks' index regions: the allocations the shards' regions of a plan live in
    JobPlan cp;
    cp.J = plan.J;
    cp.shards = plan.chunks;
    c->ix_region_bytes = region;
    if (!prepare_tables(c, &cp)) return false;
    plan.chunks = cp.shards;
    return true;
  };
  if (!room()) {
    (void)hipGetLastError();
    *rc = BROTLI_AMD_SERIAL;
    return true;
  }
  if (nchunks > c->chunk_cap) {
    if (c->d_chunks) HIP_OK(c, hipFree(c->d_chunks));
    c->d_chunks = nullptr; c->chunk_cap = 0;
    HIP_OK(c, hipMalloc((void**)&c->d_chunks, nchunks * sizeof(ShardDesc)));
    c->chunk_cap = nchunks;
  }
  if (mcap > c->mb_cap) {
    if (c->d_mdesc) HIP_OK(c, hipFree(c->d_mdesc));
    if (c->d_mstate) HIP_OK(c, hipFree(c->d_mstate));
    if (c->d_moff) HIP_OK(c, hipFree(c->d_moff));
    c->d_mdesc = nullptr; c->d_mstate = nullptr; c->d_moff = nullptr; c->mb_cap = 0;
    HIP_OK(c, hipMalloc((void**)&c->d_mdesc, mcap * sizeof(ShardDesc)));
    HIP_OK(c, hipMalloc((void**)&c->d_mstate, mcap * sizeof(ShardState)));
    HIP_OK(c, hipMalloc((void**)&c->d_moff, (mcap + 3) * sizeof(uint64_t)));     // (+ 2: what stream_tail_fix needs, k_tile.h stream_scan)
    c->mb_cap = mcap;
  }
  HIP_OK(c, hipEventRecord(c->ev[6], c->stream));
  HIP_OK(c, hipMemcpyAsync(c->d_shards, plan.shards.data(), sizeof(ShardDesc), hipMemcpyHostToDevice, c->stream));
  HIP_OK(c, hipMemcpyAsync(c->d_chunks, plan.chunks.data(), nchunks * sizeof(ShardDesc), hipMemcpyHostToDevice, c->stream));
  HIP_OK(c, hipMemcpyAsync(c->d_tiles, plan.tiles.data(), ntiles * sizeof(TileDesc), hipMemcpyHostToDevice, c->stream));
  HIP_OK(c, hipMemsetAsync(c->d_trecs, 0, ntiles * sizeof(TileRec), c->stream));
  HIP_OK(c, hipMemsetAsync(c->d_mstate, 0, mcap * sizeof(ShardState), c->stream));
  HIP_OK(c, hipMemsetAsync(c->d_counters, 0, 16 * sizeof(uint32_t), c->stream));
  const uint64_t zero_out = plan.max_out_bytes + 8 < out_cap ? plan.max_out_bytes + 8 : out_cap;
  HIP_OK(c, hipMemsetAsync(d_out, 0, zero_out, c->stream));
  JobArgs a;
  a.J = plan.J;
  a.shards = c->d_shards;
  a.states = c->d_states;
  a.T = c->d_T;
  a.input = d_in;
  a.ws = c->d_ws;
  a.nshards = 1;
  a.init_blocks_per_shard = 1;
  a.counters = c->d_counters;
  a.tiles = c->d_tiles;
  a.trecs = c->d_trecs;
  a.ntiles = ntiles;
  a.chunks = c->d_chunks;
  a.mdesc = c->d_mdesc;
  a.mstate = c->d_mstate;
  a.moff = c->d_moff;
  a.sout = d_out;
  a.mcap = mcap;
  const bool tlog = g_env.tile_log;
  const bool tlog_each = tlog && g_env.tile_log_level >= 2;     // 2: synchronize and report behind every launch (names a faulting kernel)
  double t_prev = 0;
  auto each = [&](const char* what) {
    if (!tlog_each) return;
    const hipError_t e = hipStreamSynchronize(c->stream);
    fprintf(stderr, "    behind %-22s %s\n", what, e == hipSuccess ? "ok" : hipGetErrorString(e));
    fflush(stderr);
  };
  auto lap = [&](const char* what) {
    if (!tlog) return;
    (void)hipStreamSynchronize(c->stream);
    timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    const double now = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
    if (what) fprintf(stderr, "  stream stage %-14s %8.3f ms\n", what, now - t_prev);
    t_prev = now;
  };
  lap(nullptr);
  hipLaunchKernelGGL(k_init, dim3(1), dim3(256), 0, c->stream, a); each("k_init");
  HIP_OK(c, hipEventRecord(c->ev[1], c->stream));
  {
    JobArgs x = a;                          // the index kernels see the chunks as their shards
    x.shards = c->d_chunks;
    x.nshards = nchunks;
    hipLaunchKernelGGL(k_ix_count, dim3(nchunks * plan.J.ix_slices), dim3(64), 0, c->stream, x); each("k_ix_count");
    hipLaunchKernelGGL(k_ix_scan, dim3(nchunks), dim3(64), 0, c->stream, x); each("k_ix_scan");
    hipLaunchKernelGGL(k_ix_scatter, dim3(nchunks * plan.J.ix_slices), dim3(64), (IX_CHUNK + (3u << plan.J.ix_nb_log2)) * 4u, c->stream, x); each("k_ix_scatter");
    HIP_OK(c, hipEventRecord(c->ev_ixb, c->stream));
    hipLaunchKernelGGL(k_ix_bucket_s, dim3(ix_bucket_grid(plan.J, nchunks)), dim3(64), 0, c->stream, x); each("k_ix_bucket_s");
    hipLaunchKernelGGL(k_ix_big_s, dim3(IX_BIG_GRID), dim3(64), 0, c->stream, x); each("k_ix_big_s");
    HIP_OK(c, hipEventRecord(c->ev_ix, c->stream));
  }
  lap("index");
  const uint32_t nkg = (1u << plan.J.bucket_bits) / 64u;
  const dim3 egrid(nchunks * plan.J.ix_slices);
  hipLaunchKernelGGL(k_stream_kprefix, dim3(nkg), dim3(64), 0, c->stream, a); each("k_stream_kprefix");
  {
    JobArgs z = a;
    z.aux = 1;                   // (every key run; the launches of the pass loop walk the ones that changed: SKT_DIRTY)
    hipLaunchKernelGGL(k_stream_zones, dim3(nchunks * nkg), dim3(64), 0, c->stream, z); each("k_stream_zones");
  }
  HIP_OK(c, hipEventRecord(c->ev[2], c->stream));
  const uint32_t gpw = ntiles >= 1024 ? 4u : ntiles >= 512 ? 2u : 1u;
  {
    JobArgs f = a;
    f.J.flags &= ~(3u << JOB_FLAG_GROUPS_SHIFT);
    if (gpw != 4) f.J.flags |= gpw << JOB_FLAG_GROUPS_SHIFT;
    hipLaunchKernelGGL(k_chain_tiles, dim3((ntiles + gpw - 1) / gpw), dim3(64), gpw * C_GROUP_LDS_WORDS * 4u, c->stream, f); each("k_chain_tiles");
    lap("first parse");
    hipLaunchKernelGGL(k_tile_restart, dim3(1), dim3(64), 0, c->stream, a); each("k_tile_restart");
    hipLaunchKernelGGL(k_tile_restart_clear, dim3(ntiles), dim3(64), 0, c->stream, a); each("k_tile_restart_clear");
    hipLaunchKernelGGL(k_chain_tiles, dim3((ntiles + gpw - 1) / gpw), dim3(64), gpw * C_GROUP_LDS_WORDS * 4u, c->stream, f); each("k_chain_tiles");
  }
  lap("second parse");
  uint32_t tc[16], sweeps = 0, reasons = 0, nmb = 0, passes = 0, chaotic = 0;
  auto reasons_of = [&]() -> bool {
    TileRec r0;
    HIP_OK(c, hipMemcpy(&r0, c->d_trecs, sizeof(TileRec), hipMemcpyDeviceToHost));
    reasons = r0.flags >> 8;
    return true;
  };
  // (the whole loop once more when a raw meta-block rolls the distance cache back for the tile behind it: which
  //  meta-blocks are raw is known behind k_store / k_stream_scan only, encode.c:598-614)
  for (int outer = 0;; ++outer) {
  if (outer >= 8) {
    if (info) info->reserved = sweeps | ((TILE_WHY_RAW >> 8) << 16);
    *rc = BROTLI_AMD_SERIAL;
    return true;
  }
  bool settled = false;
  for (int pass = 0; pass < 24 && !settled; ++pass, ++passes) {
    HIP_OK(c, hipMemsetAsync(c->d_counters + TILE_CNT_START, 0, 2 * sizeof(uint32_t), c->stream));
    HIP_OK(c, hipMemsetAsync(c->d_counters + TILE_CNT_RESTART, 0, sizeof(uint32_t), c->stream));
    if (passes != 0 && tc[TILE_CNT_RESTART] != 0) {
      // tiles the gate walk sent back (k_tile.h): parsed again from scratch, every search exact
      JobArgs l = a;
      l.J.flags |= JOB_FLAG_VIEWALL;
      l.J.flags &= ~(3u << JOB_FLAG_GROUPS_SHIFT);
      if (gpw != 4) l.J.flags |= gpw << JOB_FLAG_GROUPS_SHIFT;
      hipLaunchKernelGGL(k_tile_restart_clear, dim3(ntiles), dim3(64), 0, c->stream, l); each("k_tile_restart_clear");
      hipLaunchKernelGGL(k_chain_tiles, dim3((ntiles + gpw - 1) / gpw), dim3(64), gpw * C_GROUP_LDS_WORDS * 4u, c->stream, l); each("k_chain_tiles");
      lap("tiles again");
    }
    {
      JobArgs e = a;
      if (passes != 0) e.J.flags |= JOB_FLAG_SWEEP;
      hipLaunchKernelGGL(k_stream_flips, egrid, dim3(64), 0, c->stream, e); each("k_stream_flips");
      hipLaunchKernelGGL(k_stream_flipcheck, dim3(1), dim3(64), 0, c->stream, e); each("k_stream_flipcheck");
      hipLaunchKernelGGL(k_stream_events, egrid, dim3(64), 0, c->stream, e); each("k_stream_events");
    }
    hipLaunchKernelGGL(k_stream_skclear, egrid, dim3(64), 0, c->stream, a); each("k_stream_skclear");
    hipLaunchKernelGGL(k_stream_skcount, egrid, dim3(64), 0, c->stream, a); each("k_stream_skcount");
    hipLaunchKernelGGL(k_stream_kprefix, dim3(nkg), dim3(64), 0, c->stream, a); each("k_stream_kprefix");
    a.aux = 0;
    hipLaunchKernelGGL(k_stream_zones, dim3(nchunks * nkg), dim3(64), 0, c->stream, a); each("k_stream_zones");
    hipLaunchKernelGGL(k_stream_cuts, dim3(1), dim3(64), 0, c->stream, a); each("k_stream_cuts");
    hipLaunchKernelGGL(k_stream_verify, dim3((ntiles + 63u) / 64u), dim3(64), 0, c->stream, a); each("k_stream_verify");
    HIP_OK(c, hipMemcpyAsync(tc, c->d_counters, sizeof(tc), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(c, hipStreamSynchronize(c->stream));
    lap("events..verify");
    if (tlog) fprintf(stderr, "stream pass %d: start events %u, changed bits / marks %u, off the tiled path %u\n", pass,
                      tc[TILE_CNT_START], tc[TILE_CNT_FLIPS], tc[TILE_CNT_BAD]);
    if (tc[TILE_CNT_BAD] != 0) {
      if (!reasons_of()) return false;
      if (info) info->reserved = sweeps | (reasons << 16);
      *rc = BROTLI_AMD_SERIAL;
      return true;
    }
    if (tc[TILE_CNT_START] == 0 && tc[TILE_CNT_FLIPS] == 0 && tc[TILE_CNT_RESTART] == 0) { settled = true; break; }
    // Data on which a tile's parse depends chaotically on what was stored in front of it (arrays of floats: the literal
    // spree's phase behind every short match) never settles: pass after pass more than half of ALL tiles come back with
    // a new in-state (178, 195, 166, 164 ... of 256 for 16 MiB of floats, 24 passes = 12 s before the serial stream took
    // over).  A stream whose passes 1 .. 4 each changed the in-state of more than 40 % of its tiles goes there now; slow
    // settlers stay (table rows: 35 % falling; the mix, whose chaotic members are an eighth of its tiles).
    if (pass >= 1 && pass <= 4 && outer == 0) chaotic += tc[TILE_CNT_START] > ntiles / 5u * 2u ? 1u : 0u;
    if (pass == 4 && outer == 0 && chaotic == 4u) {
      if (info) info->reserved = sweeps | ((TILE_WHY_EVENTS >> 8) << 16);
      *rc = BROTLI_AMD_SERIAL;
      return true;
    }
    JobArgs b = a;
    b.J.flags |= JOB_FLAG_SWEEP;
    uint32_t sg = 2;
    { const int v = g_env.sweep_groups; if (v == 1 || v == 2 || v == 4) sg = (uint32_t)v; }
    b.J.flags &= ~(3u << JOB_FLAG_GROUPS_SHIFT);
    if (sg != 4) b.J.flags |= sg << JOB_FLAG_GROUPS_SHIFT;
    hipLaunchKernelGGL(k_chain_sweep, dim3((ntiles + sg - 1) / sg), dim3(64), sg * C_GROUP_LDS_WORDS * 4u, c->stream, b); each("k_chain_sweep");
    ++sweeps;
    lap("sweep");
  }
  if (!settled) {
    if (info) info->reserved = sweeps | ((TILE_WHY_EVENTS >> 8) << 16);
    *rc = BROTLI_AMD_SERIAL;
    return true;
  }
  a.aux = 1;
  hipLaunchKernelGGL(k_stream_cuts, dim3(1), dim3(64), 0, c->stream, a); each("k_stream_cuts");
  // (the finalize cut can still take the stream off the tiled path — more meta-blocks than planned, a tile that went
  //  bad: nothing behind it may then read the descriptors it did not write)
  HIP_OK(c, hipMemcpyAsync(tc, c->d_counters, sizeof(tc), hipMemcpyDeviceToHost, c->stream));
  HIP_OK(c, hipStreamSynchronize(c->stream));
  nmb = tc[TILE_CNT_NMB];
  if (tc[TILE_CNT_BAD] != 0 || nmb == 0 || nmb > mcap) {
    if (!reasons_of()) return false;
    if (info) info->reserved = sweeps | (nmb << 8) | (reasons << 16);
    *rc = BROTLI_AMD_SERIAL;
    return true;
  }
  hipLaunchKernelGGL(k_stream_finish, dim3(ntiles), dim3(64), 0, c->stream, a); each("k_stream_finish");
  HIP_OK(c, hipEventRecord(c->ev[3], c->stream));
  lap("cuts+finish");
  {
    JobArgs m = a;                          // build / store see the meta-blocks as their shards
    m.shards = c->d_mdesc;
    m.states = c->d_mstate;
    m.nshards = nmb;
    HipRun R{c->stream};
    hipError_t ev_err = hipSuccess;
    run_build_store(R, m, nmb, use_wide(plan.J, plan.J.max_metablock_size), [&]() { ev_err = hipEventRecord(c->ev[4], c->stream); each("build half"); });
    HIP_OK(c, ev_err);
    each("store half");
    HIP_OK(c, hipEventRecord(c->ev[5], c->stream));
  }
  lap("build+store");
  hipLaunchKernelGGL(k_stream_scan, dim3(1), dim3(64), 0, c->stream, a); each("k_stream_scan");
  HIP_OK(c, hipMemcpyAsync(tc, c->d_counters, sizeof(tc), hipMemcpyDeviceToHost, c->stream));
  HIP_OK(c, hipStreamSynchronize(c->stream));
  // a meta-block that k_build / k_store gave up on, or one that was not built: the serial device stream encodes the
  // input instead (the stock call must not fail on data the library handled before there were tiles)
  if (tc[1] != 0 || tc[TILE_CNT_RAW] != 0 || tc[TILE_CNT_BAD] != 0 || tc[TILE_CNT_NMB] != nmb) {
    if (!reasons_of()) return false;
    if (tlog) fprintf(stderr, "stream: off the tiled path behind build / store (errors %u, not built %u, bad %u)\n", tc[1], tc[TILE_CNT_RAW], tc[TILE_CNT_BAD]);
    if (info) info->reserved = sweeps | (nmb << 8) | ((reasons | (TILE_WHY_ERROR >> 8)) << 16);
    c->err = "tiled stream: a meta-block was not built";
    *rc = BROTLI_AMD_SERIAL;
    return true;
  }
  HIP_OK(c, hipMemsetAsync(c->d_counters + TILE_CNT_RBCHG, 0, sizeof(uint32_t), c->stream));
  hipLaunchKernelGGL(k_stream_rollback, dim3(1), dim3(64), 0, c->stream, a); each("k_stream_rollback");
  HIP_OK(c, hipMemcpyAsync(tc, c->d_counters, sizeof(tc), hipMemcpyDeviceToHost, c->stream));
  HIP_OK(c, hipStreamSynchronize(c->stream));
  if (tlog) fprintf(stderr, "stream: %u meta-blocks, tiles with a new roll-back behind a raw meta-block: %u\n", nmb, tc[TILE_CNT_RBCHG]);
  if (tc[TILE_CNT_RBCHG] == 0) break;
  HIP_OK(c, hipMemsetAsync(c->d_mstate, 0, mcap * sizeof(ShardState), c->stream));
  }
  uint64_t tail[3] = {0, 0, 0};          // bits of the stream; stream_tail_fix applies (0 / 1), at which bit
  HIP_OK(c, hipMemcpy(tail, c->d_moff + nmb, sizeof(tail), hipMemcpyDeviceToHost));
  const uint64_t total_bits = tail[0];
  c->tail_fix = (p->flags & BROTLI_AMD_FLAG_TAIL_FINISH) != 0 && tail[1] != 0;
  c->tail_bit = tail[2];
  c->tail_total_bits = total_bits;
  const uint64_t total = (total_bits + 7) / 8;
  *out_size = total;
  if (total + 8 > out_cap) { c->err = "output capacity too small"; *rc = BROTLI_AMD_OVERFLOW; return true; }
  hipLaunchKernelGGL(k_stream_place, dim3(nmb * STREAM_PLACE_PARTS), dim3(256), 0, c->stream, a); each("k_stream_place");
  HIP_OK(c, hipEventRecord(c->ev[7], c->stream));
  HIP_OK(c, hipStreamSynchronize(c->stream));
  HIP_OK(c, hipGetLastError());
  lap("place");
  if (info) {
    float t;
    HIP_OK(c, hipEventElapsedTime(&t, c->ev[6], c->ev[1])); info->ms_init = t;
    HIP_OK(c, hipEventElapsedTime(&t, c->ev[1], c->ev_ix)); info->ms_index = t;
    HIP_OK(c, hipEventElapsedTime(&t, c->ev_ixb, c->ev_ix)); info->ms_ix_bucket = t;
    HIP_OK(c, hipEventElapsedTime(&t, c->ev_ix, c->ev[3])); info->ms_parse = t;
    HIP_OK(c, hipEventElapsedTime(&t, c->ev[3], c->ev[4])); info->ms_build = t;
    HIP_OK(c, hipEventElapsedTime(&t, c->ev[4], c->ev[5])); info->ms_store = t;
    HIP_OK(c, hipEventElapsedTime(&t, c->ev[5], c->ev[7])); info->ms_gather = t;
    HIP_OK(c, hipEventElapsedTime(&t, c->ev[6], c->ev[7])); info->ms_total = t;
    info->rounds = 1;
    info->reserved = (sweeps & 0xFFu) | (nmb << 8);
    info->nshards = 1;
    info->ws_bytes = plan.ws_bytes;
    info->out_bytes = total;
  }
  *rc = BROTLI_AMD_OK;
  return true;
}

}  // namespace

extern "C" {

void brotli_amd_refresh_env(void) { env_read(); }

int brotli_amd_ctx_create(int device, const char* tables_path, BrotliAmdCtx** out) {
  *out = nullptr;
  BrotliAmdCtx* c = new BrotliAmdCtx();
  *out = c;   // returned even on failure so the caller can read the error
  env_read();
  c->device = device;
  c->tables_path = tables_path;
  if (!host_tables_load(tables_path, &c->ht)) {
    fail(c, "cannot load format tables from %s", tables_path);
    return BROTLI_AMD_ERROR;
  }
  auto body = [&]() -> bool {
    int n = 0;
    HIP_OK(c, hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(c, "no HIP device %d (count %d)", device, n);
    DeviceScope dev(device);
    if (!dev.ok) return fail(c, "hipSetDevice(%d) failed", device);
    hipDeviceProp_t prop;
    HIP_OK(c, hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
      return fail(c, "device %d is %s; this library contains gfx950 code only", device, prop.gcnArchName);
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_OK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (auto& e : c->ev) HIP_OK(c, hipEventCreate(&e));
    HIP_OK(c, hipEventCreate(&c->ev_ix));
    HIP_OK(c, hipEventCreate(&c->ev_ixb));
    if (!dev_upload(c, &c->d_lut, c->ht.context_lut, 2048)) return false;
    if (!dev_upload(c, &c->d_dict, c->ht.dict.data(), c->ht.dict.size())) return false;
    if (!dev_upload(c, &c->d_hash_words, c->ht.hash_words.data(), 32768 * 2)) return false;
    if (!dev_upload(c, &c->d_hash_lengths, c->ht.hash_lengths.data(), 32768)) return false;
    return ensure_log2(c, 1u << 16);
  };
  return body() ? BROTLI_AMD_OK : BROTLI_AMD_ERROR;
}

void brotli_amd_ctx_destroy(BrotliAmdCtx* c) {
  if (!c) return;
  DeviceScope dev(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  void* ptrs[] = {c->d_lut, c->d_dict, c->d_hash_words, c->d_hash_lengths, c->d_log2, c->d_T,
                  c->d_ws, c->d_shards, c->d_states, c->d_scan, c->d_counters, c->d_tiles, c->d_trecs,
                  c->d_chunks, c->d_mdesc, c->d_mstate, c->d_moff,
                  c->d_stage_in, c->d_stage_out, c->d_ffrags, c->d_fblocks, c->d_fbstate,
                  c->d_ffstate, c->d_fresult, c->d_transforms, c->d_transform_text, c->d_dec_arena,
                  c->d_dec_pieces, c->d_dec_results};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  for (uint8_t* p : c->d_table_chunks) if (p) (void)hipFree(p);
  for (void* p : c->dict_allocs) (void)hipFree(p);
  for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
  if (c->ev_ix) (void)hipEventDestroy(c->ev_ix);
  if (c->ev_ixb) (void)hipEventDestroy(c->ev_ixb);
  for (auto& l : c->lanes) {
    for (int k = 0; k < 2; ++k) { if (l.pin[k]) (void)hipHostFree(l.pin[k]); if (l.ev[k]) (void)hipEventDestroy(l.ev[k]); }
    if (l.s) (void)hipStreamDestroy(l.s);
  }
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* brotli_amd_last_error(const BrotliAmdCtx* c) { return c ? c->err.c_str() : "no context"; }

uint64_t brotli_amd_max_output(uint64_t len, const BrotliAmdJobParams* p) {
  JobPlan plan;
  if (len == 0) return 16;
  if ((p->flags & BROTLI_AMD_FLAG_STREAM_TILES) && p->quality == 5 && p->shard_size == 0 && p->stream_base == 0) {
    // a held stream on the tiled path: the stream's own bound (len + 8 per possible meta-block), not the 2 n of a
    // one-shard job — a 1 GiB call does not ask its caller for 2 GiB (a stream that leaves the tiles answers
    // BROTLI_AMD_SERIAL and writes nothing, whatever the capacity)
    JobPlan sp;
    if (plan_stream(len, p->lgwin, p->size_hint, 2048u, /*ix_in_ws=*/true, &sp)) return sp.max_out_bytes + 8;
  }
  if (!plan_job(len, p->quality, p->lgwin, p->size_hint, p->shard_size, p->stream_base,
                p->is_last != 0, &plan, true, (int)((p->flags >> BROTLI_AMD_FLAG_LGBLOCK_SHIFT) & 31u))) return 0;
  return plan.max_out_bytes;
}

int brotli_amd_encode_device(BrotliAmdCtx* c, const void* d_in, uint64_t len,
                             const BrotliAmdJobParams* p, void* d_out, uint64_t out_cap,
                             uint64_t* out_size, uint64_t* d_shard_sizes,
                             BrotliAmdJobInfo* info) {
  BrotliAmdJobInfo local;
  if (!info) info = &local;
  memset(info, 0, sizeof(*info));
  *out_size = 0;
  DeviceScope dev(c->device);
  if (!dev.ok) { fail(c, "hipSetDevice failed"); return BROTLI_AMD_ERROR; }
  if ((p->flags & BROTLI_AMD_FLAG_TAIL_FINISH) && !c->from_host) {
    // the flag's second half — the rewrite of the last meta-block's header — is done on the host by brotli_amd_encode_host;
    // a device-side caller would get the one-shot form with the raw / compressed decision of the other
    fail(c, "BROTLI_AMD_FLAG_TAIL_FINISH is honoured by brotli_amd_encode_host only");
    return BROTLI_AMD_UNSUPPORTED;
  }
  if (p->flags & BROTLI_AMD_FLAG_STREAM_TILES) {
    if (p->quality != 5 || p->shard_size != 0 || p->stream_base != 0 || !p->is_last || c->d_cd || d_shard_sizes ||
        ((p->flags >> BROTLI_AMD_FLAG_LGBLOCK_SHIFT) & 31u) != 0u) {
      fail(c, "BROTLI_AMD_FLAG_STREAM_TILES: quality 5, one whole stream, no dictionary, the default block size");
      return BROTLI_AMD_UNSUPPORTED;
    }
    int src = BROTLI_AMD_ERROR;
    c->tail_fix = false;
    if (!run_stream_job(c, len, p, (const uint8_t*)d_in, (uint8_t*)d_out, out_cap, out_size, info, &src))
      return c->err.find("device fault") != std::string::npos ? BROTLI_AMD_DEVICE_FAULT : BROTLI_AMD_ERROR;
    if (src == BROTLI_AMD_SERIAL) c->err = "the stream left the tiled path";
    return src;
  }
  JobPlan plan;
  int rc = plan_from_params(c, len, p, &plan);
  if (rc != BROTLI_AMD_OK) return rc;
  const uint32_t nshards = (uint32_t)plan.shards.size();
  auto body = [&]() -> bool {
    HIP_OK(c, hipEventRecord(c->ev[6], c->stream));
    if (!run_rounds(c, plan, (const uint8_t*)d_in, STAGE_ALL, info, nullptr)) return false;
    // Concatenate the shard outputs: exclusive scan of sizes, then one
    // 16-byte-per-lane copy grid.
    JobArgs a;
    a.J = plan.J;
    a.shards = c->d_shards;
    a.states = c->d_states;
    a.T = c->d_T;
    a.input = (const uint8_t*)d_in;
    a.ws = c->d_ws;
    a.nshards = nshards;
    a.counters = c->d_counters;
    a.init_blocks_per_shard = 1;
    HIP_OK(c, hipEventRecord(c->ev[0], c->stream));
    hipLaunchKernelGGL(k_scan_sizes, dim3(1), dim3(1024), 0, c->stream, a, c->d_scan, d_shard_sizes);
    uint64_t total = 0;
    HIP_OK(c, hipMemcpyAsync(&total, c->d_scan + nshards, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_OK(c, hipStreamSynchronize(c->stream));
    if (total > out_cap) { *out_size = total; c->err = "output capacity too small"; rc = BROTLI_AMD_OVERFLOW; return true; }
    // ~64 KiB of output per block.
    uint64_t avg = total / nshards + 1;
    uint32_t bps = (uint32_t)((avg + 65535) / 65536);
    if (bps < 1) bps = 1;
    if (bps > 1024) bps = 1024;
    hipLaunchKernelGGL(k_gather, dim3(nshards * bps), dim3(256), 0, c->stream, a, c->d_scan,
                       (uint8_t*)d_out, bps);
    HIP_OK(c, hipEventRecord(c->ev[1], c->stream));
    HIP_OK(c, hipEventRecord(c->ev[7], c->stream));
    HIP_OK(c, hipStreamSynchronize(c->stream));
    HIP_OK(c, hipGetLastError());
    float t;
    HIP_OK(c, hipEventElapsedTime(&t, c->ev[0], c->ev[1])); info->ms_gather = t;
    HIP_OK(c, hipEventElapsedTime(&t, c->ev[6], c->ev[7])); info->ms_total = t;
    info->out_bytes = total;
    *out_size = total;
#if defined(S_PROFILE) || defined(B_PROFILE)
    {  // prof_out: phase cycles of the store kernel, summed over shards
      std::vector<ShardState> st(nshards);
      HIP_OK(c, hipMemcpy(st.data(), c->d_states, nshards * sizeof(ShardState), hipMemcpyDeviceToHost));
      for (uint32_t k = 0; k < nshards; ++k) for (int i = 0; i < 12; ++i) info->prof[i] += st[k].prof[i];
    }
#endif
    return true;
  };
  if (!body()) return c->err.find("device fault") != std::string::npos ? BROTLI_AMD_DEVICE_FAULT : BROTLI_AMD_ERROR;
  return rc;
}

int brotli_amd_encode_host(BrotliAmdCtx* c, const uint8_t* in, uint64_t len,
                           const BrotliAmdJobParams* p, uint8_t* out, uint64_t out_cap,
                           uint64_t* out_size, BrotliAmdJobInfo* info) {
  *out_size = 0;
  DeviceScope dev(c->device);
  if (!dev.ok) { fail(c, "hipSetDevice failed"); return BROTLI_AMD_ERROR; }
  const uint64_t max_out = brotli_amd_max_output(len, p);
  if (max_out == 0) { fail(c, "parameters outside the GPU path"); return BROTLI_AMD_UNSUPPORTED; }
  auto stage = [&]() -> bool {
    if (len + BROTLI_AMD_INPUT_SLACK > c->stage_in_cap) {
      if (c->d_stage_in) HIP_OK(c, hipFree(c->d_stage_in));
      c->d_stage_in = nullptr; c->stage_in_cap = 0;
      HIP_OK(c, hipMalloc((void**)&c->d_stage_in, len + BROTLI_AMD_INPUT_SLACK));
      c->stage_in_cap = len + BROTLI_AMD_INPUT_SLACK;
    }
    if (max_out > c->stage_out_cap) {
      if (c->d_stage_out) HIP_OK(c, hipFree(c->d_stage_out));
      c->d_stage_out = nullptr; c->stage_out_cap = 0;
      HIP_OK(c, hipMalloc((void**)&c->d_stage_out, max_out));
      c->stage_out_cap = max_out;
    }
    if (!big_copy(c, c->d_stage_in, in, len, true)) { fail(c, "H2D copy failed"); return false; }
    HIP_OK(c, hipMemsetAsync(c->d_stage_in + len, 0, BROTLI_AMD_INPUT_SLACK, c->stream));
    return true;
  };
  // (Encoding a plan in batches while the next batch travels was measured and dropped: the chain
  // kernel takes as long for 1024 shards as for 8192, so every batch pays the whole latency.)
  const bool hlog = g_env.tile_log;       // host-side laps of the call, next to the device stages
  auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  if (!stage()) return BROTLI_AMD_ERROR;
  const double t1 = now();
  uint64_t n = 0;
  c->from_host = true;
  int rc = brotli_amd_encode_device(c, c->d_stage_in, len, p, c->d_stage_out, c->stage_out_cap, &n,
                                    nullptr, info);
  c->from_host = false;
  if (rc != BROTLI_AMD_OK) return rc;
  const double t2 = now();
  *out_size = n;
  const bool tail_fix = (p->flags & BROTLI_AMD_FLAG_STREAM_TILES) != 0 && c->tail_fix;
  if (n + (tail_fix ? 1u : 0u) > out_cap) { c->err = "output capacity too small"; return BROTLI_AMD_OVERFLOW; }
  if (!big_copy(c, out, c->d_stage_out, n, false)) {
    fail(c, "D2H copy failed");
    return BROTLI_AMD_ERROR;
  }
  if (tail_fix) {
    // (the stream came in PROCESS calls ending on a block boundary, the FINISH came empty, and the rule of the cuts
    //  closes the last meta-block there: host_plan.h — a few MB of bits moved by one, on the host)
    out[n] = 0;
    *out_size = stream_tail_fix(out, c->tail_bit, c->tail_total_bits);
    c->tail_fix = false;
  }
  if (hlog) fprintf(stderr, "  host: staging + H2D of %llu bytes %.1f ms, encode_device %.1f ms, D2H of %llu bytes %.1f ms\n",
                    (unsigned long long)len, t1 - t0, t2 - t1, (unsigned long long)n, now() - t2);
  return BROTLI_AMD_OK;
}

// ---- quality 1 ---------------------------------------------------------------------------
uint64_t brotli_amd_fast_max_output(uint64_t len, uint64_t ncalls, int lgwin) {
  if (lgwin < 10 || lgwin > 24) return 0;
  const uint64_t nfrag = ncalls + (len >> lgwin) + 1;
  return len + 8 * nfrag + 64;
}

int brotli_amd_encode_fast_device(BrotliAmdCtx* c, const void* d_in, uint64_t len,
                                  const uint64_t* call_sizes, uint64_t ncalls,
                                  const BrotliAmdFastParams* p, void* d_out, uint64_t out_cap,
                                  uint64_t* out_bits, BrotliAmdJobInfo* info) {
  BrotliAmdJobInfo local;
  if (!info) info = &local;
  memset(info, 0, sizeof(*info));
  *out_bits = 0;
  DeviceScope dev(c->device);
  if (!dev.ok) { fail(c, "hipSetDevice failed"); return BROTLI_AMD_ERROR; }
  if (p->carry_bits > 15) { fail(c, "carry_bits > 15"); return BROTLI_AMD_UNSUPPORTED; }
  FastPlan plan;
  if (!plan_fast(len, p->lgwin, call_sizes, (size_t)ncalls, &plan)) {
    fail(c, "quality 1: bad call list or lgwin %d", p->lgwin);
    return BROTLI_AMD_UNSUPPORTED;
  }
  if (plan.frags.empty()) { fail(c, "quality 1: empty job"); return BROTLI_AMD_UNSUPPORTED; }
  int rc = BROTLI_AMD_OK;
  auto body = [&]() -> bool {
    const uint64_t nf = plan.frags.size(), nb = plan.blocks.size();
    if (!ensure_log2(c, 1u << 16)) return false;
    if (!ensure_ws(c, plan.ws_bytes, 1)) return false;
    if (nf > c->ffrag_cap) {
      if (c->d_ffrags) HIP_OK(c, hipFree(c->d_ffrags));
      if (c->d_ffstate) HIP_OK(c, hipFree(c->d_ffstate));
      c->d_ffrags = nullptr; c->d_ffstate = nullptr; c->ffrag_cap = 0;
      HIP_OK(c, hipMalloc((void**)&c->d_ffrags, nf * sizeof(FastFrag)));
      HIP_OK(c, hipMalloc((void**)&c->d_ffstate, nf * sizeof(FastFragState)));
      c->ffrag_cap = nf;
    }
    if (nb + 1 > c->fblock_cap) {
      if (c->d_fblocks) HIP_OK(c, hipFree(c->d_fblocks));
      if (c->d_fbstate) HIP_OK(c, hipFree(c->d_fbstate));
      c->d_fblocks = nullptr; c->d_fbstate = nullptr; c->fblock_cap = 0;
      HIP_OK(c, hipMalloc((void**)&c->d_fblocks, (nb + 1) * sizeof(FastBlock)));
      HIP_OK(c, hipMalloc((void**)&c->d_fbstate, (nb + 1) * sizeof(FastBlockState)));
      c->fblock_cap = nb + 1;
    }
    if (!c->d_fresult) HIP_OK(c, hipMalloc((void**)&c->d_fresult, 2 * sizeof(uint64_t)));
    const uint64_t need_out = plan.max_out_bytes < out_cap ? plan.max_out_bytes : out_cap;
    HIP_OK(c, hipEventRecord(c->ev[6], c->stream));
    HIP_OK(c, hipMemcpyAsync(c->d_ffrags, plan.frags.data(), nf * sizeof(FastFrag), hipMemcpyHostToDevice, c->stream));
    if (nb) HIP_OK(c, hipMemcpyAsync(c->d_fblocks, plan.blocks.data(), nb * sizeof(FastBlock), hipMemcpyHostToDevice, c->stream));
    HIP_OK(c, hipMemsetAsync(c->d_ffstate, 0, nf * sizeof(FastFragState), c->stream));
    HIP_OK(c, hipMemsetAsync(c->d_fresult, 0, 2 * sizeof(uint64_t), c->stream));
    HIP_OK(c, hipMemsetAsync(d_out, 0, need_out, c->stream));
    FastArgs a;
    a.frags = c->d_ffrags;
    a.blocks = c->d_fblocks;
    a.bstate = c->d_fbstate;
    a.fstate = c->d_ffstate;
    a.T = c->d_T;
    a.input = (const uint8_t*)d_in;
    a.ws = c->d_ws;
    a.out = (uint8_t*)d_out;
    a.result = c->d_fresult;
    a.cmds_base = plan.cmds_base; a.lits_base = plan.lits_base; a.lsum_base = plan.lsum_base;
    a.scr_base = plan.scr_base; a.tables_base = plan.tables_base;
    a.out_cap = out_cap;
    a.nfrags = (uint32_t)nf;
    a.nblocks = (uint32_t)nb;
    a.nslots = plan.nslots;
    a.carry_bits = p->carry_bits;
    a.carry_value = p->carry_value;
    a.is_last = p->is_last ? 1u : 0u;
    HIP_OK(c, hipEventRecord(c->ev[0], c->stream));
    hipLaunchKernelGGL(k_fast_parse, dim3(a.nslots), dim3(64), 0, c->stream, a);
    HIP_OK(c, hipEventRecord(c->ev[1], c->stream));
    if (nb) hipLaunchKernelGGL(k_fast_store, dim3(a.nblocks), dim3(64), 0, c->stream, a);
    HIP_OK(c, hipEventRecord(c->ev[2], c->stream));
    hipLaunchKernelGGL(k_fast_sizes, dim3((8 * a.nfrags + 255) / 256), dim3(256), 0, c->stream, a);
    hipLaunchKernelGGL(k_fast_scan, dim3(1), dim3(64), 0, c->stream, a);
    if (nb) hipLaunchKernelGGL(k_fast_emit, dim3(a.nblocks), dim3(256), 0, c->stream, a);
    HIP_OK(c, hipEventRecord(c->ev[3], c->stream));
    uint64_t result[2] = {0, 0};
    HIP_OK(c, hipMemcpyAsync(result, c->d_fresult, sizeof(result), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(c, hipEventRecord(c->ev[7], c->stream));
    HIP_OK(c, hipStreamSynchronize(c->stream));
    HIP_OK(c, hipGetLastError());
    float t;
    HIP_OK(c, hipEventElapsedTime(&t, c->ev[0], c->ev[1])); info->ms_parse = t;
    HIP_OK(c, hipEventElapsedTime(&t, c->ev[1], c->ev[2])); info->ms_store = t;
    HIP_OK(c, hipEventElapsedTime(&t, c->ev[2], c->ev[3])); info->ms_gather = t;
    HIP_OK(c, hipEventElapsedTime(&t, c->ev[6], c->ev[7])); info->ms_total = t;
    info->nshards = nf;
    info->rounds = 1;
    info->ws_bytes = plan.ws_bytes;
    if (result[1] & 2u) { c->err = "output capacity too small"; rc = BROTLI_AMD_OVERFLOW; return true; }
    if (result[1]) return fail(c, "quality 1: a block overflowed its scratch (device fault)");
    *out_bits = result[0];
    info->out_bytes = (result[0] + 7) / 8;
    return true;
  };
  if (!body()) return c->err.find("device fault") != std::string::npos ? BROTLI_AMD_DEVICE_FAULT : BROTLI_AMD_ERROR;
  return rc;
}

int brotli_amd_encode_fast_host(BrotliAmdCtx* c, const uint8_t* in, uint64_t len,
                                const uint64_t* call_sizes, uint64_t ncalls,
                                const BrotliAmdFastParams* p, uint8_t* out, uint64_t out_cap,
                                uint64_t* out_bits, BrotliAmdJobInfo* info) {
  *out_bits = 0;
  DeviceScope dev(c->device);
  if (!dev.ok) { fail(c, "hipSetDevice failed"); return BROTLI_AMD_ERROR; }
  const uint64_t max_out = brotli_amd_fast_max_output(len, ncalls, p->lgwin);
  if (max_out == 0) { fail(c, "quality 1: lgwin %d", p->lgwin); return BROTLI_AMD_UNSUPPORTED; }
  auto stage = [&]() -> bool {
    if (len + BROTLI_AMD_INPUT_SLACK > c->stage_in_cap) {
      if (c->d_stage_in) HIP_OK(c, hipFree(c->d_stage_in));
      c->d_stage_in = nullptr; c->stage_in_cap = 0;
      HIP_OK(c, hipMalloc((void**)&c->d_stage_in, len + BROTLI_AMD_INPUT_SLACK));
      c->stage_in_cap = len + BROTLI_AMD_INPUT_SLACK;
    }
    if (max_out > c->stage_out_cap) {
      if (c->d_stage_out) HIP_OK(c, hipFree(c->d_stage_out));
      c->d_stage_out = nullptr; c->stage_out_cap = 0;
      HIP_OK(c, hipMalloc((void**)&c->d_stage_out, max_out));
      c->stage_out_cap = max_out;
    }
    if (len && !big_copy(c, c->d_stage_in, in, len, true)) return fail(c, "H2D copy failed");
    HIP_OK(c, hipMemsetAsync(c->d_stage_in + len, 0, BROTLI_AMD_INPUT_SLACK, c->stream));
    return true;
  };
  if (!stage()) return BROTLI_AMD_ERROR;
  uint64_t nbits = 0;
  int rc = brotli_amd_encode_fast_device(c, c->d_stage_in, len, call_sizes, ncalls, p, c->d_stage_out,
                                         c->stage_out_cap, &nbits, info);
  if (rc != BROTLI_AMD_OK) return rc;
  const uint64_t n = (nbits + 7) / 8;
  *out_bits = nbits;
  if (n > out_cap) { c->err = "output capacity too small"; return BROTLI_AMD_OVERFLOW; }
  if (n && !big_copy(c, out, c->d_stage_out, n, false)) {
    fail(c, "D2H copy failed");
    return BROTLI_AMD_ERROR;
  }
  return BROTLI_AMD_OK;
}

// ---- incremental single-shard stream ------------------------------------------------
}  // extern "C"

struct BrotliAmdStream {
  BrotliAmdCtx* c = nullptr;
  JobParams J;
  ShardDesc D;
  uint8_t* d_in = nullptr;
  uint64_t in_cap = 0;
  uint8_t* d_ws = nullptr;
  uint8_t* d_out = nullptr;
  uint64_t out_cap = 0;
  ShardDesc* d_desc = nullptr;
  ShardState* d_state = nullptr;
  uint32_t* d_counters = nullptr;
  uint64_t fed = 0;
  uint64_t flushed = 0;             // `fed` when the last FLUSH / FINISH completed: nothing older is waiting for a meta-block
  bool finished = false;
  std::vector<uint8_t> host_out;
  // attached dictionaries (k_dict.h): the device copies of the chunks and of the CompoundDict
  std::vector<void*> dict_allocs;
  CompoundDict* d_cd = nullptr;
};

namespace {
bool stream_init(BrotliAmdStream* s, uint32_t stream_offset) {
  BrotliAmdCtx* c = s->c;
  const JobParams& J = s->J;
  const uint64_t mb = J.max_metablock_size;
  // (the FastLog2 table grows with the bytes actually fed, stream_run: a histogram cannot count more
  // than the stream holds, and most streams are far shorter than a full meta-block of 8 - 16 Mi entries)
  ShardDesc& D = s->D;
  memset(&D, 0, sizeof(D));
  uint64_t so = stream_offset;
  if (so > (1u << 30)) so = 1u << 30;
  if (so > J.max_backward_limit) so = J.max_backward_limit;
  D.stream_offset = (uint32_t)so;
  D.cmd_cap = (uint32_t)(mb / 2 + (mb >> J.lgblock) + 64);
  uint64_t off = 0;
  D.table_off = off; off = plan_align(off + ((uint64_t)J.rec_bytes << J.bucket_bits));
  D.num_off = off;   off = plan_align(off + ((J.flags & JOB_FLAG_DEEP) ? ((uint64_t)2 << J.bucket_bits) : 0));
  D.cmds_off = off;  off = plan_align(off + (uint64_t)D.cmd_cap * sizeof(Command));
  D.lits_off = off;  off = plan_align(off + (mb + 8) * 2);
  D.dsym_off = off;  off = plan_align(off + (uint64_t)D.cmd_cap * 2);
  D.mb_off = off;    off = plan_align(off + mb_work_bytes(mb));
  D.scratch_off = off; off = plan_align(off + (mb / 256 + 64) * 8 + (2 * mb + 64) * 4);
  HIP_OK(c, hipMalloc((void**)&s->d_ws, off));
  HIP_OK(c, hipMalloc((void**)&s->d_desc, sizeof(ShardDesc)));
  HIP_OK(c, hipMalloc((void**)&s->d_state, sizeof(ShardState)));
  HIP_OK(c, hipMalloc((void**)&s->d_counters, 16 * sizeof(uint32_t)));
  HIP_OK(c, hipMemcpyAsync(s->d_desc, &D, sizeof(D), hipMemcpyHostToDevice, c->stream));
  JobArgs a;
  a.J = J;
  a.shards = s->d_desc;
  a.states = s->d_state;
  a.T = c->d_T;
  a.input = nullptr;
  a.ws = s->d_ws;
  a.nshards = 1;
  a.init_blocks_per_shard = 64;
  a.counters = s->d_counters;
  hipLaunchKernelGGL(k_init, dim3(64), dim3(256), 0, c->stream, a);
  HIP_OK(c, hipStreamSynchronize(c->stream));
  HIP_OK(c, hipGetLastError());
  return true;
}

bool stream_run(BrotliAmdStream* s, const uint8_t* data, uint64_t len, int op) {
  BrotliAmdCtx* c = s->c;
  const JobParams& J = s->J;
  if (s->fed + len >= (3ull << 30)) return fail(c, "stream longer than 3 GiB is not supported");
  {
    const uint64_t need = std::min<uint64_t>(J.max_metablock_size, s->fed + len) + 2;
    if (need > c->log2_n) {
      uint64_t n = 1u << 16;
      while (n < need) n <<= 1;
      if (!ensure_log2(c, (uint32_t)std::min<uint64_t>(n, (uint64_t)J.max_metablock_size + 2))) return false;
    }
  }
  // input: the whole stream stays resident (positions are stream offsets)
  const uint64_t need_in = s->fed + len + BROTLI_AMD_INPUT_SLACK;
  if (need_in > s->in_cap) {
    uint64_t cap = s->in_cap ? s->in_cap * 2 : (1u << 20);
    while (cap < need_in) cap *= 2;
    uint8_t* n = nullptr;
    HIP_OK(c, hipMalloc((void**)&n, cap));
    if (s->d_in) {
      HIP_OK(c, hipMemcpyAsync(n, s->d_in, s->fed, hipMemcpyDeviceToDevice, c->stream));
      HIP_OK(c, hipStreamSynchronize(c->stream));
      HIP_OK(c, hipFree(s->d_in));
    }
    s->d_in = n;
    s->in_cap = cap;
  }
  if (len) HIP_OK(c, hipMemcpyAsync(s->d_in + s->fed, data, len, hipMemcpyHostToDevice, c->stream));
  HIP_OK(c, hipMemsetAsync(s->d_in + s->fed + len, 0, BROTLI_AMD_INPUT_SLACK, c->stream));
  s->fed += len;
  // output of this call: everything not yet emitted can come out at once
  const uint64_t need_out = 2 * (len + (uint64_t)J.max_metablock_size + (2ull << J.lgblock)) + 8192;
  if (need_out > s->out_cap) {
    if (s->d_out) HIP_OK(c, hipFree(s->d_out));
    s->d_out = nullptr;
    HIP_OK(c, hipMalloc((void**)&s->d_out, need_out));
    s->out_cap = need_out;
  }
  ShardDesc& D = s->D;
  D.in_off = 0;
  D.len = (uint32_t)s->fed;
  D.final_op = (uint32_t)op;
  D.out_off = (uint64_t)(s->d_out - s->d_ws);   // ws + out_off == d_out (mod 2^64)
  D.out_cap = s->out_cap;
  HIP_OK(c, hipMemcpyAsync(s->d_desc, &D, sizeof(D), hipMemcpyHostToDevice, c->stream));
  const uint32_t zero32 = 0;
  const uint64_t zero64 = 0;
  HIP_OK(c, hipMemcpyAsync(&s->d_state->done, &zero32, 4, hipMemcpyHostToDevice, c->stream));
  HIP_OK(c, hipMemcpyAsync(&s->d_state->out_bytes, &zero64, 8, hipMemcpyHostToDevice, c->stream));
  JobArgs a;
  a.J = J;
  a.shards = s->d_desc;
  a.states = s->d_state;
  a.T = c->d_T;
  a.input = s->d_in;
  a.ws = s->d_ws;
  a.nshards = 1;
  a.init_blocks_per_shard = 1;
  a.counters = s->d_counters;
  a.cd = s->d_cd;
  for (uint64_t round = 0;; ++round) {
    if (round > (s->fed >> 10) + 64) return fail(c, "stream rounds do not converge (device fault)");
    HIP_OK(c, hipMemsetAsync(s->d_counters, 0, 16 * sizeof(uint32_t), c->stream));
    if (J.flags & JOB_FLAG_QUICK) hipLaunchKernelGGL(k_parse_quick, dim3(1), dim3(64), 0, c->stream, a);
    else if (!(J.flags & JOB_FLAG_DEEP)) hipLaunchKernelGGL(k_parse, dim3(1), dim3(64), 0, c->stream, a);
    else if (J.block_bits <= 6) hipLaunchKernelGGL(k_parse_deep<1>, dim3(1), dim3(64), 0, c->stream, a);
    else if (J.block_bits == 7) hipLaunchKernelGGL(k_parse_deep<2>, dim3(1), dim3(64), 0, c->stream, a);
    else hipLaunchKernelGGL(k_parse_deep<4>, dim3(1), dim3(64), 0, c->stream, a);
    {
      HipRun R{c->stream};
      // (the many-wave build / store for what can actually be pending — a FLUSH of a few hundred bytes keeps the two
      //  one-wave kernels)
      run_build_store(R, a, 1u, use_wide(J, std::min<uint64_t>(J.max_metablock_size, s->fed - s->flushed)), []() {});
    }
    uint32_t counters[16];
    HIP_OK(c, hipMemcpyAsync(counters, s->d_counters, sizeof(counters), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(c, hipStreamSynchronize(c->stream));
    HIP_OK(c, hipGetLastError());
    if (counters[1]) return fail(c, "stream shard reported a device fault");
    if (counters[0] == 0) break;
  }
  ShardState st;
  HIP_OK(c, hipMemcpy(&st, s->d_state, sizeof(st), hipMemcpyDeviceToHost));
  if (st.error) return fail(c, "stream shard error %u", st.error);
  s->host_out.resize(st.out_bytes);
  if (st.out_bytes) HIP_OK(c, hipMemcpy(s->host_out.data(), s->d_out, st.out_bytes, hipMemcpyDeviceToHost));
  return true;
}
}  // namespace

extern "C" {

int brotli_amd_stream_create(BrotliAmdCtx* c, int quality, int lgwin, uint32_t size_hint,
                             uint32_t stream_offset, uint32_t flags, BrotliAmdStream** out) {
  *out = nullptr;
  DeviceScope dev(c->device);
  if (!dev.ok) { fail(c, "hipSetDevice failed"); return BROTLI_AMD_ERROR; }
  BrotliAmdStream* s = new BrotliAmdStream();
  s->c = c;
  if (!plan_params(quality, lgwin, size_hint, &s->J, (int)((flags >> BROTLI_AMD_FLAG_LGBLOCK_SHIFT) & 31u))) {
    delete s;
    fail(c, "parameters outside the GPU path (quality %d lgwin %d)", quality, lgwin);
    return BROTLI_AMD_UNSUPPORTED;
  }
  if (quality != 5) s->J.flags |= JOB_FLAG_DEEP;   // k_parse_deep.h; k_parse.h serves the 16-slot hashers
  if (flags & BROTLI_AMD_FLAG_NO_HEADER) s->J.flags |= JOB_FLAG_NO_HEADER;
  if (flags & BROTLI_AMD_FLAG_NO_LITERAL_CONTEXT) s->J.flags |= JOB_FLAG_NO_LITCTX;
  s->J.log2_lut_size = s->J.max_metablock_size + 2;
  if (!stream_init(s, stream_offset)) { brotli_amd_stream_destroy(s); return BROTLI_AMD_ERROR; }
  *out = s;
  return BROTLI_AMD_OK;
}

int brotli_amd_stream_write(BrotliAmdStream* s, const uint8_t* data, uint64_t len, int op,
                            const uint8_t** out, uint64_t* out_len) {
  *out = nullptr;
  *out_len = 0;
  BrotliAmdCtx* c = s->c;
  DeviceScope dev(c->device);
  if (!dev.ok) { fail(c, "hipSetDevice failed"); return BROTLI_AMD_ERROR; }
  if (s->finished) { fail(c, "stream already finished"); return BROTLI_AMD_ERROR; }
  if (op < 0 || op > 3) { fail(c, "bad stream op"); return BROTLI_AMD_UNSUPPORTED; }
  if (!stream_run(s, data, len, op)) return BROTLI_AMD_ERROR;
  if (op != BROTLI_AMD_OP_PROCESS) s->flushed = s->fed;
  if (op == BROTLI_AMD_OP_FINISH) s->finished = true;
  *out = s->host_out.data();
  *out_len = s->host_out.size();
  return BROTLI_AMD_OK;
}

int brotli_amd_stream_attach_dictionary(BrotliAmdStream* s, const BrotliAmdDictChunk* chunks, uint32_t nchunks) {
  BrotliAmdCtx* c = s->c;
  DeviceScope dev(c->device);
  if (!dev.ok) { fail(c, "hipSetDevice failed"); return BROTLI_AMD_ERROR; }
  return upload_dictionary(c, chunks, nchunks, &s->dict_allocs, &s->d_cd);
}

int brotli_amd_ctx_set_dictionary(BrotliAmdCtx* c, const BrotliAmdDictChunk* chunks, uint32_t nchunks) {
  DeviceScope dev(c->device);
  if (!dev.ok) { fail(c, "hipSetDevice failed"); return BROTLI_AMD_ERROR; }
  return upload_dictionary(c, chunks, nchunks, &c->dict_allocs, &c->d_cd);
}

int brotli_amd_stream_take_partial(BrotliAmdStream* s, uint32_t* nbits, uint32_t* value) {
  BrotliAmdCtx* c = s->c;
  *nbits = 0;
  *value = 0;
  DeviceScope dev(c->device);
  if (!dev.ok) { fail(c, "hipSetDevice failed"); return BROTLI_AMD_ERROR; }
  uint32_t lb[2] = {0, 0};   // last_bytes, last_bytes_bits are adjacent in ShardState
  static_assert(offsetof(ShardState, last_bytes_bits) == offsetof(ShardState, last_bytes) + 4, "layout");
  if (hipMemcpy(lb, &s->d_state->last_bytes, 8, hipMemcpyDeviceToHost) != hipSuccess) { fail(c, "state read failed"); return BROTLI_AMD_ERROR; }
  const uint32_t zero[2] = {0, 0};
  if (hipMemcpy(&s->d_state->last_bytes, zero, 8, hipMemcpyHostToDevice) != hipSuccess) { fail(c, "state write failed"); return BROTLI_AMD_ERROR; }
  *value = lb[0];
  *nbits = lb[1];
  return BROTLI_AMD_OK;
}

void brotli_amd_stream_destroy(BrotliAmdStream* s) {
  if (!s) return;
  DeviceScope dev(s->c->device);
  (void)hipStreamSynchronize(s->c->stream);
  void* ptrs[] = {s->d_in, s->d_ws, s->d_out, s->d_desc, s->d_state, s->d_counters};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  for (void* p : s->dict_allocs) (void)hipFree(p);
  delete s;
}

// ---- decoder ---------------------------------------------------------------------------------
static_assert(sizeof(BrotliAmdDecodePiece) == sizeof(DecPiece) && sizeof(BrotliAmdDecodeResult) == sizeof(DecResult),
              "the C ABI structs are the kernel's");
static_assert(sizeof(DecTransform) == 8, "transform records are 8 bytes");

int brotli_amd_decode_device(BrotliAmdCtx* c, const void* d_in, uint64_t in_len,
                             const BrotliAmdDecodePiece* pieces, uint64_t npieces, void* d_out,
                             uint64_t out_cap, BrotliAmdDecodeResult* results, float* ms) {
  if (ms) *ms = 0.f;
  DeviceScope dev(c->device);
  if (!dev.ok) { fail(c, "hipSetDevice failed"); return BROTLI_AMD_ERROR; }
  if (npieces == 0) return BROTLI_AMD_OK;
  if (npieces > (1u << 22)) { fail(c, "too many pieces"); return BROTLI_AMD_UNSUPPORTED; }
  for (uint64_t k = 0; k < npieces; ++k) {
    const BrotliAmdDecodePiece& p = pieces[k];
    if (p.in_len > in_len || p.in_off > in_len - p.in_len || p.out_cap > out_cap || p.out_off > out_cap - p.out_cap ||
        (!(p.flags & BROTLI_AMD_PIECE_HEADER) && (p.lgwin < 10 || p.lgwin > 24))) {
      fail(c, "piece %llu does not fit its buffers", (unsigned long long)k);
      return BROTLI_AMD_UNSUPPORTED;
    }
  }
  auto body = [&]() -> bool {
    if (!c->d_transforms) {
      if (!host_transforms_load(c->tables_path.c_str(), &c->htr)) return fail(c, "cannot load brotli_transforms.bin next to %s", c->tables_path.c_str());
      if (!dev_upload(c, &c->d_transforms, c->htr.records.data(), c->htr.records.size())) return false;
      if (!dev_upload(c, &c->d_transform_text, c->htr.text.data(), c->htr.text.size())) return false;
    }
    // arena per piece: everything a meta-block can ask for while the job is small, 48 Ki dwords (the
    // encoder kernels' streams need a fraction of that) once thousands of pieces share the memory
    const uint32_t full = dec_arena_words_max();
    uint64_t words = full;
    if (npieces * (uint64_t)full * 4u > (8ull << 30)) words = 48u << 10;
    if (npieces * words > c->dec_arena_cap) {
      if (c->d_dec_arena) HIP_OK(c, hipFree(c->d_dec_arena));
      c->d_dec_arena = nullptr;
      c->dec_arena_cap = 0;
      HIP_OK(c, hipMalloc((void**)&c->d_dec_arena, npieces * words * 4u));
      c->dec_arena_cap = npieces * words;
    }
    if (npieces > c->dec_piece_cap) {
      if (c->d_dec_pieces) HIP_OK(c, hipFree(c->d_dec_pieces));
      if (c->d_dec_results) HIP_OK(c, hipFree(c->d_dec_results));
      c->d_dec_pieces = nullptr;
      c->d_dec_results = nullptr;
      c->dec_piece_cap = 0;
      HIP_OK(c, hipMalloc((void**)&c->d_dec_pieces, npieces * sizeof(DecPiece)));
      HIP_OK(c, hipMalloc((void**)&c->d_dec_results, npieces * sizeof(DecResult)));
      c->dec_piece_cap = npieces;
    }
    HIP_OK(c, hipMemcpyAsync(c->d_dec_pieces, pieces, npieces * sizeof(DecPiece), hipMemcpyHostToDevice, c->stream));
    HIP_OK(c, hipMemsetAsync(c->d_dec_results, 0xFF, npieces * sizeof(DecResult), c->stream));
    DecArgs a;
    a.pieces = c->d_dec_pieces;
    a.results = c->d_dec_results;
    a.T = c->d_T;
    a.transforms = c->d_transforms;
    a.transform_text = c->d_transform_text;
    a.input = (const uint8_t*)d_in;
    a.out = (uint8_t*)d_out;
    a.arena = c->d_dec_arena;
    a.arena_words = (uint32_t)words;
    a.npieces = (uint32_t)npieces;
    // measurement switches (tools/gpu_decode_variants.py): BROTLI_AMD_DECODE_VARIANT = waves per SIMD (4 / 8),
    // + 16 = no LDS cache
    const int variant = g_env.decode_variant >= 0 ? g_env.decode_variant : DECODE_WAVES;
    a.flags = (variant & 16) ? DEC_ARG_NO_LDS_CACHE : 0u;
    a.pad = 0;
    HIP_OK(c, hipEventRecord(c->ev[0], c->stream));
    if ((variant & 15) == 8) hipLaunchKernelGGL(k_decode<8>, dim3((uint32_t)npieces), dim3(64), 0, c->stream, a);
    else hipLaunchKernelGGL(k_decode<4>, dim3((uint32_t)npieces), dim3(64), 0, c->stream, a);
    HIP_OK(c, hipEventRecord(c->ev[1], c->stream));
    HIP_OK(c, hipMemcpyAsync(results, c->d_dec_results, npieces * sizeof(DecResult), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(c, hipStreamSynchronize(c->stream));
    HIP_OK(c, hipGetLastError());
    if (ms) HIP_OK(c, hipEventElapsedTime(ms, c->ev[0], c->ev[1]));
    return true;
  };
  if (!body()) return BROTLI_AMD_ERROR;
  for (uint64_t k = 0; k < npieces; ++k) {
    if (results[k].error) {
      fail(c, "piece %llu: decoder error %u", (unsigned long long)k, results[k].error);
      return BROTLI_AMD_DEVICE_FAULT;
    }
  }
  return BROTLI_AMD_OK;
}

int brotli_amd_decode_host(BrotliAmdCtx* c, const uint8_t* in, uint64_t in_len,
                           const BrotliAmdDecodePiece* pieces, uint64_t npieces, uint8_t* out,
                           uint64_t out_cap, BrotliAmdDecodeResult* results, float* ms) {
  DeviceScope dev(c->device);
  if (!dev.ok) { fail(c, "hipSetDevice failed"); return BROTLI_AMD_ERROR; }
  uint8_t *d_in = nullptr, *d_out = nullptr;
  int rc = BROTLI_AMD_ERROR;
  auto body = [&]() -> bool {
    HIP_OK(c, hipMalloc((void**)&d_in, in_len + BROTLI_AMD_DECODE_SLACK));
    HIP_OK(c, hipMalloc((void**)&d_out, out_cap + 64));
    if (in_len) HIP_OK(c, hipMemcpy(d_in, in, in_len, hipMemcpyHostToDevice));
    HIP_OK(c, hipMemset(d_in + in_len, 0, BROTLI_AMD_DECODE_SLACK));
    rc = brotli_amd_decode_device(c, d_in, in_len, pieces, npieces, d_out, out_cap, results, ms);
    if (rc == BROTLI_AMD_OK || rc == BROTLI_AMD_DEVICE_FAULT) {
      if (out_cap) HIP_OK(c, hipMemcpy(out, d_out, out_cap, hipMemcpyDeviceToHost));
    }
    return true;
  };
  const bool ok = body();
  if (d_in) (void)hipFree(d_in);
  if (d_out) (void)hipFree(d_out);
  return ok ? rc : BROTLI_AMD_ERROR;
}

int brotli_amd_debug_parse(BrotliAmdCtx* c, const void* d_in, uint64_t len,
                           const BrotliAmdJobParams* p, void* h_cmds, uint64_t cmd_cap,
                           uint64_t* ncmds, BrotliAmdJobInfo* info) {
  BrotliAmdJobInfo local;
  if (!info) info = &local;
  memset(info, 0, sizeof(*info));
  *ncmds = 0;
  DeviceScope dev(c->device);
  if (!dev.ok) { fail(c, "hipSetDevice failed"); return BROTLI_AMD_ERROR; }
  JobPlan plan;
  int rc = plan_from_params(c, len, p, &plan);
  if (rc != BROTLI_AMD_OK) return rc;
  std::vector<ShardState> st;
  if (!run_rounds(c, plan, (const uint8_t*)d_in, STAGE_PARSE, info, &st)) return BROTLI_AMD_ERROR;
  uint64_t n = 0;
  Command* dst = (Command*)h_cmds;
  for (size_t k = 0; k < st.size(); ++k) {
    const uint64_t m = st[k].ncmds;
    info->searches += st[k].stat_searches;
    info->search_steps += st[k].stat_pairs;
    info->exact_searches += st[k].ix_slow;
    for (int i = 0; i < 12; ++i) info->prof[i] += st[k].prof[i];
    if (n + m <= cmd_cap && m) {
      if (hipMemcpy(dst + n, c->d_ws + plan.shards[k].cmds_off, m * sizeof(Command),
                    hipMemcpyDeviceToHost) != hipSuccess) {
        fail(c, "command download failed");
        return BROTLI_AMD_ERROR;
      }
    }
    n += m;
  }
  info->commands = n;
  *ncmds = n;
  return n <= cmd_cap ? BROTLI_AMD_OK : BROTLI_AMD_OVERFLOW;
}

}  // extern "C"
