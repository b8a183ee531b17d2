// tests/simt/sim_hip_layer.cc — TEST ONLY: the subset of include/brotli_amd_hip.h that
// brotli_amd/csrc/encode_abi.c calls, implemented on the host SIMT simulator, so that the
// BrotliEncoder* boundary (parameter latching, size hints, call lists, flush / metadata
// hand-off, routing to plan / stream / quality-1 jobs) can be driven with the reference's call
// sequences next to the reference library WITHOUT a GPU.  Built into
// tests/simt/libbrotlienc_sim.so together with encode_abi.c; never shipped, never loaded by
// the product (which fails loudly without libbrotli_amd_hip.so and a gfx950 device).
#include "sim_driver.cc"

#include <string>

#include "../../include/brotli_amd_hip.h"

struct SimDict {
  std::vector<std::vector<uint8_t>> src;
  std::vector<std::vector<uint32_t>> starts, items;
  CompoundDict cd;
  bool have = false;
  int set(const BrotliAmdDictChunk* chunks, uint32_t nchunks) {
    if (nchunks > DICT_MAX_CHUNKS) return BROTLI_AMD_UNSUPPORTED;
    src.assign(nchunks, {});
    starts.assign(nchunks, {});
    items.assign(nchunks, {});
    memset(&cd, 0, sizeof(cd));
    uint32_t total = 0;
    for (uint32_t k = 0; k < nchunks; ++k) {
      const BrotliAmdDictChunk& h = chunks[k];
      const size_t nkeys = (size_t)1 << h.bucket_bits;
      src[k].assign(h.source, h.source + h.source_size);
      src[k].resize(h.source_size + DICT_SOURCE_SLACK, 0);
      starts[k].assign(h.starts, h.starts + nkeys + 1);
      items[k].assign(h.items, h.items + h.starts[nkeys]);
      items[k].push_back(0);
      DictChunk& g = cd.chunks[k];
      g.source = src[k].data();
      g.starts = starts[k].data();
      g.items = items[k].data();
      g.source_size = h.source_size;
      g.bucket_bits = h.bucket_bits;
      g.offset = total;
      total += h.source_size;
    }
    cd.num_chunks = nchunks;
    cd.total_size = total;
    have = nchunks != 0;
    return BROTLI_AMD_OK;
  }
};

struct BrotliAmdCtx {
  std::string tables, err;
  HostTables ht;
  SimDict dict;       // brotli_amd_ctx_set_dictionary
};

struct BrotliAmdStream {
  BrotliAmdCtx* c = nullptr;
  JobParams J;
  ShardDesc D;
  ShardState state;
  std::vector<uint8_t> ws, input, outbuf, host_out;
  std::vector<double> lut;
  DeviceTables T;
  uint32_t counters[16];
  uint64_t fed = 0;
  bool finished = false;
  SimDict dict;
};

namespace {
int set_err(BrotliAmdCtx* c, const char* msg, int rc) { c->err = msg; return rc; }

// sim_encode's body for a finished plan (tables carved out of the workspace)
long run_plan_on_sim(BrotliAmdCtx* c, JobPlan& plan, const uint8_t* in, size_t len, uint8_t* out, size_t out_cap) {
  std::vector<uint8_t> input(len + 64, 0);
  memcpy(input.data(), in, len);
  std::vector<uint8_t> ws(plan.ws_bytes, 0xCD);
  std::vector<ShardState> states(plan.shards.size());
  std::vector<double> log2lut;
  DeviceTables T;
  host_tables_fill(c->ht, plan.J.log2_lut_size, &log2lut, &T);
  JobArgs a;
  a.J = plan.J;
  a.shards = plan.shards.data();
  a.states = states.data();
  a.T = &T;
  a.input = input.data();
  a.ws = ws.data();
  a.nshards = (uint32_t)plan.shards.size();
  a.init_blocks_per_shard = 2;
  uint32_t counters[16] = {0};
  a.counters = counters;
  a.cd = c->dict.have ? &c->dict.cd : nullptr;
  run(k_init, a, a.nshards * a.init_blocks_per_shard, 256, 0);
  for (int round = 0; round < 100000; ++round) {
    memset(counters, 0, sizeof(counters));
    if (plan.J.flags & JOB_FLAG_QUICK) {
      run(k_parse_quick, a, a.nshards, 64, 0);
    } else if (plan.J.flags & JOB_FLAG_DEEP) {
      if (plan.J.block_bits <= 6) run(k_parse_deep<1>, a, a.nshards, 64, 0);
      else if (plan.J.block_bits == 7) run(k_parse_deep<2>, a, a.nshards, 64, 0);
      else run(k_parse_deep<4>, a, a.nshards, 64, 0);
    } else if (plan.J.flags & JOB_FLAG_QUAD) {
      run(k_parse4, a, (a.nshards + q_groups_per_wave(a.J) - 1) / q_groups_per_wave(a.J), 64, 0);
    } else {
      run(k_parse, a, a.nshards, 64, 0);
    }
    {
      SimRun R{0};
      run_build_store(R, a, a.nshards, sim_wide(), []() {});
    }
    if (counters[1]) return -3;
    if (counters[0] == 0) break;
  }
  size_t n = 0;
  for (size_t k = 0; k < plan.shards.size(); ++k) {
    const uint64_t m = states[k].out_bytes;
    if (n + m > out_cap) return -4;
    memcpy(out + n, ws.data() + plan.shards[k].out_off, m);
    n += m;
  }
  return (long)n;
}
}  // namespace

extern "C" {

int brotli_amd_ctx_create(int device, const char* tables_path, BrotliAmdCtx** out) {
  (void)device;
  BrotliAmdCtx* c = new BrotliAmdCtx();
  *out = c;
  c->tables = tables_path;
  if (!host_tables_load(tables_path, &c->ht)) return set_err(c, "cannot load format tables", BROTLI_AMD_ERROR);
  return BROTLI_AMD_OK;
}
void brotli_amd_ctx_destroy(BrotliAmdCtx* c) { delete c; }
const char* brotli_amd_last_error(const BrotliAmdCtx* c) { return c ? c->err.c_str() : "no context"; }

uint64_t brotli_amd_max_output(uint64_t len, const BrotliAmdJobParams* p) {
  JobPlan plan;
  if (len == 0) return 16;
  if (!plan_job(len, p->quality, p->lgwin, p->size_hint, p->shard_size, p->stream_base, p->is_last != 0, &plan, true,
                (int)((p->flags >> BROTLI_AMD_FLAG_LGBLOCK_SHIFT) & 31u))) return 0;
  return plan.max_out_bytes;
}

int brotli_amd_encode_host(BrotliAmdCtx* c, const uint8_t* in, uint64_t len, const BrotliAmdJobParams* p,
                           uint8_t* out, uint64_t out_cap, uint64_t* out_size, BrotliAmdJobInfo* info) {
  *out_size = 0;
  if (info) memset(info, 0, sizeof(*info));
  if (p->flags & BROTLI_AMD_FLAG_STREAM_TILES) {
    // (hip_layer.hip: run_stream_job)
    if (p->quality != 5 || p->shard_size != 0 || p->stream_base != 0 || !p->is_last || c->dict.have ||
        ((p->flags >> BROTLI_AMD_FLAG_LGBLOCK_SHIFT) & 31u) != 0u)
      return set_err(c, "BROTLI_AMD_FLAG_STREAM_TILES: quality 5, one whole stream, no dictionary, the default block size", BROTLI_AMD_UNSUPPORTED);
    uint32_t sinfo[4] = {0, 0, 0, 0};
    int jflags = 0;
    if (p->flags & BROTLI_AMD_FLAG_NO_LITERAL_CONTEXT) jflags |= (int)JOB_FLAG_NO_LITCTX;
    if (p->flags & BROTLI_AMD_FLAG_NO_HEADER) jflags |= (int)JOB_FLAG_NO_HEADER;
    if (p->flags & BROTLI_AMD_FLAG_TAIL_FINISH) jflags |= (int)JOB_FLAG_TAILFIN;
    const long n = sim_encode_stream(c->tables.c_str(), in, (size_t)len, p->lgwin, p->size_hint, 0, jflags, out, (size_t)out_cap, sinfo);
    if (info) info->reserved = sinfo[1] | (sinfo[2] << 8) | ((sinfo[0] >> 8) << 16);
    if (n == -10 || n == -2) return set_err(c, "the stream left the tiled path", BROTLI_AMD_SERIAL);
    if (n == -4) return set_err(c, "output capacity too small", BROTLI_AMD_OVERFLOW);
    if (n < 0) return set_err(c, "device fault", BROTLI_AMD_DEVICE_FAULT);
    *out_size = (uint64_t)n;
    return BROTLI_AMD_OK;
  }
  JobPlan plan;
  if (len == 0 || !plan_job(len, p->quality, p->lgwin, p->size_hint, p->shard_size, p->stream_base,
                            p->is_last != 0, &plan, true, (int)((p->flags >> BROTLI_AMD_FLAG_LGBLOCK_SHIFT) & 31u)))
    return set_err(c, "parameters outside the GPU path", BROTLI_AMD_UNSUPPORTED);
  uint32_t lim = 0;
  // (a dictionary on the context: the hash-table kernels, no scout groups, as hip_layer.hip does)
  if (!plan_choose_kernels(&plan, p->flags, 256, &lim)) return set_err(c, "shard too long", BROTLI_AMD_UNSUPPORTED);
  if (c->dict.have) plan.J.flags &= ~(uint32_t)JOB_FLAG_DUO;
  const long n = run_plan_on_sim(c, plan, in, (size_t)len, out, (size_t)out_cap);
  if (n == -4) return set_err(c, "output capacity too small", BROTLI_AMD_OVERFLOW);
  if (n < 0) return set_err(c, "device fault", BROTLI_AMD_DEVICE_FAULT);
  *out_size = (uint64_t)n;
  return BROTLI_AMD_OK;
}

uint64_t brotli_amd_fast_max_output(uint64_t len, uint64_t ncalls, int lgwin) {
  if (lgwin < 10 || lgwin > 24) return 0;
  return len + 8 * (ncalls + (len >> lgwin) + 1) + 64;
}

int brotli_amd_encode_fast_host(BrotliAmdCtx* c, const uint8_t* in, uint64_t len, const uint64_t* call_sizes,
                                uint64_t ncalls, const BrotliAmdFastParams* p, uint8_t* out, uint64_t out_cap,
                                uint64_t* out_bits, BrotliAmdJobInfo* info) {
  *out_bits = 0;
  if (info) memset(info, 0, sizeof(*info));
  const long nbits = sim_encode_fast(c->tables.c_str(), in, (size_t)len, p->lgwin, call_sizes, (size_t)ncalls,
                                     p->carry_bits, p->carry_value, p->is_last, 0, out, (size_t)out_cap);
  if (nbits == -4) return set_err(c, "output capacity too small", BROTLI_AMD_OVERFLOW);
  if (nbits < 0) return set_err(c, "quality 1 job failed", nbits == -2 ? BROTLI_AMD_UNSUPPORTED : BROTLI_AMD_DEVICE_FAULT);
  *out_bits = (uint64_t)nbits;
  return BROTLI_AMD_OK;
}

// ---- the incremental stream (hip_layer.hip stream_init / stream_run on the simulator) ----
int brotli_amd_stream_create(BrotliAmdCtx* c, int quality, int lgwin, uint32_t size_hint, uint32_t stream_offset,
                             uint32_t flags, BrotliAmdStream** out) {
  *out = nullptr;
  BrotliAmdStream* s = new BrotliAmdStream();
  s->c = c;
  if (!plan_params(quality, lgwin, size_hint, &s->J, (int)((flags >> BROTLI_AMD_FLAG_LGBLOCK_SHIFT) & 31u))) { delete s; return set_err(c, "parameters outside the GPU path", BROTLI_AMD_UNSUPPORTED); }
  JobParams& J = s->J;
  if (quality != 5) J.flags |= JOB_FLAG_DEEP;
  if (flags & BROTLI_AMD_FLAG_NO_HEADER) J.flags |= JOB_FLAG_NO_HEADER;
  if (flags & BROTLI_AMD_FLAG_NO_LITERAL_CONTEXT) J.flags |= JOB_FLAG_NO_LITCTX;
  const uint64_t mb = J.max_metablock_size;
  J.log2_lut_size = (uint32_t)(mb + 2);
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
  s->ws.assign(off, 0xCD);
  host_tables_fill(c->ht, J.log2_lut_size, &s->lut, &s->T);
  JobArgs a;
  a.J = J;
  a.shards = &s->D;
  a.states = &s->state;
  a.T = &s->T;
  a.input = nullptr;
  a.ws = s->ws.data();
  a.nshards = 1;
  a.init_blocks_per_shard = 64;
  a.counters = s->counters;
  run(k_init, a, 64, 256, 0);
  *out = s;
  return BROTLI_AMD_OK;
}

int brotli_amd_stream_write(BrotliAmdStream* s, const uint8_t* data, uint64_t len, int op, const uint8_t** out,
                            uint64_t* out_len) {
  *out = nullptr;
  *out_len = 0;
  BrotliAmdCtx* c = s->c;
  const JobParams& J = s->J;
  if (s->finished) return set_err(c, "stream already finished", BROTLI_AMD_ERROR);
  if (op < 0 || op > 3) return set_err(c, "bad stream op", BROTLI_AMD_UNSUPPORTED);
  s->input.resize(s->fed + len + 64);
  if (len) memcpy(s->input.data() + s->fed, data, len);
  memset(s->input.data() + s->fed + len, 0, 64);
  s->fed += len;
  s->outbuf.assign(2 * (len + (uint64_t)J.max_metablock_size + (2ull << J.lgblock)) + 8192, 0xCD);
  ShardDesc& D = s->D;
  D.in_off = 0;
  D.len = (uint32_t)s->fed;
  D.final_op = (uint32_t)op;
  D.out_off = (uint64_t)(s->outbuf.data() - s->ws.data());   // ws + out_off == outbuf (mod 2^64)
  D.out_cap = s->outbuf.size();
  s->state.done = 0;
  s->state.out_bytes = 0;
  JobArgs a;
  a.J = J;
  a.shards = &s->D;
  a.states = &s->state;
  a.T = &s->T;
  a.input = s->input.data();
  a.ws = s->ws.data();
  a.nshards = 1;
  a.init_blocks_per_shard = 1;
  a.counters = s->counters;
  a.cd = s->dict.have ? &s->dict.cd : nullptr;
  for (uint64_t round = 0;; ++round) {
    if (round > (s->fed >> 10) + 64) return set_err(c, "stream rounds do not converge (device fault)", BROTLI_AMD_ERROR);
    memset(s->counters, 0, sizeof(s->counters));
    if (J.flags & JOB_FLAG_QUICK) run(k_parse_quick, a, 1, 64, 0);
    else if (!(J.flags & JOB_FLAG_DEEP)) run(k_parse, a, 1, 64, 0);
    else if (J.block_bits <= 6) run(k_parse_deep<1>, a, 1, 64, 0);
    else if (J.block_bits == 7) run(k_parse_deep<2>, a, 1, 64, 0);
    else run(k_parse_deep<4>, a, 1, 64, 0);
    {
      SimRun R{0};
      run_build_store(R, a, 1, sim_wide(), []() {});
    }
    if (s->counters[1]) return set_err(c, "stream shard reported a device fault", BROTLI_AMD_ERROR);
    if (s->counters[0] == 0) break;
  }
  if (s->state.error) return set_err(c, "stream shard error", BROTLI_AMD_ERROR);
  s->host_out.assign(s->outbuf.begin(), s->outbuf.begin() + s->state.out_bytes);
  if (op == BROTLI_AMD_OP_FINISH) s->finished = true;
  *out = s->host_out.data();
  *out_len = s->host_out.size();
  return BROTLI_AMD_OK;
}

int brotli_amd_stream_attach_dictionary(BrotliAmdStream* s, const BrotliAmdDictChunk* chunks, uint32_t nchunks) {
  const int rc = s->dict.set(chunks, nchunks);
  return rc == BROTLI_AMD_OK ? rc : set_err(s->c, "more than 15 dictionary chunks", rc);
}

int brotli_amd_ctx_set_dictionary(BrotliAmdCtx* c, const BrotliAmdDictChunk* chunks, uint32_t nchunks) {
  const int rc = c->dict.set(chunks, nchunks);
  return rc == BROTLI_AMD_OK ? rc : set_err(c, "more than 15 dictionary chunks", rc);
}

int brotli_amd_stream_take_partial(BrotliAmdStream* s, uint32_t* nbits, uint32_t* value) {
  *value = s->state.last_bytes;
  *nbits = s->state.last_bytes_bits;
  s->state.last_bytes = 0;
  s->state.last_bytes_bits = 0;
  return BROTLI_AMD_OK;
}

void brotli_amd_stream_destroy(BrotliAmdStream* s) { delete s; }

}  // extern "C"
