// tests/simt/sim_driver.cc — TEST ONLY: runs the kernels of brotli_amd/csrc on
// the host SIMT simulator so their logic can be checked against the oracle
// in CI without a GPU.  Exposes a small C API for tests/test_sim_*.py.
#define BROTLI_AMD_SIMT_SIM 1
#include "simt.h"

#include <stdio.h>
#include <stdlib.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

#include "../../brotli_amd/csrc/kernels.h"
#include "../../brotli_amd/csrc/host_plan.h"

unsigned long long g_sim_counts[16];
namespace {
struct Launch { void (*fn)(JobArgs); JobArgs a; };
void tramp(void* p) { Launch* l = (Launch*)p; l->fn(l->a); }
void run(void (*fn)(JobArgs), const JobArgs& a, unsigned grid, unsigned block, int reverse) {
  Launch l{fn, a};
  simt::launch(grid, block, tramp, &l, reverse);
}
// What run_build_store (kernels.h) launches through; SIM_WIDE=K: the many-wave kernels of k_wide.h for every meta-block, K waves each.
struct SimRun {
  int reverse;
  void operator()(void (*fn)(JobArgs), const JobArgs& a, unsigned grid, unsigned block) { run(fn, a, grid, block, reverse); }
};
uint32_t sim_wide() { const char* e = getenv("SIM_WIDE"); return e ? (uint32_t)atoi(e) : 0u; }     // 0 = off, else waves per meta-block
// The index kernels of an indexed job (k_index.h), once per job.
void run_index(const JobArgs& a, int reverse) {
  run(k_ix_count, a, a.nshards * a.J.ix_slices, 64, reverse);
  run(k_ix_scan, a, a.nshards, 64, reverse);
  run(k_ix_scatter, a, a.nshards * a.J.ix_slices, 64, reverse);
  run((a.J.flags & JOB_FLAG_STREAMT) ? k_ix_bucket_s : k_ix_bucket, a, ix_bucket_grid(a.J, a.nshards), 64, reverse);
  run((a.J.flags & JOB_FLAG_STREAMT) ? k_ix_big_s : k_ix_big, a, 8u * 3u, 64, reverse);      // (three waves per list)
}
void run_parse_kernel(JobArgs a, int reverse, int round = 0) {
  if (a.J.flags & JOB_FLAG_QUICK) {
    run(k_parse_quick, a, a.nshards, 64, reverse);
  } else if (a.J.flags & JOB_FLAG_DEEP) {
    if (a.J.block_bits <= 6) run(k_parse_deep<1>, a, a.nshards, 64, reverse);
    else if (a.J.block_bits == 7) run(k_parse_deep<2>, a, a.nshards, 64, reverse);
    else run(k_parse_deep<4>, a, a.nshards, 64, reverse);
  } else if ((a.J.flags & JOB_FLAG_TILED) && round != 0) {
    // later rounds of a shard that left the tiled path (several meta-blocks): the plain chain
    JobArgs c = a;
    c.J.flags &= ~(uint32_t)(JOB_FLAG_TILED | JOB_FLAG_SWEEP);
    run(k_chain, c, (a.nshards + q_groups_per_wave(a.J) - 1) / q_groups_per_wave(a.J), 64, reverse);
    run(k_cmd_encode, c, a.nshards * CE_SPLIT, 64, reverse);
  } else if (a.J.flags & JOB_FLAG_TILED) {
    // the tiles' parses, then verify / events / sweep until nothing is pending (what run_rounds of hip_layer.hip does)
    const uint32_t gpw = q_groups_per_wave(a.J);
    run(k_chain_tiles, a, (a.ntiles + gpw - 1) / gpw, 64, reverse);
    // shards whose tile 0 ended with the static dictionary's gate open: their other tiles once more, gate taken as open
    run(k_tile_restart, a, a.nshards, 64, reverse);
    run(k_tile_restart_clear, a, a.ntiles, 64, reverse);
    run(k_chain_tiles, a, (a.ntiles + gpw - 1) / gpw, 64, reverse);
    if (getenv("SIM_TILE_LOG")) fprintf(stderr, "shards restarted with the gate taken as open: %u\n", a.counters[TILE_CNT_RESTART]);
    bool settled = false;
    int rounds = 0;
    for (; rounds < 12 && !settled; ++rounds) {
      a.counters[TILE_CNT_START] = a.counters[TILE_CNT_FLIPS] = a.counters[TILE_CNT_RESTART] = 0;
      if (rounds != 0) {       // tiles the gate walk sent back (k_tile.h): parsed again from scratch, every search exact
        JobArgs l = a;
        l.J.flags |= JOB_FLAG_VIEWALL;
        run(k_tile_restart_clear, l, a.ntiles, 64, reverse);
        run(k_chain_tiles, l, (a.ntiles + gpw - 1) / gpw, 64, reverse);
      }
      {
        JobArgs e = a;
        if (rounds != 0 || getenv("SIM_EVENTS_ALL")) e.J.flags |= JOB_FLAG_SWEEP;     // (first pass: cross-tile successors only, k_tile.h)
        run(k_tile_events, e, a.nshards * a.J.ix_slices, 64, reverse);
      }
      run(k_tile_verify, a, a.nshards, 64, reverse);
      if (getenv("SIM_TILE_LOG")) fprintf(stderr, "tile round %d: start events %u, changed skip bits %u, shards off the tiled path %u\n", rounds,
                                          a.counters[TILE_CNT_START], a.counters[TILE_CNT_FLIPS], a.counters[TILE_CNT_BAD]);
      if (a.counters[TILE_CNT_START] == 0 && a.counters[TILE_CNT_FLIPS] == 0 && a.counters[TILE_CNT_RESTART] == 0) { settled = true; break; }
      JobArgs b = a;
      b.J.flags |= JOB_FLAG_SWEEP;
      const unsigned long long c7 = g_sim_counts[7], c15 = g_sim_counts[15], c5 = g_sim_counts[5];
      {
        uint32_t sg = getenv("SIM_SWEEP_GROUPS") ? (uint32_t)atoi(getenv("SIM_SWEEP_GROUPS")) : 2u;      // (the library's default)
        if (sg != 1 && sg != 2 && sg != 4) sg = 2;
        b.J.flags &= ~(3u << JOB_FLAG_GROUPS_SHIFT);
        if (sg != 4) b.J.flags |= sg << JOB_FLAG_GROUPS_SHIFT;
        run(k_chain_sweep, b, (a.ntiles + sg - 1) / sg, 64, reverse);
      }
      if (getenv("SIM_TILE_LOG")) fprintf(stderr, "  sweep: %llu main-loop rounds, %llu replay rounds, %llu exact resolves (%u tiles)\n",
                                          g_sim_counts[7] - c7, g_sim_counts[15] - c15, g_sim_counts[5] - c5, a.ntiles);
    }
    if (!settled) {           // give up on the tiles: every shard the plain way
      for (uint32_t k = 0; k < a.nshards; ++k) if (a.shards[k].ntiles > 1) a.trecs[a.shards[k].tile_base].flags |= TILE_BAD;
      a.counters[TILE_CNT_BAD] = 1;
    }
    run(k_tile_finish, a, a.ntiles, 64, reverse);
    if (a.counters[TILE_CNT_BAD] != 0) {
      run(k_tile_fallback, a, a.nshards, 64, reverse);
      JobArgs c = a;
      c.J.flags &= ~(uint32_t)(JOB_FLAG_TILED | JOB_FLAG_SWEEP);
      run(k_chain, c, (a.nshards + gpw - 1) / gpw, 64, reverse);
      run(k_cmd_encode, c, a.nshards * CE_SPLIT, 64, reverse);
    }
  } else if (a.J.flags & JOB_FLAG_INDEXED) {
    run(k_chain, a, (a.nshards + q_groups_per_wave(a.J) - 1) / q_groups_per_wave(a.J), 64, reverse);
    run(k_cmd_encode, a, a.nshards * CE_SPLIT, 64, reverse);
  } else if (a.J.flags & JOB_FLAG_QUAD) {
    run(k_parse4, a, (a.nshards + q_groups_per_wave(a.J) - 1) / q_groups_per_wave(a.J), 64, reverse);
  } else {
    run(k_parse, a, a.nshards, 64, reverse);
  }
}
}  // namespace

extern "C" {

// Parses every shard of a plan (first meta-block of each shard only) and
// returns the concatenated command lists.  Returns number of commands or -1.
long sim_parse(const char* tables_path, const uint8_t* in, size_t len, int quality, int lgwin,
               uint32_t size_hint, size_t shard_size, int reverse, int no_pair,
               Command* cmds_out, size_t cap, uint64_t* stats /*[3]*/) {
  HostTables ht;
  if (!host_tables_load(tables_path, &ht)) return -1;
  JobPlan plan;
  if (!plan_job(len, quality, lgwin, size_hint, shard_size, 0, true, &plan)) return -2;
  if (no_pair & 1) plan.J.flags |= JOB_FLAG_NO_PAIR;
  if (no_pair & 2) plan.J.flags |= JOB_FLAG_QUAD;
  if (no_pair & 4) plan.J.flags |= JOB_FLAG_FORCE_SLOW;
  plan.J.flags |= (uint32_t)(no_pair & ~7);   // other job flags pass through (DUO, GROUPS)
  if (plan.J.quality != 5) plan.J.flags = (plan.J.flags & ~(JOB_FLAG_QUAD | JOB_FLAG_INDEXED)) | JOB_FLAG_DEEP;
  if (plan.J.flags & JOB_FLAG_INDEXED) plan_add_index(&plan, true);
  if ((plan.J.flags & JOB_FLAG_INDEXED) && getenv("SIM_TILE_KB"))
    plan_add_tiles(&plan, (uint32_t)atoi(getenv("SIM_TILE_KB")), getenv("SIM_TILE_WARM") ? (uint32_t)atoi(getenv("SIM_TILE_WARM")) : 2048u);
  std::vector<TileRec> trecs(plan.tiles.size() + 1);
  memset(trecs.data(), 0, trecs.size() * sizeof(TileRec));
  std::vector<uint8_t> input(len + 64, 0);
  memcpy(input.data(), in, len);
  std::vector<uint8_t> ws(plan.ws_bytes, 0xCD);
  std::vector<ShardState> states(plan.shards.size());
  std::vector<double> log2lut;
  DeviceTables T;
  host_tables_fill(ht, plan.J.log2_lut_size, &log2lut, &T);
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
  a.tiles = plan.tiles.data();
  a.trecs = trecs.data();
  a.ntiles = (uint32_t)plan.tiles.size();
  run(k_init, a, a.nshards * a.init_blocks_per_shard, 256, 0);
  if (plan.J.flags & JOB_FLAG_INDEXED) run_index(a, reverse);
  run_parse_kernel(a, reverse);
  if (getenv("SIM_COUNTS")) {
    fprintf(stderr, "sim counts: steps=%llu with_bucket_cand=%llu ext=%llu store_steps=%llu dict=%llu slow=%llu dup=%llu\n",
            g_sim_counts[0], g_sim_counts[1], g_sim_counts[2], g_sim_counts[3], g_sim_counts[4], g_sim_counts[5], g_sim_counts[6]);
    memset(g_sim_counts, 0, sizeof(g_sim_counts));
  }
  size_t n = 0;
  stats[0] = stats[1] = stats[2] = 0;
  for (size_t k = 0; k < plan.shards.size(); ++k) {
    const ShardState& S = states[k];
    if (S.error) return -3;
    const Command* c = (const Command*)(ws.data() + plan.shards[k].cmds_off);
    for (uint32_t i = 0; i < S.ncmds; ++i) {
      if (n < cap) cmds_out[n] = c[i];
      ++n;
    }
    stats[0] += S.stat_searches; stats[1] += S.stat_pairs; stats[2] += S.stat_b_used;
  }
  return (long)n;
}


// Full encode of a plan on the simulator: init, then parse/build/store rounds
// until every shard is done; outputs concatenated in shard order.
// Returns the number of bytes or a negative error.
long sim_encode(const char* tables_path, const uint8_t* in, size_t len, int quality, int lgwin,
                uint32_t size_hint, size_t shard_size, uint64_t stream_base, int is_last,
                int reverse, int flags, uint8_t* out, size_t out_cap) {
  HostTables ht;
  if (!host_tables_load(tables_path, &ht)) return -1;
  JobPlan plan;
  if (!plan_job(len, quality, lgwin, size_hint, shard_size, stream_base, is_last != 0, &plan, true, (flags >> 24) & 31)) return -2;
  plan.J.flags |= (uint32_t)flags & 0xFFFFFFu;       // (bits 24..28: BROTLI_PARAM_LGBLOCK)
  if (plan.J.quality != 5) plan.J.flags = (plan.J.flags & ~(JOB_FLAG_QUAD | JOB_FLAG_INDEXED)) | JOB_FLAG_DEEP;
  if (plan.J.flags & JOB_FLAG_INDEXED) plan_add_index(&plan, true);
  if ((plan.J.flags & JOB_FLAG_INDEXED) && getenv("SIM_TILE_KB"))
    plan_add_tiles(&plan, (uint32_t)atoi(getenv("SIM_TILE_KB")), getenv("SIM_TILE_WARM") ? (uint32_t)atoi(getenv("SIM_TILE_WARM")) : 2048u);
  std::vector<TileRec> trecs(plan.tiles.size() + 1);
  memset(trecs.data(), 0, trecs.size() * sizeof(TileRec));
  std::vector<uint8_t> input(len + 64, 0);
  memcpy(input.data(), in, len);
  std::vector<uint8_t> ws(plan.ws_bytes, 0xCD);
  std::vector<ShardState> states(plan.shards.size());
  std::vector<double> log2lut;
  DeviceTables T;
  host_tables_fill(ht, plan.J.log2_lut_size, &log2lut, &T);
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
  a.tiles = plan.tiles.data();
  a.trecs = trecs.data();
  a.ntiles = (uint32_t)plan.tiles.size();
  run(k_init, a, a.nshards * a.init_blocks_per_shard, 256, 0);
  if (plan.J.flags & JOB_FLAG_INDEXED) run_index(a, reverse);
  for (int round = 0; round < 100000; ++round) {
    memset(counters, 0, sizeof(counters));
    run_parse_kernel(a, reverse, round);
    {
      SimRun R{reverse};
      run_build_store(R, a, a.nshards, sim_wide(), [&]() {
    if (getenv("SIM_DEBUG")) {
        for (size_t k = 0; k < plan.shards.size(); ++k) {
          if (!states[k].mb_valid) continue;
          MbLayout L;
          mb_layout(plan.shards[k].len < plan.J.max_metablock_size ? plan.shards[k].len : plan.J.max_metablock_size, &L);
          const uint8_t* mb = ws.data() + plan.shards[k].mb_off;
          const MbInfo* I = (const MbInfo*)(mb + L.info);
          fprintf(stderr, "shard %zu raw=%u nc=%u kind=%u nlits=%u ndist=%u ncmds=%u\n", k, states[k].mb_raw,
                  I->num_contexts, I->map_kind, I->nlits, I->ndist, I->ncmds);
          for (int c = 0; c < 3; ++c) {
            fprintf(stderr, "  cat %d types=%u blocks=%u:", c, I->split[c].num_types, I->split[c].num_blocks);
            const uint8_t* ty = mb + L.types[c];
            const uint32_t* le = (const uint32_t*)(mb + L.lengths[c]);
            for (uint32_t b = 0; b < I->split[c].num_blocks && b < 40; ++b) fprintf(stderr, " %u:%u", ty[b], le[b]);
            fprintf(stderr, "\n");
          }
        }
      }
      });
    }
    if (counters[1]) return -3;
    if (counters[0] == 0) break;
  }
  if (getenv("SIM_COUNTS") && (plan.J.flags & JOB_FLAG_INDEXED)) {
    uint64_t se = 0, sl = 0;
    for (const ShardState& S : states) { se += S.stat_searches; sl += S.ix_slow; }
    fprintf(stderr, "indexed parse: %llu searches, %llu exact (in-chain) searches; wave steps %llu; evaluated positions %llu: "
            "index-undecidable %llu, bloom %llu, gate-dependent %llu, long %llu; fast-loop steps %llu\n", (unsigned long long)se, (unsigned long long)sl,
            g_sim_counts[7], g_sim_counts[12], g_sim_counts[8], g_sim_counts[9], g_sim_counts[10], g_sim_counts[11], g_sim_counts[14]);
    fprintf(stderr, "group fast steps: commit %llu, literals only %llu | stuck: dictionary after exact %llu, lazy chain into undecidable %llu, empty probe with open gate %llu, spree or gate at first %llu, other %llu\n",
            g_sim_counts[0], g_sim_counts[1], g_sim_counts[2], g_sim_counts[3], g_sim_counts[4], g_sim_counts[5], g_sim_counts[6]);
    memset(g_sim_counts, 0, sizeof(g_sim_counts));
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

// One unpartitioned quality-5 stream on the tiled path (JOB_FLAG_STREAMT; what run_stream_job of hip_layer.hip
// does).  Returns the number of bytes, -10 when the stream leaves the tiled path (info[0] = the reasons, TILE_WHY_*:
// the library then takes the serial path), other negatives on errors.  info: [0] reasons, [1] sweeps, [2] meta-blocks.
long sim_encode_stream(const char* tables_path, const uint8_t* in, size_t len, int lgwin, uint32_t size_hint,
                       int reverse, int flags, uint8_t* out, size_t out_cap, uint32_t* info) {
  HostTables ht;
  if (!host_tables_load(tables_path, &ht)) return -1;
  JobPlan plan;
  const uint32_t warm = getenv("SIM_TILE_WARM") ? (uint32_t)atoi(getenv("SIM_TILE_WARM")) : 2048u;
  if (!plan_stream(len, lgwin, size_hint, warm, true, &plan)) return -2;
  plan.J.flags |= (uint32_t)flags;
  std::vector<TileRec> trecs(plan.tiles.size() + 1);
  memset(trecs.data(), 0, trecs.size() * sizeof(TileRec));
  std::vector<uint8_t> input(len + 64, 0);
  memcpy(input.data(), in, len);
  std::vector<uint8_t> ws(plan.ws_bytes, 0xCD);
  std::vector<ShardState> states(1);
  std::vector<ShardDesc> mdesc(plan.mcap);
  std::vector<ShardState> mstate(plan.mcap);
  memset(mstate.data(), 0, mstate.size() * sizeof(ShardState));
  std::vector<uint64_t> moff(plan.mcap + 3, 0);
  std::vector<uint8_t> sout(plan.max_out_bytes + 64, 0);
  std::vector<double> log2lut;
  DeviceTables T;
  host_tables_fill(ht, plan.J.log2_lut_size, &log2lut, &T);
  uint32_t counters[16] = {0};
  JobArgs a;
  a.J = plan.J;
  a.shards = plan.shards.data();
  a.states = states.data();
  a.T = &T;
  a.input = input.data();
  a.ws = ws.data();
  a.nshards = 1;
  a.init_blocks_per_shard = 1;
  a.counters = counters;
  a.tiles = plan.tiles.data();
  a.trecs = trecs.data();
  a.ntiles = (uint32_t)plan.tiles.size();
  a.chunks = plan.chunks.data();
  a.mdesc = mdesc.data();
  a.mstate = mstate.data();
  a.moff = moff.data();
  a.sout = sout.data();
  a.mcap = plan.mcap;
  info[0] = info[1] = info[2] = 0;
  const bool tlog = getenv("SIM_TILE_LOG") != nullptr;
  auto lap = [&](const char* what) {
    static double t_prev = 0;
    timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    const double now = ts.tv_sec + ts.tv_nsec * 1e-9;
    if (tlog && what) fprintf(stderr, "  [sim] %-16s %7.2f s\n", what, now - t_prev);
    t_prev = now;
  };
  lap(nullptr);
  run(k_init, a, 1, 256, 0);
  {
    JobArgs c = a;                                  // the index kernels see the chunks as their shards
    c.shards = plan.chunks.data();
    c.nshards = plan.J.nchunks;
    run_index(c, reverse);
  }
  lap("index");
  const uint32_t gpw = q_groups_per_wave(a.J);
  const uint32_t nkg = (1u << a.J.bucket_bits) / 64u;
  // the searches behind a wrap of the 16-bit store counter, as far as they can be told before the parse
  const bool zones = !getenv("SIM_NO_ZONES");      // (test knob: without them a stream with a counter wrap comes out wrong)
  if (zones) run(k_stream_kprefix, a, nkg, 64, reverse);
  a.aux = 1;                       // (every key run; the pass loop's launches walk the ones that changed: SKT_DIRTY)
  if (zones) run(k_stream_zones, a, a.J.nchunks * nkg, 64, reverse);
  a.aux = 0;
  lap("zones0");
  run(k_chain_tiles, a, (a.ntiles + gpw - 1) / gpw, 64, reverse);
  lap("first parse");
  run(k_tile_restart, a, 1, 64, reverse);
  run(k_tile_restart_clear, a, a.ntiles, 64, reverse);
  run(k_chain_tiles, a, (a.ntiles + gpw - 1) / gpw, 64, reverse);
  if (tlog) fprintf(stderr, "restarted with the gate taken as open: %u\n", counters[TILE_CNT_RESTART]);
  lap("second parse");
  uint32_t nmb = 0;
  int rounds = 0, outer = 0;
  for (;; ++outer) {           // (again when a raw meta-block rolls the distance cache back for the tile behind it)
  if (outer >= 8) { info[0] = TILE_WHY_RAW; return -10; }
  bool settled = false;
  for (int r = 0; r < 16 && !settled; ++r, ++rounds) {
    counters[TILE_CNT_START] = counters[TILE_CNT_FLIPS] = counters[TILE_CNT_RESTART] = 0;
    if (rounds != 0) {         // tiles the gate walk sent back (k_tile.h): parsed again from scratch, every search exact
      JobArgs l = a;
      l.J.flags |= JOB_FLAG_VIEWALL;
      run(k_tile_restart_clear, l, a.ntiles, 64, reverse);
      run(k_chain_tiles, l, (a.ntiles + gpw - 1) / gpw, 64, reverse);
    }
    {
      JobArgs e = a;
      if (rounds != 0 || getenv("SIM_EVENTS_ALL")) e.J.flags |= JOB_FLAG_SWEEP;
      run(k_stream_flips, e, a.J.nchunks * a.J.ix_slices, 64, reverse);
      run(k_stream_flipcheck, e, 1, 64, reverse);
      run(k_stream_events, e, a.J.nchunks * a.J.ix_slices, 64, reverse);
    }
    lap("events");
    run(k_stream_skclear, a, a.J.nchunks * a.J.ix_slices, 64, reverse);
    run(k_stream_skcount, a, a.J.nchunks * a.J.ix_slices, 64, reverse);
    if (zones) run(k_stream_kprefix, a, nkg, 64, reverse);
    a.aux = 0;
    if (zones) run(k_stream_zones, a, a.J.nchunks * nkg, 64, reverse);
    lap("zones");
    run(k_stream_cuts, a, 1, 64, reverse);
    run(k_stream_verify, a, (a.ntiles + 63) / 64, 64, reverse);
    lap("cuts+verify");
    if (getenv("SIM_TILE_LOG")) fprintf(stderr, "stream round %d: start events %u, changed skip bits %u, bad %u (flags %x)\n", rounds,
                                        counters[TILE_CNT_START], counters[TILE_CNT_FLIPS], counters[TILE_CNT_BAD], trecs[0].flags);
    if (counters[TILE_CNT_BAD] != 0) { info[0] = trecs[0].flags; info[1] = (uint32_t)rounds; return -10; }
    if (counters[TILE_CNT_START] == 0 && counters[TILE_CNT_FLIPS] == 0 && counters[TILE_CNT_RESTART] == 0) { settled = true; break; }
    JobArgs b = a;
    b.J.flags |= JOB_FLAG_SWEEP;
    uint32_t sg = getenv("SIM_SWEEP_GROUPS") ? (uint32_t)atoi(getenv("SIM_SWEEP_GROUPS")) : 2u;
    if (sg != 1 && sg != 2 && sg != 4) sg = 2;
    b.J.flags &= ~(3u << JOB_FLAG_GROUPS_SHIFT);
    if (sg != 4) b.J.flags |= sg << JOB_FLAG_GROUPS_SHIFT;
    run(k_chain_sweep, b, (a.ntiles + sg - 1) / sg, 64, reverse);
    lap("sweep");
  }
  info[1] = (uint32_t)rounds;
  if (!settled) { info[0] = TILE_WHY_EVENTS; return -10; }
  if (tlog && (trecs[0].flags & TILE_GATE_OPEN))
    for (uint32_t t = 0; t < a.ntiles; ++t)
      fprintf(stderr, "  gate: tile %u hyp %u in (%u, %u) moved (%u, %u) closed at end %u flags %x\n", t, trecs[t].hyp, trecs[t].in_l, trecs[t].in_m,
              trecs[t].dlookups, trecs[t].dmatches, trecs[t].out_gate, trecs[t].flags);
  a.aux = 1;
  run(k_stream_cuts, a, 1, 64, reverse);
  if (counters[TILE_CNT_BAD] != 0) { info[0] = trecs[0].flags; return -10; }
  run(k_stream_finish, a, a.ntiles, 64, reverse);
  nmb = counters[TILE_CNT_NMB];
  info[2] = nmb;
  if (const char* path = getenv("SIM_STREAM_CMDS")) {        // (debugging: the stream's commands, all meta-blocks)
    uint64_t total = 0;
    for (uint32_t m = 0; m < nmb; ++m) total += mstate[m].ncmds;
    if (FILE* f = fopen(path, "wb")) { fwrite(ws.data() + plan.shards[0].cmds_off, sizeof(Command), total, f); fclose(f); }
  }
  {
    JobArgs m = a;                                  // build / store see the meta-blocks as their shards
    m.shards = mdesc.data();
    m.states = mstate.data();
    m.nshards = nmb;
    lap("finish");
    SimRun R{reverse};
    run_build_store(R, m, nmb, sim_wide(), [&]() { lap("build"); });
    lap("store");
  }
  run(k_stream_scan, a, 1, 64, reverse);
  if (counters[1]) return -3;
  if (counters[TILE_CNT_RAW] != 0) return -3;
  counters[TILE_CNT_RBCHG] = 0;
  run(k_stream_rollback, a, 1, 64, reverse);
  if (tlog) {
    uint32_t nraw = 0;
    for (uint32_t m = 0; m < nmb; ++m) nraw += mstate[m].mb_was_raw != 0;
    fprintf(stderr, "outer %d: %u meta-blocks (%u raw), tiles with a new roll-back %u\n", outer, nmb, nraw, counters[TILE_CNT_RBCHG]);
  }
  if (counters[TILE_CNT_RBCHG] == 0) break;
  memset(mstate.data(), 0, mstate.size() * sizeof(ShardState));
  }
  run(k_stream_place, a, nmb * STREAM_PLACE_PARTS, 256, reverse);
  uint64_t nbytes = (moff[nmb] + 7) / 8;
  if (nbytes + 1 > out_cap) return -4;
  memcpy(out, sout.data(), nbytes);
  if ((flags & (int)JOB_FLAG_TAILFIN) != 0 && moff[nmb + 1] != 0) {      // (hip_layer.hip: brotli_amd_encode_host)
    out[nbytes] = 0;
    nbytes = stream_tail_fix(out, moff[nmb + 2], moff[nmb]);
  }
  return (long)nbytes;
}

// The workspace arithmetic of a tiled stream (plan_stream + stream_emit_mb) at sizes the simulator cannot run: a
// stream of `len` bytes cut into meta-blocks at random tile boundaries (at least `min_tiles` tiles each, as many
// meta-blocks as the plan allows at most) with the most commands a meta-block can have; every meta-block's literal /
// symbol / scratch / work / output regions must lie inside the stream's regions and must not overlap its
// neighbour's.  Returns 0, or the number of the first check that failed.
long sim_stream_layout_check(uint64_t len, int lgwin, uint32_t seed) {
  JobPlan plan;
  if (!plan_stream(len, lgwin, 0, 2048, /*ix_in_ws=*/false, &plan)) return -2;
  const JobParams& J = plan.J;
  const ShardDesc& D = plan.shards[0];
  const uint32_t nt = D.ntiles, mcap = plan.mcap;
  std::vector<ShardDesc> md(mcap);
  std::vector<ShardState> ms(mcap);
  uint8_t* input = (uint8_t*)calloc(len + 64, 1);          // (stream_emit_mb reads the two bytes in front of a meta-block)
  if (!input) return -3;
  struct Free { uint8_t* p; ~Free() { free(p); } } guard{input};
  // a meta-block holds at least max_literals bytes unless it is the last one (a cut needs that many literals, or
  // twice as many commands' bytes, or the size limit): cut as early as that allows, sometimes later
  const uint32_t min_tiles = (J.max_literals + (1u << J.tile_log2) - 1u) >> J.tile_log2;
  uint32_t s = 0, m = 0, cmd_lo = 0, rng = seed * 2654435761u + 1u;
  uint64_t end_prev[5] = {0, 0, 0, 0, 0};
  while (s < nt) {
    rng = rng * 1664525u + 1013904223u;
    uint32_t e = s + (min_tiles ? min_tiles : 1u) - 1u + ((rng >> 20) % 3u == 0 ? (rng >> 8) % 40u : 0u);
    const uint32_t max_tiles = J.max_metablock_size >> J.tile_log2;
    if (e - s + 1u > max_tiles) e = s + max_tiles - 1u;
    if (e >= nt - 1u) e = nt - 1u;
    if (m >= mcap) return 1;
    const uint64_t start = (uint64_t)s << J.tile_log2, end = e + 1u == nt ? len : (uint64_t)(e + 1u) << J.tile_log2, bytes = end - start;
    const uint32_t ncmds = (uint32_t)(bytes / 2 + (e - s + 1u) + 1u);
    stream_emit_mb(J, D, input, md.data(), ms.data(), m, s, e, cmd_lo, ncmds, (uint32_t)bytes, e + 1u == nt);
    const ShardDesc& X = md[m];
    if (X.len != bytes || ms[m].mb_start != start) return 2;
    const uint64_t lo[5] = {X.lits_off, X.dsym_off, X.mb_off, X.scratch_off, X.out_off};
    const uint64_t need[5] = {(bytes + 8) * 2, (uint64_t)ncmds * 2, mb_work_bytes(bytes), (bytes / 256 + 64) * 8 + (2 * bytes + 64) * 4, X.out_cap};
    const uint64_t rlo[5] = {D.lits_off, D.dsym_off, D.mb_off, D.scratch_off, D.out_off};
    const uint64_t rhi[5] = {D.dsym_off, D.mb_off, D.scratch_off, D.out_off, D.cmds2_off};
    if (X.out_cap < 2 * bytes + 536) return 3;
    for (int k = 0; k < 5; ++k) {
      if (lo[k] < rlo[k] || lo[k] + need[k] > rhi[k]) return 10 + k;
      if (lo[k] < end_prev[k]) return 20 + k;
      end_prev[k] = lo[k] + need[k];
    }
    if (X.cmds_off < D.cmds_off || X.cmds_off + (uint64_t)ncmds * 16 > D.lits_off) return 4;
    cmd_lo += ncmds;
    s = e + 1u;
    ++m;
  }
  return 0;
}

// Quality 1 on the simulator: the k_fast_* pipeline for one run of calls ending in
// FINISH (is_last) or at a byte-pending point.  Returns the number of output bits
// (bytes = (bits + 7) / 8 are written), negative on error.
namespace {
struct FLaunch { void (*fn)(FastArgs); FastArgs a; };
void ftramp(void* p) { FLaunch* l = (FLaunch*)p; l->fn(l->a); }
void frun(void (*fn)(FastArgs), const FastArgs& a, unsigned grid, unsigned block, int reverse) {
  FLaunch l{fn, a};
  simt::launch(grid, block, ftramp, &l, reverse);
}
}  // namespace

long sim_encode_fast(const char* tables_path, const uint8_t* in, size_t len, int lgwin,
                     const uint64_t* call_sizes, size_t ncalls, uint32_t carry_bits,
                     uint32_t carry_value, int is_last, int reverse, uint8_t* out, size_t out_cap) {
  HostTables ht;
  if (!host_tables_load(tables_path, &ht)) return -1;
  FastPlan plan;
  if (!plan_fast(len, lgwin, call_sizes, ncalls, &plan)) return -2;
  std::vector<uint8_t> input(len + 64, 0);
  memcpy(input.data(), in, len);
  std::vector<uint8_t> ws(plan.ws_bytes, 0xCD);
  std::vector<FastBlockState> bstate(plan.blocks.size() + 1);
  std::vector<FastFragState> fstate(plan.frags.size() + 1);
  memset(fstate.data(), 0, fstate.size() * sizeof(FastFragState));
  memset(bstate.data(), 0, bstate.size() * sizeof(FastBlockState));
  std::vector<double> log2lut;
  DeviceTables T;
  host_tables_fill(ht, 4096, &log2lut, &T);
  std::vector<uint8_t> obuf(plan.max_out_bytes + 64, 0);
  uint64_t result[2] = {0, 0};
  FastArgs a;
  a.frags = plan.frags.data();
  a.blocks = plan.blocks.data();
  a.bstate = bstate.data();
  a.fstate = fstate.data();
  a.T = &T;
  a.input = input.data();
  a.ws = ws.data();
  a.out = obuf.data();
  a.result = result;
  a.cmds_base = plan.cmds_base; a.lits_base = plan.lits_base; a.lsum_base = plan.lsum_base;
  a.scr_base = plan.scr_base; a.tables_base = plan.tables_base;
  a.out_cap = plan.max_out_bytes;
  a.nfrags = (uint32_t)plan.frags.size();
  a.nblocks = (uint32_t)plan.blocks.size();
  a.nslots = plan.nslots;
  if (getenv("SIM_FAST_SLOTS")) a.nslots = (uint32_t)atoi(getenv("SIM_FAST_SLOTS"));
  a.carry_bits = carry_bits; a.carry_value = carry_value; a.is_last = (uint32_t)is_last;
  frun(k_fast_parse, a, a.nslots, 64, reverse);
  if (a.nblocks) frun(k_fast_store, a, a.nblocks, 64, reverse);
  frun(k_fast_sizes, a, (8 * a.nfrags + 255) / 256, 256, reverse);
  frun(k_fast_scan, a, 1, 64, reverse);
  if (a.nblocks) frun(k_fast_emit, a, a.nblocks, 256, reverse);
  if (result[1]) return -3;
  const size_t nbytes = (size_t)((result[0] + 7) / 8);
  if (nbytes > out_cap) return -4;
  memcpy(out, obuf.data(), nbytes);
  return (long)result[0];
}

// One encoder instance fed call by call (the HIP layer's BrotliAmdStream, hip_layer.hip
// stream_init / stream_run, restated for the simulator): qualities 5 (k_parse) and 6-9
// (k_parse_deep).  Call k hands call_sizes[k] bytes of `in` with operation call_ops[k]
// (0 PROCESS, 1 FLUSH, 2 FINISH, 3 EMIT_METADATA: those bytes are the metadata payload and
// the header continues the stream's open byte, as encode_abi.c does).  Returns the number
// of output bytes, negative on error (-5: the rounds did not converge).
long sim_stream(const char* tables_path, const uint8_t* in, size_t len, int quality, int lgwin,
                uint32_t size_hint, uint32_t stream_offset, const uint64_t* call_sizes,
                const uint8_t* call_ops, size_t ncalls, int reverse, uint8_t* out, size_t out_cap) {
  HostTables ht;
  if (!host_tables_load(tables_path, &ht)) return -1;
  JobParams J;
  if (!plan_params(quality, lgwin, size_hint, &J, (reverse >> 24) & 31)) return -2;       // (bits 24..28 of `reverse`: BROTLI_PARAM_LGBLOCK)
  reverse &= 1;
  if (quality != 5) J.flags |= JOB_FLAG_DEEP;
  const uint64_t mb = J.max_metablock_size;
  J.log2_lut_size = (uint32_t)(mb + 2);
  ShardDesc D;
  memset(&D, 0, sizeof(D));
  uint64_t so = stream_offset;
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
  D.out_off = off;
  D.out_cap = 2 * (len + mb + (2ull << J.lgblock)) + 8192;
  off = plan_align(off + D.out_cap);
  std::vector<uint8_t> ws(off, 0xCD);
  std::vector<uint8_t> input(len + 64, 0);
  ShardState state;
  std::vector<double> log2lut;
  DeviceTables T;
  host_tables_fill(ht, J.log2_lut_size, &log2lut, &T);
  uint32_t counters[16] = {0};
  JobArgs a;
  a.J = J;
  a.shards = &D;
  a.states = &state;
  a.T = &T;
  a.input = input.data();
  a.ws = ws.data();
  a.nshards = 1;
  a.init_blocks_per_shard = 64;
  a.counters = counters;
  run(k_init, a, 64, 256, 0);
  size_t n_out = 0, pos = 0;
  uint64_t fed = 0;
  for (size_t k = 0; k < ncalls; ++k) {
    const uint64_t n = call_sizes[k];
    const int op = call_ops[k];
    if (pos + n > len) return -2;
    if (op != 3) { memcpy(input.data() + fed, in + pos, n); fed += n; }
    D.len = (uint32_t)fed;
    D.final_op = (uint32_t)op;
    state.done = 0;
    state.out_bytes = 0;
    int round = 0;
    for (;; ++round) {
      if (round > 4096) return -5;
      memset(counters, 0, sizeof(counters));
      if (J.flags & JOB_FLAG_QUICK) run(k_parse_quick, a, 1, 64, reverse);
      else if (!(J.flags & JOB_FLAG_DEEP)) run(k_parse, a, 1, 64, reverse);
      else if (J.block_bits <= 6) run(k_parse_deep<1>, a, 1, 64, reverse);
      else if (J.block_bits == 7) run(k_parse_deep<2>, a, 1, 64, reverse);
      else run(k_parse_deep<4>, a, 1, 64, reverse);
      {
        SimRun R{reverse};
        run_build_store(R, a, 1, sim_wide(), []() {});
      }
      if (counters[1]) return -3;
      if (counters[0] == 0) break;
    }
    if (n_out + state.out_bytes > out_cap) return -4;
    memcpy(out + n_out, ws.data() + D.out_off, state.out_bytes);
    n_out += state.out_bytes;
    if (op == 3) {
      // WriteMetadataHeader (encode.c:1223-1249) over the open byte, then the payload
      uint64_t bits = state.last_bytes;
      uint32_t nbits = state.last_bytes_bits;
      state.last_bytes = 0;
      state.last_bytes_bits = 0;
      bits |= (uint64_t)0x6u << nbits;
      nbits += 4;
      if (n == 0) {
        nbits += 2;
      } else {
        uint32_t lb = 1;
        if (n > 1) { uint32_t v = (uint32_t)n - 1; lb = 0; while (v) { ++lb; v >>= 1; } }
        const uint32_t nbytes = (lb + 7) / 8;
        bits |= (uint64_t)nbytes << nbits;
        nbits += 2;
        bits |= (uint64_t)(n - 1) << nbits;
        nbits += 8 * nbytes;
      }
      const size_t hb = (nbits + 7) >> 3;
      if (n_out + hb + n > out_cap) return -4;
      for (size_t i = 0; i < hb; ++i) out[n_out++] = (uint8_t)(bits >> (8 * i));
      memcpy(out + n_out, in + pos, n);
      n_out += n;
    }
    pos += n;
  }
  return (long)n_out;
}

// The decoder kernel (k_decode.h) over `npieces` pieces: pieces[k] = {in_off, in_len, out_off, out_cap, flags |
// lgwin << 32}; results[k] = {out_bytes, in_bits, error | finished << 32, lgwin | metablocks << 32}.
long sim_decode(const char* tables_path, const uint8_t* in, size_t in_len, const uint64_t* pieces, size_t npieces,
                uint32_t arena_words, int reverse, uint8_t* out, size_t out_cap, uint64_t* results) {
  // `reverse`: bit 0 = lane scheduling order of the simulator, bit 1 = DEC_ARG_NO_LDS_CACHE
  HostTables ht;
  HostTransforms tr;
  if (!host_tables_load(tables_path, &ht) || !host_transforms_load(tables_path, &tr)) return -1;
  std::vector<double> lut;
  DeviceTables T;
  host_tables_fill(ht, 16, &lut, &T);
  std::vector<uint8_t> input(in_len + 64, 0);
  memcpy(input.data(), in, in_len);
  std::vector<DecPiece> P(npieces);
  for (size_t k = 0; k < npieces; ++k) {
    P[k].in_off = pieces[5 * k]; P[k].in_len = pieces[5 * k + 1]; P[k].out_off = pieces[5 * k + 2];
    P[k].out_cap = pieces[5 * k + 3]; P[k].flags = (uint32_t)pieces[5 * k + 4]; P[k].lgwin = (uint32_t)(pieces[5 * k + 4] >> 32);
    if (P[k].in_off + P[k].in_len > in_len || P[k].out_off + P[k].out_cap > out_cap) return -2;
  }
  if (arena_words == 0) arena_words = dec_arena_words_max();
  std::vector<DecResult> R(npieces);
  std::vector<uint32_t> arena((size_t)npieces * arena_words, 0xCDCDCDCDu);
  DecArgs a;
  a.pieces = P.data();
  a.results = R.data();
  a.T = &T;
  a.transforms = (const DecTransform*)tr.records.data();
  a.transform_text = tr.text.data();
  a.input = input.data();
  a.out = out;
  a.arena = arena.data();
  a.arena_words = arena_words;
  a.npieces = (uint32_t)npieces;
  a.flags = (reverse & 2) ? DEC_ARG_NO_LDS_CACHE : 0u;
  a.pad = 0;
  struct DLaunch { DecArgs a; } l{a};
  simt::launch((unsigned)npieces, 64, [](void* p) { k_decode<4>(((DLaunch*)p)->a); }, &l, reverse & 1);
  for (size_t k = 0; k < npieces; ++k) {
    results[4 * k] = R[k].out_bytes;
    results[4 * k + 1] = R[k].in_bits;
    results[4 * k + 2] = R[k].error | ((uint64_t)R[k].finished << 32);
    results[4 * k + 3] = R[k].lgwin | ((uint64_t)R[k].metablocks << 32);
  }
  return 0;
}

}  // extern "C"
