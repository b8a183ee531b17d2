// tests/simt/sim_driver.cc — TEST ONLY: runs the kernels of brotli_amd/csrc on
// the host SIMT simulator so their logic can be checked against the oracle
// in CI without a GPU.  Exposes a small C API for tests/test_sim_*.py.
#define BROTLI_AMD_SIMT_SIM 1
#include "simt.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

#include "../../brotli_amd/csrc/kernels.h"
#include "../../brotli_amd/csrc/host_plan.h"

namespace {
struct Launch { void (*fn)(JobArgs); JobArgs a; };
void tramp(void* p) { Launch* l = (Launch*)p; l->fn(l->a); }
void run(void (*fn)(JobArgs), const JobArgs& a, unsigned grid, unsigned block, int reverse) {
  Launch l{fn, a};
  simt::launch(grid, block, tramp, &l, reverse);
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
  if (no_pair) plan.J.flags |= JOB_FLAG_NO_PAIR;
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
  run(k_init, a, a.nshards * a.init_blocks_per_shard, 256, 0);
  run(k_parse, a, a.nshards, 64, reverse);
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

}  // extern "C"
