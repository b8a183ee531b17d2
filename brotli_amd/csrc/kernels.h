// brotli_amd/csrc/kernels.h — __global__ entry points.  One 64-lane wavefront
// (= one workgroup) per shard for the encoder stages; plain data-parallel
// grids for table initialisation and output gathering.
#ifndef BROTLI_AMD_CSRC_KERNELS_H_
#define BROTLI_AMD_CSRC_KERNELS_H_

#include "k_round.h"

struct JobArgs {
  JobParams J;
  const ShardDesc* shards;
  ShardState* states;
  const DeviceTables* T;
  const uint8_t* input;
  uint8_t* ws;
  uint32_t nshards;
  uint32_t init_blocks_per_shard;
};

// grid = nshards * init_blocks_per_shard, block = 256
__global__ void __launch_bounds__(256) k_init(JobArgs a) {
  const uint32_t shard = blockIdx.x / a.init_blocks_per_shard;
  const uint32_t b = blockIdx.x % a.init_blocks_per_shard;
  if (shard >= a.nshards) return;
  const ShardDesc& D = a.shards[shard];
  init_shard_table(a.ws + D.table_off, 1u << a.J.bucket_bits,
                   b * blockDim.x + threadIdx.x, a.init_blocks_per_shard * blockDim.x);
  if (b == 0 && threadIdx.x == 0) init_shard_state(a.J, D, &a.states[shard]);
}

// grid = nshards, block = 64
__global__ void __launch_bounds__(64) k_parse(JobArgs a) {
  const uint32_t shard = blockIdx.x;
  if (shard >= a.nshards) return;
  parse_round(a.J, a.shards[shard], &a.states[shard], a.T, a.input, a.ws);
}

#endif  // BROTLI_AMD_CSRC_KERNELS_H_
