// tools/latency_probe.hip — dependent-load latency of one wave on gfx950, as a function of the
// footprint the chain wanders over (cache levels, TLB reach).  Evidence for DESIGN.md §5: the
// encoder chains are sequences of such dependent accesses.
//   hipcc --offload-arch=gfx950 -O2 tools/latency_probe.hip -o build/var/latency_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <random>
#include <vector>

__global__ void chase(const uint64_t* buf, uint64_t start, uint32_t steps, uint64_t* out, int all_lanes) {
  uint64_t p = start + (all_lanes ? threadIdx.x : 0);
  if (!all_lanes && threadIdx.x != 0) return;
  for (uint32_t i = 0; i < steps; ++i) p = buf[p];
  out[threadIdx.x] = p;
}

__global__ void link(uint64_t* buf, const uint64_t* nodes, uint64_t nn) {
  const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i >= nn) return;
  const uint64_t nxt = nodes[(i + 1) % nn] * 16;
  for (int k = 0; k < 16; ++k) buf[nodes[i] * 16 + k] = nxt;
}

int main() {
  const uint64_t sizes[] = {1ull << 20, 4ull << 20, 64ull << 20, 1ull << 30, 16ull << 30, 64ull << 30};
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  uint64_t* d_out; hipMalloc(&d_out, 64 * 8);
  for (uint64_t bytes : sizes) {
    uint64_t* d; if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc %llu failed\n", (unsigned long long)bytes); continue; }
    // nodes: up to 1M distinct 128-byte lines spread over the region, one random cycle through them
    const uint64_t lines = bytes / 128;
    const uint64_t nn = std::min<uint64_t>(lines, 1u << 20);
    std::mt19937_64 rng(12345);
    std::vector<uint64_t> node(nn);
    if (nn == lines) { for (uint64_t i = 0; i < nn; ++i) node[i] = i; }
    else { for (uint64_t i = 0; i < nn; ++i) node[i] = (rng() % (lines / nn)) + i * (lines / nn); }
    std::shuffle(node.begin(), node.end(), rng);
    // buf[node[i] * 16 + k] = node[i + 1] * 16 (u64 units): every word of a line points to the next line
    uint64_t* d_nodes;
    hipMalloc(&d_nodes, nn * 8);
    hipMemcpy(d_nodes, node.data(), nn * 8, hipMemcpyHostToDevice);
    link<<<(unsigned)((nn + 255) / 256), 256>>>(d, d_nodes, nn);
    hipFree(d_nodes);
    hipDeviceSynchronize();
    for (int all = 0; all < 2; ++all) {
      const uint32_t steps = 200000;
      chase<<<1, 64>>>(d, node[0] * 16, 1000, d_out, all);   // warm
      hipEventRecord(e0);
      chase<<<1, 64>>>(d, node[0] * 16, steps, d_out, all);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("footprint %8.1f MiB  nodes %7llu  %s: %.1f ns per dependent load\n", bytes / 1048576.0,
             (unsigned long long)nn, all ? "64 lanes (same line)" : "1 lane", ms * 1e6 / steps);
    }
    hipFree(d);
  }
  return 0;
}
