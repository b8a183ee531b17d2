// tools/scatter_probe.hip — how well does the L2 of an XCD merge 8-byte stores scattered over a window before they
// leave for memory?  Emulates k_ix_bucket's res[p] writes: "shards" with a window of W bytes each, WPS waves per shard,
// every 8-byte slot of the window written exactly once by some wave of the shard, a wave's writes spread over the whole
// window; workgroup b runs on XCD b % 8 and all waves of a shard get the same b % 8 (as ix_bucket_kernel maps them).
// SPIN: s_sleep-ish delay between a wave's rows of 64 stores (time a line stays half written).
// Build: hipcc -O2 --offload-arch=gfx950 tools/scatter_probe.hip -o build/scatter_probe;  run under rocprofv3 --pmc WRITE_SIZE
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

template <int TAG>
__global__ void __launch_bounds__(64) k_scatter(uint64_t* out, uint32_t nshards, uint32_t wps, uint32_t slots_log2, uint32_t spin, uint32_t nt) {
  const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  const uint32_t shard = (slot / wps) * 8u + xcd, w = slot % wps;
  if (shard >= nshards) return;
  const uint32_t slots = 1u << slots_log2, per = slots / wps;
  uint64_t* win = out + ((uint64_t)shard << slots_log2);
  for (uint32_t i = threadIdx.x; i < per; i += 64u) {
    const uint32_t idx = w * per + i;
    const uint32_t s = (idx * 2654435761u + 12345u) & (slots - 1u);        // a bijection of [0, slots): odd multiplier
    if (nt) __builtin_nontemporal_store((uint64_t)idx | ((uint64_t)shard << 32), &win[s]);
    else win[s] = (uint64_t)idx | ((uint64_t)shard << 32);
    for (uint32_t k = 0; k < spin; ++k) __builtin_amdgcn_s_sleep(8);
  }
}

template <int TAG>
static void run(const char* name, uint64_t* d, uint32_t nshards, uint32_t wps, uint32_t slots_log2, uint32_t spin, uint32_t nt) {
  const uint32_t grid = ((nshards + 7u) / 8u) * 8u * wps;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_scatter<TAG>, dim3(grid), dim3(64), 0, 0, d, nshards, wps, slots_log2, spin, nt);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_scatter<TAG>, dim3(grid), dim3(64), 0, 0, d, nshards, wps, slots_log2, spin, nt);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)nshards * (8.0 * (1u << slots_log2));
  printf("k_scatter<%d> %-44s shards=%u window=%u KiB waves/shard=%u spin=%u nt=%u : %.3f ms, %.1f GB/s of payload (%.2f GB)\n",
         TAG, name, nshards, (8u << slots_log2) >> 10, wps, spin, nt, ms, bytes / 1e6 / ms, bytes / 1e9);
}

int main() {
  const uint64_t total = 8ull << 30;                    // 8 GiB of payload, as res[] of a 1 GiB job
  uint64_t* d;
  if (hipMalloc(&d, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(d, 0, total);
  // window 1 MiB (128 KiB shard), 256 waves per shard: what k_ix_bucket does at 2 buckets per wave
  run<0>("1 MiB window, 256 waves/shard", d, 8192, 256, 17, 0, 0);
  run<1>("1 MiB window, 512 waves/shard", d, 8192, 512, 17, 0, 0);
  run<2>("1 MiB window, 1024 waves/shard", d, 8192, 1024, 17, 0, 0);
  run<3>("1 MiB window, 128 waves/shard", d, 8192, 128, 17, 0, 0);
  run<4>("256 KiB window, 256 waves/shard", d, 32768, 256, 15, 0, 0);
  run<5>("256 KiB window, 64 waves/shard", d, 32768, 64, 15, 0, 0);
  run<6>("1 MiB window, 256 waves/shard, slow waves", d, 8192, 256, 17, 4, 0);
  run<7>("1 MiB window, 512 waves/shard, slow waves", d, 8192, 512, 17, 4, 0);
  run<8>("8 MiB window, 1024 waves/shard", d, 1024, 1024, 20, 0, 0);
  run<9>("1 MiB window, 256 waves/shard, nontemporal", d, 8192, 256, 17, 0, 1);
  run<10>("64 KiB window, 16 waves/shard", d, 131072, 16, 13, 0, 0);
  return 0;
}
