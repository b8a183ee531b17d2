// tools/write_calib.hip — calibration of rocprofv3's WRITE_SIZE / FETCH_SIZE on gfx950 for the access patterns of
// k_ix_bucket (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access pattern"):
//   A  coalesced 16-byte-per-lane stores of N bytes                    (known: N bytes written)
//   B  8-byte stores, every 8-byte slot of N bytes exactly once, in a scattered order inside 1 MiB windows
//      (what res[p] of a 128 KiB shard gets: 131072 slots of 8 bytes, hit in (key, position) order)     (known: N)
//   C  4-byte coalesced loads of N bytes                                 (known: N bytes read)
//   D  16-byte gathers from random places of a 128 KiB window, N / 16 of them   (known: N bytes requested, L2 hits)
// Run under `rocprofv3 --pmc WRITE_SIZE --kernel-trace` and `--pmc FETCH_SIZE --kernel-trace`; tools/gpu_r03_f.sh
// prints counter / known for each kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void calib_a_coalesced_store16(uint4* out, uint64_t n16) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n16) out[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
__global__ void calib_b_scatter_store8(uint64_t* out, uint64_t n8) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const uint64_t win = i >> 17, k = i & 131071u;
  const uint64_t slot = (k * 40503u + 12345u) & 131071u;          // odd multiplier: a permutation of the window's slots
  out[(win << 17) + slot] = i;
}
__global__ void calib_c_coalesced_load4(const uint32_t* in, uint32_t* sink, uint64_t n4) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t v = 0;
  if (i < n4) v = in[i];
  if (v == 0x12345679u) sink[0] = v;
}
__global__ void calib_d_gather16(const uint8_t* in, uint32_t* sink, uint64_t n16) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n16) return;
  const uint64_t win = (i >> 13) << 17, k = i & 8191u;            // 8192 gathers per 128 KiB window
  const uint64_t at = win + ((k * 2654435761u) & 131071u & ~15ull);
  uint4 v;
  __builtin_memcpy(&v, in + at, 16);
  if (v.x == 0x12345679u && v.y == 77u) sink[0] = v.z;
}

int main(int argc, char** argv) {
  const uint64_t n = (argc > 1 ? strtoull(argv[1], nullptr, 10) : 1024ull) << 20;   // MiB
  uint8_t* buf = nullptr;
  uint32_t* sink = nullptr;
  if (hipMalloc((void**)&buf, n + 64) != hipSuccess || hipMalloc((void**)&sink, 64) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
  hipMemset(buf, 0, n + 64);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(calib_a_coalesced_store16, dim3((unsigned)((n / 16 + 255) / 256)), dim3(256), 0, 0, (uint4*)buf, n / 16);
    hipLaunchKernelGGL(calib_b_scatter_store8, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, 0, (uint64_t*)buf, n / 8);
    hipLaunchKernelGGL(calib_c_coalesced_load4, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, 0, (const uint32_t*)buf, sink, n / 4);
    hipLaunchKernelGGL(calib_d_gather16, dim3((unsigned)((n / 16 + 255) / 256)), dim3(256), 0, 0, buf, sink, n / 16);
  }
  hipDeviceSynchronize();
  printf("calibration kernels ran on %llu MiB\n", (unsigned long long)(n >> 20));
  return 0;
}
