// tools/scan_probe.hip — the DPP form of wave_incl_scan (device_common.h) against a serial prefix sum, on the device.
// Build: hipcc -O2 --offload-arch=gfx950 -Ibrotli_amd/csrc -Iinclude tools/scan_probe.hip -o build/scan_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "device_common.h"

__global__ void k_scan(const uint32_t* in, uint32_t* out, int rows) {
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const uint32_t v = in[r * 64 + threadIdx.x];
    // (the second scan starts from the first one's result: back-to-back DPP chains, as in k_store)
    const uint32_t a = wave_incl_scan(v);
    const uint32_t b = wave_incl_scan(a & 0xFFFFu);
    out[(r * 64 + threadIdx.x) * 2] = a;
    out[(r * 64 + threadIdx.x) * 2 + 1] = b;
  }
}

int main() {
  const int rows = 4096;
  std::vector<uint32_t> h(rows * 64), o(rows * 128);
  uint64_t s = 88172645463325252ull;
  for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (uint32_t)(s >> 20) & ((s & 1) ? 0xFFFFFFFFu : 0xFFu); }
  uint32_t *di, *dout;
  hipMalloc(&di, h.size() * 4); hipMalloc(&dout, o.size() * 4);
  hipMemcpy(di, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_scan, dim3(256), dim3(64), 0, 0, di, dout, rows);
  hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int r = 0; r < rows; ++r) {
    uint32_t acc = 0, acc2 = 0;
    for (int l = 0; l < 64; ++l) {
      acc += h[r * 64 + l];
      acc2 += acc & 0xFFFFu;
      if (o[(r * 64 + l) * 2] != acc || o[(r * 64 + l) * 2 + 1] != acc2) { if (bad < 5) printf("row %d lane %d: got %u / %u want %u / %u\n", r, l, o[(r * 64 + l) * 2], o[(r * 64 + l) * 2 + 1], acc, acc2); ++bad; }
    }
  }
  printf("wave_incl_scan (DPP): %d mismatches in %d rows\n", bad, rows);
  return bad != 0;
}
