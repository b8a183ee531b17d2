// tools/dpp_probe.hip — prints which lane a DPP row operation reads from (gfx950), to pin the
// direction conventions used in brotli_amd/csrc/wave.h.  hipcc --offload-arch=gfx950 -o build/dpp_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
  const int lane = threadIdx.x;
  out[lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x121, 0xF, 0xF, true);        // row_ror:1
  out[64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x101, 0xF, 0xF, true);   // row_shl:1
  out[128 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x111, 0xF, 0xF, true);  // row_shr:1
  out[192 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x12F, 0xF, 0xF, true);  // row_ror:15
}
int main() {
  int* d; int h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"row_ror:1", "row_shl:1", "row_shr:1", "row_ror:15"};
  for (int r = 0; r < 4; ++r) {
    printf("%s: lane reads", names[r]);
    for (int i = 0; i < 20; ++i) printf(" %d", h[64 * r + i]);
    printf("\n");
  }
  return 0;
}
