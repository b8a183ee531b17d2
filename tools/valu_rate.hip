// tools/valu_rate.hip — issue cost of the integer instructions the index / chain kernels are made of, on the device
// they run on: cycles a SIMD is busy per wave-instruction, measured with 8 waves per SIMD in flight (latency hidden)
// and four independent dependency chains per wave.  Build: hipcc -O2 --offload-arch=gfx950 tools/valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define REP16(X) X X X X X X X X X X X X X X X X

#define DEF_KERNEL(NAME, ASM, ...)                                                             \
  __global__ void __launch_bounds__(64) NAME(uint32_t* out, int iters, uint32_t seed) {       \
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B9u, c = a + 77u, d = b + 1234567u; \
    uint32_t e = a | 1u, f = b | 3u;                                                           \
    uint64_t p = ((uint64_t)a << 32) | b, q = ((uint64_t)c << 32) | d;                         \
    __shared__ uint32_t lds[2048];                                                             \
    lds[threadIdx.x] = a; lds[threadIdx.x + 64] = b;                                           \
    __syncthreads();                                                                           \
    uint32_t la = (threadIdx.x * 4u) & 1023u, lb = ((threadIdx.x * 8u) & 1023u);               \
    for (int i = 0; i < iters; ++i) {                                                          \
      REP16(asm volatile(ASM : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(p), "+v"(q) : "v"(e), "v"(f), "v"(la), "v"(lb) : __VA_ARGS__);) \
    }                                                                                          \
    out[blockIdx.x * 64 + threadIdx.x] = a ^ b ^ c ^ d ^ (uint32_t)p ^ (uint32_t)q;           \
  }

// every ASM body holds FOUR instructions on four chains (a, b, c, d / p, q) => 64 instructions per loop iteration
DEF_KERNEL(k_add,      "v_add_u32 %0, %0, %6\n v_add_u32 %1, %1, %7\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %7", "memory")
DEF_KERNEL(k_xor,      "v_xor_b32 %0, %0, %6\n v_xor_b32 %1, %1, %7\n v_xor_b32 %2, %2, %6\n v_xor_b32 %3, %3, %7", "memory")
DEF_KERNEL(k_mul_lo,   "v_mul_lo_u32 %0, %0, %6\n v_mul_lo_u32 %1, %1, %7\n v_mul_lo_u32 %2, %2, %6\n v_mul_lo_u32 %3, %3, %7", "memory")
DEF_KERNEL(k_mul_hi,   "v_mul_hi_u32 %0, %0, %6\n v_mul_hi_u32 %1, %1, %7\n v_mul_hi_u32 %2, %2, %6\n v_mul_hi_u32 %3, %3, %7", "memory")
DEF_KERNEL(k_mul_u24,  "v_mul_u32_u24 %0, %0, %6\n v_mul_u32_u24 %1, %1, %7\n v_mul_u32_u24 %2, %2, %6\n v_mul_u32_u24 %3, %3, %7", "memory")
DEF_KERNEL(k_mad_u24,  "v_mad_u32_u24 %0, %0, %6, %7\n v_mad_u32_u24 %1, %1, %7, %6\n v_mad_u32_u24 %2, %2, %6, %7\n v_mad_u32_u24 %3, %3, %7, %6", "memory")
DEF_KERNEL(k_mad_u64,  "v_mad_u64_u32 %4, vcc, %0, %6, %4\n v_mad_u64_u32 %5, vcc, %1, %7, %5\n v_mad_u64_u32 %4, vcc, %2, %6, %4\n v_mad_u64_u32 %5, vcc, %3, %7, %5", "memory", "vcc")
DEF_KERNEL(k_lshl_b64, "v_lshlrev_b64 %4, 1, %4\n v_lshlrev_b64 %5, 1, %5\n v_lshlrev_b64 %4, 3, %4\n v_lshlrev_b64 %5, 3, %5", "memory")
DEF_KERNEL(k_cmp_u64,  "v_cmp_ne_u64 s[20:21], %4, %5\n v_cmp_ne_u64 s[22:23], %5, %4\n v_cmp_eq_u64 s[24:25], %4, %5\n v_cmp_eq_u64 s[26:27], %5, %4", "memory", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27")
DEF_KERNEL(k_cmp_u32,  "v_cmp_ne_u32 s[20:21], %0, %1\n v_cmp_ne_u32 s[22:23], %1, %2\n v_cmp_eq_u32 s[24:25], %2, %3\n v_cmp_eq_u32 s[26:27], %3, %0", "memory", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27")
DEF_KERNEL(k_cmp_vcc,  "v_cmp_ne_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %6, vcc\n v_cmp_ne_u32 vcc, %2, %3\n v_cndmask_b32 %2, %2, %7, vcc", "memory", "vcc")
DEF_KERNEL(k_ffbl,     "v_ffbl_b32 %0, %0\n v_ffbl_b32 %1, %1\n v_ffbh_u32 %2, %2\n v_ffbh_u32 %3, %3", "memory")
DEF_KERNEL(k_bfe,      "v_bfe_u32 %0, %0, 3, 9\n v_bfe_u32 %1, %1, 5, 7\n v_lshl_or_b32 %2, %2, 3, %6\n v_lshl_or_b32 %3, %3, 5, %7", "memory")
DEF_KERNEL(k_alignbyte,"v_alignbyte_b32 %0, %0, %6, 1\n v_alignbyte_b32 %1, %1, %7, 3\n v_perm_b32 %2, %2, %6, %7\n v_perm_b32 %3, %3, %7, %6", "memory")
DEF_KERNEL(k_min3,     "v_min3_u32 %0, %0, %6, %7\n v_max3_u32 %1, %1, %6, %7\n v_min_u32 %2, %2, %6\n v_max_u32 %3, %3, %7", "memory")
DEF_KERNEL(k_add64,    "v_add_co_u32 %0, vcc, %0, %6\n v_addc_co_u32 %1, vcc, %1, %7, vcc\n v_add_co_u32 %2, vcc, %2, %6\n v_addc_co_u32 %3, vcc, %3, %7, vcc", "memory", "vcc")
DEF_KERNEL(k_dpp,      "v_mov_b32_dpp %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_ror:2 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "memory")
DEF_KERNEL(k_sdwa,     "v_add_u32_sdwa %0, %0, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %1, %1, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n v_add_u32_sdwa %2, %2, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %3, %3, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD", "memory")
DEF_KERNEL(k_bperm,    "ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %9, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %9, %3\n s_waitcnt lgkmcnt(0)", "memory")
DEF_KERNEL(k_ds_r32,   "ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n s_waitcnt lgkmcnt(0)", "memory")
DEF_KERNEL(k_ds_r64,   "ds_read_b64 %4, %9\n ds_read_b64 %5, %9 offset:1024\n ds_read_b64 %4, %9 offset:2048\n ds_read_b64 %5, %9 offset:3072\n s_waitcnt lgkmcnt(0)", "memory")
DEF_KERNEL(k_ds_r128,  "ds_read_b128 v[40:43], %9\n ds_read_b128 v[44:47], %9 offset:1024\n ds_read_b128 v[40:43], %9 offset:2048\n ds_read_b128 v[44:47], %9 offset:3072\n s_waitcnt lgkmcnt(0)", "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47")
DEF_KERNEL(k_readlane, "v_readlane_b32 s20, %0, 5\n v_readlane_b32 s21, %1, 9\n v_readfirstlane_b32 s22, %2\n v_readfirstlane_b32 s23, %3", "memory", "s20", "s21", "s22", "s23")
DEF_KERNEL(k_salu,     "s_add_u32 s20, s20, 1\n s_and_b32 s21, s21, s20\n s_lshl_b32 s22, s22, 1\n s_or_b32 s23, s23, s20", "memory", "s20", "s21", "s22", "s23", "scc")
DEF_KERNEL(k_valu_salu,"v_add_u32 %0, %0, %6\n s_add_u32 s20, s20, 1\n v_add_u32 %1, %1, %7\n s_and_b32 s21, s21, s20", "memory", "s20", "s21", "scc")
DEF_KERNEL(k_saveexec, "v_cmp_ne_u32 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n v_add_u32 %0, %0, %6\n s_or_b64 exec, exec, s[20:21]", "memory", "vcc", "s20", "s21")
DEF_KERNEL(k_mbcnt,    "v_mbcnt_lo_u32_b32 %0, %6, %0\n v_mbcnt_hi_u32_b32 %1, %7, %1\n v_bcnt_u32_b32 %2, %6, %2\n v_bcnt_u32_b32 %3, %7, %3", "memory")

struct Test { const char* name; void (*fn)(uint32_t*, int, uint32_t); int per_body; };

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const double mhz = prop.clockRate / 1000.0;
  printf("device %s, %d CUs, clock %.0f MHz (nominal)\n", prop.name, cus, mhz);
  const int waves_per_simd = 8, grid = cus * 4 * waves_per_simd, iters = 4000;
  uint32_t* out;
  hipMalloc(&out, (size_t)grid * 64 * 4);
  std::vector<Test> T = {
    {"v_add_u32", k_add, 4}, {"v_xor_b32", k_xor, 4}, {"v_mul_lo_u32", k_mul_lo, 4}, {"v_mul_hi_u32", k_mul_hi, 4},
    {"v_mul_u32_u24", k_mul_u24, 4}, {"v_mad_u32_u24", k_mad_u24, 4}, {"v_mad_u64_u32", k_mad_u64, 4},
    {"v_lshlrev_b64", k_lshl_b64, 4}, {"v_cmp_*_u64 (sgpr)", k_cmp_u64, 4}, {"v_cmp_*_u32 (sgpr)", k_cmp_u32, 4},
    {"v_cmp vcc + v_cndmask", k_cmp_vcc, 4}, {"v_ffbl / v_ffbh", k_ffbl, 4}, {"v_bfe / v_lshl_or", k_bfe, 4},
    {"v_alignbyte / v_perm", k_alignbyte, 4}, {"v_min3 / max3 / min / max", k_min3, 4}, {"v_add_co + v_addc_co", k_add64, 4},
    {"v_mov_dpp", k_dpp, 4}, {"v_add_u32_sdwa", k_sdwa, 4}, {"ds_bpermute_b32", k_bperm, 4}, {"ds_read_b32", k_ds_r32, 4},
    {"ds_read_b64", k_ds_r64, 4}, {"ds_read_b128", k_ds_r128, 4}, {"v_readlane / readfirstlane", k_readlane, 4},
    {"salu", k_salu, 4}, {"2 valu + 2 salu", k_valu_salu, 4}, {"cmp + saveexec + add + or exec", k_saveexec, 4},
    {"v_mbcnt / v_bcnt", k_mbcnt, 4},
  };
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  // clock: calibrate with v_add_u32 assumed 4 cycles per wave-instruction per SIMD
  double add_ms = 0;
  for (auto& t : T) {
    hipLaunchKernelGGL(t.fn, dim3(grid), dim3(64), 0, 0, out, 10, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(t.fn, dim3(grid), dim3(64), 0, 0, out, iters, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)waves_per_simd * iters * 16.0 * t.per_body;
    const double ns_per = ms * 1e6 / instr_per_simd;
    if (add_ms == 0) add_ms = ns_per;
    printf("%-32s %8.3f ms  %6.3f ns per wave-instruction per SIMD = %5.2f x v_add_u32 (= %4.1f cycles if v_add is 4)\n",
           t.name, ms, ns_per, ns_per / add_ms, 4.0 * ns_per / add_ms);
  }
  return 0;
}
