// brotli_amd/csrc/wave.h — wavefront-level primitives for gfx950 (wave64).
//
// Every kernel in this directory is written for ONE 64-lane wavefront per
// workgroup owning one encoder shard: control flow is wave-uniform, the
// cross-lane operations below are only ever called with all 64 lanes active.
//
// tests/simt (a host fiber simulator used to check kernel *logic* in CI
// without a GPU) provides the same names when BROTLI_AMD_SIMT_SIM is defined;
// the shipped library is always built without it.
#ifndef BROTLI_AMD_CSRC_WAVE_H_
#define BROTLI_AMD_CSRC_WAVE_H_

#include <stdint.h>

#if defined(BROTLI_AMD_SIMT_SIM)

#include "simt.h"
#define WAVE_SITE __LINE__
#define wave_ballot(p) simt_ballot((p), WAVE_SITE)
#define wave_shfl(v, src) ((uint32_t)simt_shfl64((uint64_t)(uint32_t)(v), (src), WAVE_SITE))
#define wave_shfl64(v, src) simt_shfl64((uint64_t)(v), (src), WAVE_SITE)
#define wave_bcast(v, lane) ((uint32_t)simt_shfl64((uint64_t)(uint32_t)(v), (lane), WAVE_SITE))
#define wave_bcast64(v, lane) simt_shfl64((uint64_t)(v), (lane), WAVE_SITE)
#define wave_sync() simt_sync(WAVE_SITE)
#define wave_mem_barrier() simt_sync(WAVE_SITE)
static inline int wave_lane() { return (int)(threadIdx.x & 63); }
static inline int dev_ctz64(uint64_t x) { return __builtin_ctzll(x); }
static inline int dev_ctz32(uint32_t x) { return __builtin_ctz(x); }
static inline int dev_clz32(uint32_t x) { return __builtin_clz(x); }
static inline int dev_popc64(uint64_t x) { return __builtin_popcountll(x); }
static inline uint32_t dev_mul24(uint32_t a, uint32_t b) { return a * b; }   // (both < 2^24)
// Index of the lowest set bit, 0xFFFFFFFF for 0 (v_ffbl_b32 on the device).
static inline uint32_t dev_ffbl32(uint32_t x) { return x ? (uint32_t)__builtin_ctz(x) : 0xFFFFFFFFu; }
template <class T> static inline T lds_atomic_add(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T lds_atomic_or(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T lds_atomic_max(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
static inline uint32_t glb_atomic_or(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o | v; return o; }
static inline uint32_t glb_atomic_add(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
static inline uint32_t glb_atomic_and(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o & v; return o; }
static inline uint32_t glb_atomic_max(uint32_t* p, uint32_t v) { uint32_t o = *p; if (v > o) *p = v; return o; }
static inline uint32_t glb_load_l2(const uint32_t* p) { return *p; }
static inline uint32_t dev_bitrev32(uint32_t x) {
  x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
  x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
  x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
  return __builtin_bswap32(x);
}
#define block_sync() simt_sync(WAVE_SITE)
// Lanes [0, n) hold v: prev = highest lane below this one with the same v (-1 if
// none), next = lowest lane above it with the same v (64 if none).
static inline void wave_equal_neighbours(uint32_t v, int n, int* prev, int* next) {
  simt::rendezvous(v, 7001);
  const int lane = wave_lane();
  int p = -1, x = 64;
  for (int k = 0; k < n; ++k) {
    if ((uint32_t)simt::peek((int)((threadIdx.x & ~63u) + k)) != v) continue;
    if (k < lane) p = k; else if (k > lane && x == 64) x = k;
  }
  *prev = p; *next = x;
}
// Rotation inside rows of 16 lanes (DPP row_ror on the device: lane i reads lane (i - k) & 15 of
// its row, measured with tools/dpp_probe.hip).
#define wave_row_ror(v, k) ((uint32_t)simt_shfl64((uint64_t)(uint32_t)(v), (wave_lane() & 48) | ((wave_lane() - (k)) & 15), WAVE_SITE))
// Maximum of v over the wave (uniform result).
static inline uint32_t wave_max_u32(uint32_t v) {
  simt::rendezvous(v, 7002);
  uint32_t m = 0;
  for (int k = 0; k < 64; ++k) { const uint32_t x = (uint32_t)simt::peek((int)((threadIdx.x & ~63u) + k)); if (x > m) m = x; }
  return m;
}
// Exchange inside quads of 4 lanes (DPP quad_perm on the device).
#define wave_quad_xor(v, m) ((uint32_t)simt_shfl64((uint64_t)(uint32_t)(v), wave_lane() ^ (m), WAVE_SITE))

#else  // ---- gfx950 ---------------------------------------------------------

#include <hip/hip_runtime.h>

// The cross-lane code of this directory (row_ror / row_shr / row_bcast DPP controls, 64-lane ballots, v_readlane) is
// wave64 GCN / CDNA code: another target would assemble it wrongly or not at all, so it is refused here.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "brotli_amd kernels are written for gfx950 (wave64, DPP row_bcast): build with --offload-arch=gfx950"
#endif

__device__ __forceinline__ int wave_lane() { return (int)(threadIdx.x & 63); }

// 64-bit mask of lanes whose predicate holds (s_and / v_cmp into an SGPR pair).
__device__ __forceinline__ uint64_t wave_ballot(bool p) { return __ballot(p); }

// Arbitrary per-lane source (ds_bpermute_b32).
__device__ __forceinline__ uint32_t wave_shfl(uint32_t v, int src) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)v);
}
__device__ __forceinline__ uint64_t wave_shfl64(uint64_t v, int src) {
  uint32_t lo = wave_shfl((uint32_t)v, src), hi = wave_shfl((uint32_t)(v >> 32), src);
  return ((uint64_t)hi << 32) | lo;
}

// Wave-uniform source lane (v_readlane_b32 -> SGPR): the result is uniform.
__device__ __forceinline__ uint32_t wave_bcast(uint32_t v, int lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
__device__ __forceinline__ uint64_t wave_bcast64(uint64_t v, int lane) {
  uint32_t lo = wave_bcast((uint32_t)v, lane), hi = wave_bcast((uint32_t)(v >> 32), lane);
  return ((uint64_t)hi << 32) | lo;
}

// Orders this wave's earlier global/LDS accesses before its later ones as seen
// by the other lanes of the same wave.  Wavefront scope needs no cache action
// or s_waitcnt on gfx950 (one wave issues its memory instructions in order to
// one L1); it stops the compiler from moving accesses across.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Waits until this wave's earlier stores / atomics have reached the device
// level (L2) and drops stale L1 lines: used a few times per meta-block where
// plain stores, dword atomics and plain loads touch the same output bytes.
__device__ __forceinline__ void wave_mem_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int dev_ctz64(uint64_t x) { return __builtin_ctzll(x); }
__device__ __forceinline__ int dev_ctz32(uint32_t x) { return __builtin_ctz(x); }
__device__ __forceinline__ int dev_clz32(uint32_t x) { return __builtin_clz(x); }
__device__ __forceinline__ int dev_popc64(uint64_t x) { return __builtin_popcountll(x); }
// Product of two values below 2^24: v_mul_u32_u24 (full rate; v_mul_lo_u32 runs at a quarter).
__device__ __forceinline__ uint32_t dev_mul24(uint32_t a, uint32_t b) { return __umul24(a, b); }
// Index of the lowest set bit, 0xFFFFFFFF for 0: one v_ffbl_b32, no select around it.
__device__ __forceinline__ uint32_t dev_ffbl32(uint32_t x) {
  uint32_t r;
  asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
template <class T> __device__ __forceinline__ T lds_atomic_add(T* p, T v) { return atomicAdd(p, v); }
template <class T> __device__ __forceinline__ T lds_atomic_or(T* p, T v) { return atomicOr(p, v); }
template <class T> __device__ __forceinline__ T lds_atomic_max(T* p, T v) { return atomicMax(p, v); }
// Device-scope OR on a global dword (executed at the L2; result optional).
__device__ __forceinline__ uint32_t glb_atomic_or(uint32_t* p, uint32_t v) { return atomicOr(p, v); }
__device__ __forceinline__ uint32_t glb_atomic_add(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
__device__ __forceinline__ uint32_t glb_atomic_and(uint32_t* p, uint32_t v) { return atomicAnd(p, v); }
__device__ __forceinline__ uint32_t glb_atomic_max(uint32_t* p, uint32_t v) { return atomicMax(p, v); }
// A load served by the L2 (not the CU's L1): sees this wave's earlier atomics on the same word.
__device__ __forceinline__ uint32_t glb_load_l2(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t dev_bitrev32(uint32_t x) { return __builtin_bitreverse32(x); }
__device__ __forceinline__ void block_sync() { __syncthreads(); }
// Lanes [0, n) hold v: prev = highest lane below this one with the same v (-1 if
// none), next = lowest lane above it with the same v (64 if none).  n readlanes
// (SGPR broadcast) and two compares each: no LDS traffic.
__device__ __forceinline__ void wave_equal_neighbours(uint32_t v, int n, int* prev, int* next) {
  const int lane = wave_lane();
  int p = -1, x = 64;
  for (int k = 0; k < n; ++k) {
    const uint32_t vk = (uint32_t)__builtin_amdgcn_readlane((int)v, k);
    if (vk == v) {
      if (k < lane) p = k; else if (k > lane && x == 64) x = k;
    }
  }
  *prev = p; *next = x;
}
// Rotation inside rows of 16 lanes: one VALU v_mov_b32 with DPP row_ror:k (lane i reads lane
// (i - k) & 15 of its row), no LDS crossbar round trip (ds_bpermute costs ~100 cycles of latency each).
#define wave_row_ror(v, k) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), 0x120 + (k), 0xF, 0xF, true))
// Maximum of v over the wave (uniform result): DPP row reductions + two readlanes.
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  uint32_t o;
  o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true); v = o > v ? o : v;   // row_ror:8
  o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, true); v = o > v ? o : v;   // row_ror:4
  o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xF, 0xF, true); v = o > v ? o : v;   // row_ror:2
  o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xF, 0xF, true); v = o > v ? o : v;   // row_ror:1
  const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
  const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
  const uint32_t ab = a > b ? a : b, cd = c > d ? c : d;
  return ab > cd ? ab : cd;
}
// Value of lane ^ m (m = 1 or 2) inside quads of 4 lanes: DPP quad_perm [1,0,3,2] / [2,3,0,1].
#define wave_quad_xor(v, m) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (m) == 1 ? 0xB1 : 0x4E, 0xF, 0xF, true))

#endif

#define DEV __device__ __forceinline__

// The value x, in a form the optimizer cannot look through (an empty asm statement that "modifies" it; nothing on the
// simulator).  Used where a choice between LOADED values must stay a choice between values: the optimizer otherwise
// folds select(load a[1], load a[2]) into a load through a computed address, and a struct that is indexed through a
// computed address cannot be kept in registers (k_store.h, sel3).
#if defined(BROTLI_AMD_SIMT_SIM)
template <class T> static inline T dev_opaque(T x) { return x; }
#else
template <class T> __device__ __forceinline__ T dev_opaque(T x) { asm volatile("" : "+v"(x)); return x; }
#endif

#endif  // BROTLI_AMD_CSRC_WAVE_H_
