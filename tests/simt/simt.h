// tests/simt/simt.h — TEST INFRASTRUCTURE ONLY.
//
// A tiny host-side SIMT simulator used to exercise the *logic* of the HIP
// kernels in brotli_amd/csrc on a machine without a GPU (this container).
// One 64-lane wavefront = 64 fibers on one OS thread.  A fiber runs until it
// reaches a cross-lane operation (ballot / shuffle / wave_sync / barrier),
// where all 64 lanes rendezvous.  Between rendezvous points lanes run one
// after another (lane 0 first, or lane 63 first in "reverse" mode), so any
// inter-lane communication through memory that is not separated by a
// wave_sync()/__syncthreads() shows up as a wrong result in one of the two
// orders.  This is stricter than the hardware; it is not a performance model.
//
// The product never includes this file: brotli_amd/csrc/wave.h pulls it in
// only when BROTLI_AMD_SIMT_SIM is defined, which only tests/simt/Makefile does.
#ifndef BROTLI_AMD_TESTS_SIMT_H_
#define BROTLI_AMD_TESTS_SIMT_H_

#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__ __restrict

namespace simt {

struct Dim3 {
  unsigned x, y, z;
  Dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};

extern Dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

// Rendezvous of all lanes of the running block.  Each lane deposits `in`
// (8 bytes); returns after every lane has arrived.  `site` tags the call site
// so divergent use is detected.  After return `peek(lane)` gives any lane's
// deposited value for this round.
void rendezvous(uint64_t in, int site);
uint64_t peek(int lane);

// Runs `body(arg)` for every thread of a grid of `grid` blocks x `block`
// threads (block <= 1024, multiple of 64 or < 64).  reverse=1 schedules lanes
// high-to-low.
void launch(unsigned grid, unsigned block, void (*body)(void*), void* arg,
            int reverse);

}  // namespace simt

#define threadIdx (simt::g_threadIdx)
#define blockIdx (simt::g_blockIdx)
#define blockDim (simt::g_blockDim)
#define gridDim (simt::g_gridDim)

// ---- cross-lane operations (device spelling lives in wave.h) --------------

static inline uint64_t simt_ballot(bool p, int site) {
  simt::rendezvous(p ? 1 : 0, site);
  uint64_t m = 0;
  unsigned n = simt::g_blockDim.x < 64 ? simt::g_blockDim.x : 64;
  unsigned base = (simt::g_threadIdx.x / 64) * 64;
  for (unsigned i = 0; i < n; ++i)
    if (simt::peek((int)(base + i))) m |= 1ull << i;
  return m;
}

static inline uint64_t simt_shfl64(uint64_t v, int src, int site) {
  simt::rendezvous(v, site);
  unsigned base = (simt::g_threadIdx.x / 64) * 64;
  return simt::peek((int)(base + (src & 63)));
}

static inline void simt_sync(int site) { simt::rendezvous(0, site); }

#endif  // BROTLI_AMD_TESTS_SIMT_H_
