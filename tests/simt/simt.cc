// tests/simt/simt.cc — fiber scheduler for the host SIMT simulator (test only).
#include "simt.h"

#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>

namespace simt {

Dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

namespace {

struct Fiber {
  void* sp;
  char* stack;
  uint64_t round;
  bool done;
};

constexpr size_t kStack = 512 * 1024;
constexpr int kMaxThreads = 1024;
Fiber g_f[kMaxThreads];
int g_n, g_cur, g_reverse;
uint64_t g_slots[2][kMaxThreads];
int g_sites[2];
int g_arrived[2];
uint64_t g_completed;
void* g_main_sp;
void (*g_body)(void*);
void* g_arg;

extern "C" void simt_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl simt_switch
.type simt_switch,@function
simt_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
)");

void set_ids(int t) { g_threadIdx = Dim3((unsigned)t, 0, 0); }

int next_alive(int from) {
  for (int k = 1; k <= g_n; ++k) {
    int t = g_reverse ? ((from - k) % g_n + g_n) % g_n : (from + k) % g_n;
    if (!g_f[t].done) return t;
  }
  return -1;
}

void switch_to(int t) {
  int me = g_cur;
  g_cur = t;
  set_ids(t);
  simt_switch(&g_f[me].sp, g_f[t].sp);
  // resumed
  set_ids(g_cur);
}

void fiber_main() {
  g_body(g_arg);
  Fiber& f = g_f[g_cur];
  f.done = true;
  // A lane that exits while others are blocked at a rendezvous is a bug in
  // kernels that use cross-lane ops; lanes that never rendezvous just finish.
  int t = next_alive(g_cur);
  if (t < 0) {
    void* dummy;
    simt_switch(&dummy, g_main_sp);
  } else {
    if (g_arrived[0] || g_arrived[1]) {
      fprintf(stderr, "simt: lane %d exited while others wait at a rendezvous\n", g_cur);
      abort();
    }
    int me = g_cur;
    g_cur = t;
    set_ids(t);
    simt_switch(&g_f[me].sp, g_f[t].sp);
  }
  abort();
}

extern "C" void simt_trampoline() { fiber_main(); }

}  // namespace

void rendezvous(uint64_t in, int site) {
  Fiber& f = g_f[g_cur];
  const uint64_t r = f.round;
  const int par = (int)(r & 1);
  if (g_arrived[par] == 0) g_sites[par] = site;
  else if (g_sites[par] != site) {
    fprintf(stderr, "simt: divergent cross-lane op (lane %d at site %d, others at %d)\n",
            g_cur, site, g_sites[par]);
    abort();
  }
  g_slots[par][g_cur] = in;
  if (++g_arrived[par] == g_n) {
    g_arrived[par] = 0;
    g_completed = r + 1;
  } else {
    while (g_completed <= r) {
      int t = next_alive(g_cur);
      if (t < 0 || t == g_cur) {
        fprintf(stderr, "simt: deadlock at rendezvous site %d (lane %d)\n", site, g_cur);
        abort();
      }
      switch_to(t);
    }
  }
  f.round = r + 1;
}

uint64_t peek(int lane) {
  return g_slots[(g_f[g_cur].round - 1) & 1][lane];
}

void launch(unsigned grid, unsigned block, void (*body)(void*), void* arg, int reverse) {
  if (block > (unsigned)kMaxThreads) abort();
  g_body = body;
  g_arg = arg;
  g_reverse = reverse;
  g_gridDim = Dim3(grid);
  g_blockDim = Dim3(block);
  for (unsigned b = 0; b < grid; ++b) {
    g_blockIdx = Dim3(b);
    g_n = (int)block;
    g_arrived[0] = g_arrived[1] = 0;
    g_completed = 0;
    for (int t = 0; t < g_n; ++t) {
      Fiber& f = g_f[t];
      if (!f.stack) {
        f.stack = (char*)mmap(nullptr, kStack, PROT_READ | PROT_WRITE,
                              MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (f.stack == (char*)MAP_FAILED) abort();
      }
      uint64_t* top = (uint64_t*)(f.stack + kStack);
      // [r15 r14 r13 r12 rbx rbp ret pad]
      top[-1] = 0;
      top[-2] = (uint64_t)(void*)&simt_trampoline;
      for (int k = 3; k <= 8; ++k) top[-k] = 0;
      f.sp = (void*)(top - 8);
      f.round = 0;
      f.done = false;
    }
    g_cur = reverse ? g_n - 1 : 0;
    set_ids(g_cur);
    simt_switch(&g_main_sp, g_f[g_cur].sp);
  }
}

}  // namespace simt
