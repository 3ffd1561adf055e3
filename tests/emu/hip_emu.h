// TEST INFRASTRUCTURE — a minimal host-side SIMT emulator for the HIP kernels in
// ft-fsd-path-planning_amd/csrc/.  It lets the CPU test-suite execute the *kernel source*
// (one fiber per lane, 64 lanes per block, cooperative scheduling, cross-lane ops realised
// as rendezvous) so kernel logic can be checked against the oracle where no GPU exists.
// It is NOT a product fallback: nothing in the package loads the library built from this,
// and the product fails loudly when the HIP library is missing.
//
// Supported subset (what the kernels use): __global__/__device__/__shared__, threadIdx /
// blockIdx / blockDim / gridDim (.x), __syncthreads, __ballot, __shfl/__shfl_xor/__shfl_down/
// __shfl_up (int, unsigned, long long, double), __popcll, __ffsll, __clzll, atomicOr/atomicAnd/atomicAdd, __hip_atomic_load/store, __threadfence
// on shared ints.  Cross-lane operations are rendezvous of an aligned G-lane group (G = 64
// by default; the path stage runs several frames per wavefront with G = 16): all lanes of a group must reach them
// (group-uniform control flow) — the same discipline the kernels follow on hardware; groups may diverge.
#pragma once
#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define FSDP_EMU 1

namespace emu {

struct Dim3 {
  unsigned x = 1, y = 1, z = 1;
};

constexpr int WAVE = 64;
constexpr size_t STACK_BYTES = 512 * 1024;

struct Lane {
  ucontext_t ctx;
  char* stack = nullptr;
  bool done = false;
  Dim3 tid;
};

struct Block {
  ucontext_t main_ctx;
  Lane lanes[WAVE];
  int n_lanes = WAVE;
  int cur = 0;
  int live = 0;
  // rendezvous state per group size (index log2 G) and group: groups of different sizes may be in flight at once
  // (a quad exchange inside a 16-lane group), so they must not share counters
  int garrived[7][WAVE] = {{0}};
  unsigned ggen[7][WAVE] = {{0}};
  Dim3 bid, bdim, gdim;
  uint64_t slots[WAVE];
  int line[WAVE] = {0};  // source line of each lane's last __syncthreads (dead-lock report)
  std::function<void()> body;
};

inline thread_local Block* B = nullptr;

inline void yield_lane() {
  Block* b = B;
  int from = b->cur;
  int nxt = from;
  for (int step = 0; step < b->n_lanes; step++) {
    nxt = (nxt + 1) % b->n_lanes;
    if (!b->lanes[nxt].done) break;
  }
  if (nxt == from) return;
  b->cur = nxt;
  swapcontext(&b->lanes[from].ctx, &b->lanes[nxt].ctx);
}

// Rendezvous of the G-lane group the current lane belongs to (G = 64: the whole block).  Groups are aligned
// (first lane = cur & ~(G-1)); a group's lanes follow the same control flow, different groups may diverge.
inline int group_live(Block* b, int first, int G) {
  int n = 0;
  for (int l = first; l < first + G && l < b->n_lanes; l++)
    if (!b->lanes[l].done) n++;
  return n;
}
inline void gbarrier(int G) {
  Block* b = B;
  int first = b->cur & ~(G - 1);
  const int lg = __builtin_ctz((unsigned)G);
  unsigned g = b->ggen[lg][first];
  if (++b->garrived[lg][first] >= group_live(b, first, G)) {
    b->garrived[lg][first] = 0;
    b->ggen[lg][first]++;
    return;
  }
  // a rendezvous that never completes = lanes of one group took different paths to different rendezvous points (or left):
  // report instead of spinning forever
  for (unsigned long spins = 0; b->ggen[lg][first] == g; spins++) {
    if (spins > 20000000ul) {
      fprintf(stderr, "emu: dead-locked rendezvous of a %d-lane group (block %u); lane: last __syncthreads line / done\n", G, b->bid.x);
      for (int l = 0; l < b->n_lanes; l++) fprintf(stderr, " %d:%d%s", l, b->line[l], b->lanes[l].done ? "/done" : "");
      fprintf(stderr, "\n");
      abort();
    }
    yield_lane();
  }
}
inline void barrier() { gbarrier(WAVE); }
inline void barrier_at(int line) {
  B->line[B->cur] = line;
  gbarrier(WAVE);
}

inline void lane_entry() {
  Block* b = B;
  b->body();
  b->lanes[b->cur].done = true;
  b->live--;
  // a lane that leaves while others of its group wait at a rendezvous would dead-lock them: the kernels never do
  // that (all lanes of a group reach the end together), so just hand over.
  if (b->live > 0) yield_lane();
  swapcontext(&b->lanes[b->cur].ctx, &b->main_ctx);
}

template <class F>
void launch(unsigned grid, unsigned block, F&& f) {
  static thread_local Block* blk = nullptr;
  if (!blk) {
    blk = new Block();
    for (int i = 0; i < WAVE; i++) blk->lanes[i].stack = (char*)malloc(STACK_BYTES);
  }
  if (block > (unsigned)WAVE) {
    fprintf(stderr, "emu: block size %u > 64 unsupported\n", block);
    abort();
  }
  for (unsigned bx = 0; bx < grid; bx++) {
    Block* b = blk;
    B = b;
    b->n_lanes = (int)block;
    b->live = (int)block;
    for (int q = 0; q < 7; q++)
      for (int i = 0; i < WAVE; i++) {
        b->garrived[q][i] = 0;
        b->ggen[q][i] = 0;
      }
    b->bid.x = bx;
    b->bdim.x = block;
    b->gdim.x = grid;
    b->body = f;
    for (unsigned l = 0; l < block; l++) {
      Lane& L = b->lanes[l];
      L.done = false;
      L.tid.x = l;
      getcontext(&L.ctx);
      L.ctx.uc_stack.ss_sp = L.stack;
      L.ctx.uc_stack.ss_size = STACK_BYTES;
      L.ctx.uc_link = nullptr;
      makecontext(&L.ctx, (void (*)())lane_entry, 0);
    }
    b->cur = 0;
    swapcontext(&b->main_ctx, &b->lanes[0].ctx);
    // returns here when a lane finished and found no live lane... make sure all are done
    while (b->live > 0) {
      int nxt = -1;
      for (int l = 0; l < b->n_lanes; l++)
        if (!b->lanes[l].done) {
          nxt = l;
          break;
        }
      if (nxt < 0) break;
      b->cur = nxt;
      swapcontext(&b->main_ctx, &b->lanes[nxt].ctx);
    }
  }
}

template <class T>
inline uint64_t to_bits(T v) {
  uint64_t u = 0;
  static_assert(sizeof(T) <= 8, "shfl payload too large");
  memcpy(&u, &v, sizeof(T));
  return u;
}
template <class T>
inline T from_bits(uint64_t u) {
  T v;
  memcpy(&v, &u, sizeof(T));
  return v;
}

// src = absolute lane index inside the block (must belong to the caller's group)
template <class T>
inline T gexchange(T v, int src, int G) {
  Block* b = B;
  int me = b->cur;
  b->slots[me] = to_bits(v);
  gbarrier(G);
  int first = me & ~(G - 1);
  T r = (src >= first && src < first + G && src < b->n_lanes) ? from_bits<T>(b->slots[src]) : v;
  gbarrier(G);
  return r;
}
template <class T>
inline T exchange(T v, int src) {
  return gexchange(v, src, WAVE);
}

// exchange of an arbitrary POD (<= 64 bytes) in one rendezvous
struct BigSlots {
  unsigned char b[WAVE][64];
};
inline thread_local BigSlots g_big;
template <class T>
inline T gexchange_struct(const T& v, int src, int G) {
  static_assert(sizeof(T) <= 64, "struct too large");
  Block* b = B;
  int me = b->cur;
  memcpy(g_big.b[me], &v, sizeof(T));
  gbarrier(G);
  int first = me & ~(G - 1);
  T r = v;
  if (src >= first && src < first + G && src < b->n_lanes) memcpy(&r, g_big.b[src], sizeof(T));
  gbarrier(G);
  return r;
}
template <class T>
inline T exchange_struct(const T& v, int src) {
  return gexchange_struct(v, src, WAVE);
}

// ballot over the caller's group; bit i = lane (first + i)
inline unsigned long long gballot(int pred, int G) {
  Block* b = B;
  int me = b->cur;
  b->slots[me] = pred ? 1 : 0;
  gbarrier(G);
  int first = me & ~(G - 1);
  unsigned long long m = 0;
  for (int l = first; l < first + G && l < b->n_lanes; l++)
    if (!b->lanes[l].done && b->slots[l]) m |= (1ull << (l - first));
  gbarrier(G);
  return m;
}
inline unsigned long long ballot(int pred) { return gballot(pred, WAVE); }

}  // namespace emu

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__
#define __restrict__
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define threadIdx (emu::B->lanes[emu::B->cur].tid)
#define blockIdx (emu::B->bid)
#define blockDim (emu::B->bdim)
#define gridDim (emu::B->gdim)

using std::isfinite;
using std::isinf;
using std::isnan;

#define __syncthreads() emu::barrier_at(__LINE__)
inline unsigned long long __ballot(int p) { return emu::ballot(p); }
template <class T>
inline T __shfl(T v, int src, int width = 64) {
  (void)width;
  return emu::exchange(v, src);
}
template <class T>
inline T __shfl_xor(T v, int mask, int width = 64) {
  (void)width;
  return emu::exchange(v, emu::B->cur ^ mask);
}
template <class T>
inline T __shfl_down(T v, unsigned d, int width = 64) {
  (void)width;
  int s = emu::B->cur + (int)d;
  return emu::exchange(v, s < emu::WAVE ? s : emu::B->cur);
}
template <class T>
inline T __shfl_up(T v, unsigned d, int width = 64) {
  (void)width;
  int s = emu::B->cur - (int)d;
  return emu::exchange(v, s >= 0 ? s : emu::B->cur);
}
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
inline int __clzll(unsigned long long v) { return v ? __builtin_clzll(v) : 64; }
template <class T>
inline T atomicOr(T* p, T v) {
  T o = *p;
  *p = o | v;
  return o;
}
template <class T>
inline T atomicAnd(T* p, T v) {
  T o = *p;
  *p = o & v;
  return o;
}
template <class T>
inline T atomicAdd(T* p, T v) {
  T o = *p;
  *p = o + v;
  return o;
}
// scoped atomics / fences of kernels that hand data from one workgroup to another (csrc/skidpad_kernel.h): the emulator
// runs the workgroups of a launch one after the other, in blockIdx order
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
inline void __threadfence() {}
#define __builtin_amdgcn_fence(order, scope) ((void)0)
inline void __builtin_amdgcn_s_waitcnt(int) {}
inline void __builtin_amdgcn_s_sleep(int) {}
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }  // (only ever applied to values all lanes hold)
