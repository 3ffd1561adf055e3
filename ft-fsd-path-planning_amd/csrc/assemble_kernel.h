// Last kernel of a pass: the three stage records of every frame -> one fsdp_frame_result (include/fsdp.h) in HBM, laid out
// exactly as the C ABI hands it to the caller, so the result leaves the GPU with ONE device-to-host copy straight into the
// caller's (pinned) buffer and the host never touches a frame (it used to rebuild 2.4 KB per frame from three copies: at
// 5 M frames/s that is 12 GB/s of host memcpy on one thread).  Pure data movement: consecutive lanes on consecutive 4-byte
// words of the destination, every source run contiguous (the records are written once by the stage
// kernels and read once here: 2.4 KB in, 2.4 KB out per frame, ~20 MB per 4096-frame pass).  One wavefront per frame.
//
// It also closes the pass's bookkeeping on the device: the lengths of the two hand-off lists (frames beyond the sorting
// kernel's LDS capacities; frames for the exact path kernel) go to the slot's host-visible trailer and the counters are
// reset for the slot's next pass — the host learns from the trailer whether the route kernels it did (not) launch were
// needed (fsdp_lib.hip verify_pass), and no hipMemsetAsync sits between the kernels of a pass any more.
#pragma once

#include <stddef.h>

#include "../../include/fsdp.h"
#include "fsdp_device.h"

namespace fsdp {

struct PassTrailer {
  int32_t n_big;    // frames sort_kernel appended to the big list in this pass
  int32_t n_retry;  // frames the path stage's fast kernels appended to the retry list
  int32_t seq;      // pass counter of the slot (written last)
  int32_t pad;
};

constexpr int RESULT_WORDS = (int)(sizeof(fsdp_frame_result) / 4);
static_assert(sizeof(fsdp_frame_result) % 4 == 0, "result words");

// word w of a frame's result <- which record, which word (all fields are 4- or 8-byte scalars at 4-byte granularity)
#define FSDP_RW(field) ((int)(offsetof(fsdp_frame_result, field) / 4))
#define FSDP_SW(field) ((int)(offsetof(SortOut, field) / 4))
#define FSDP_MW(field) ((int)(offsetof(MatchOut, field) / 4))
#define FSDP_PW(field) ((int)(offsetof(PathOut, field) / 4))

// skid != 0: a skidpad step — no sorting / matching records: those fields read as the reference's empty intermediates
// (counts 0, indices -1), status and path come from the path record.
__global__ void __launch_bounds__(256) assemble_kernel(int n_frames, const SortOut* __restrict__ sorted, const MatchOut* __restrict__ matched,
                                                        const PathOut* __restrict__ paths, fsdp_frame_result* __restrict__ results,
                                                        int* __restrict__ big, int* __restrict__ retry, PassTrailer* __restrict__ trailer,
                                                        int seq, const int32_t* __restrict__ extra_src = nullptr, int32_t* __restrict__ extra_dst = nullptr,
                                                        int extra_words = 0, const int32_t* __restrict__ remap = nullptr,
                                                        const int32_t* __restrict__ remap_off = nullptr) {
  // One wavefront per frame and trip: the result is five contiguous runs of its stage records (the static_asserts below
  // pin the layouts), each copied by consecutive lanes on consecutive 4-byte words — a handful of instructions per 64 words
  // (a per-word lookup of "which record, which word" was 636 wave-instructions per frame, this is ~120).
  const int32_t* s32 = (const int32_t*)sorted;
  const int32_t* m32 = (const int32_t*)matched;
  const int32_t* p32 = (const int32_t*)paths;
  int32_t* r32 = (int32_t*)results;
  constexpr int SW = (int)(sizeof(SortOut) / 4), MW = (int)(sizeof(MatchOut) / 4), PW = (int)(sizeof(PathOut) / 4);
  const int lane = (int)(threadIdx.x & 63);
  const int waves_per_block = (int)(blockDim.x >> 6);
  const long long wave0 = (long long)blockIdx.x * waves_per_block + (threadIdx.x >> 6);
  const long long n_waves = (long long)gridDim.x * waves_per_block;
  for (long long f = wave0; f < n_frames; f += n_waves) {
    const int32_t* s = s32 ? s32 + (size_t)f * SW : nullptr;
    const int32_t* m = m32 ? m32 + (size_t)f * MW : nullptr;
    const int32_t* p = p32 + (size_t)f * PW;
    int32_t* r = r32 + (size_t)f * RESULT_WORDS;
    // use_unknown_cones = False: indices of the compacted frame -> the caller's.  Only indices that ARE indices of that frame: a frame
    // the sorting kernel handed to sort_big_kernel in a pass that does not carry it (the pass is then repeated) leaves its record's
    // index fields unwritten, and mapping such a word read past the map (a GPU memory fault, found by round 6's compact-record tests)
    const int32_t n_map = remap ? remap_off[f + 1] - remap_off[f] : 0;
    // status: the latest stage that reported something decides (fsdp_lib.hip assemble())
    if (lane == 0) {
      int st = s ? s[FSDP_SW(status)] : 0;
      if (m && m[FSDP_MW(status)] != 0) st = m[FSDP_MW(status)];
      if (p[FSDP_PW(status)] != 0) st = p[FSDP_PW(status)];
      r[FSDP_RW(status)] = st;
      r[FSDP_RW(n_left_v)] = m ? m[FSDP_MW(n_left_v)] : 0;
      r[FSDP_RW(n_right_v)] = m ? m[FSDP_MW(n_right_v)] : 0;
      for (int w = FSDP_RW(n_right_v) + 1; w < FSDP_RW(left_v); w++) r[w] = 0;  // alignment padding
      r[FSDP_RW(path_fallback)] = p[FSDP_PW(fallback)];
      r[FSDP_RW(n_dense)] = p[FSDP_PW(n_dense)];
    }
    // n_left, n_right, left_idx, right_idx
    for (int w = FSDP_RW(n_left) + lane; w < FSDP_RW(n_left_v); w += 64) {
      int32_t v = s ? s[FSDP_SW(n_left) + (w - FSDP_RW(n_left))] : ((w >= FSDP_RW(left_idx)) ? -1 : 0);
      // (use_unknown_cones = False, filter_kernel.h: the sorter saw a compacted frame; indices go out in the caller's index space)
      if (remap && w >= FSDP_RW(left_idx) && v >= 0 && v < n_map) v = remap[remap_off[f] + v];
      r[w] = v;
    }
    // left_v | right_v | l2r | r2l
    for (int w = FSDP_RW(left_v) + lane; w < FSDP_RW(path); w += 64)
      r[w] = m ? m[FSDP_MW(left_v) + (w - FSDP_RW(left_v))] : ((w >= FSDP_RW(l2r)) ? -1 : 0);
    // path
    for (int w = FSDP_RW(path) + lane; w < FSDP_RW(n_configs_left); w += 64) r[w] = p[FSDP_PW(path) + (w - FSDP_RW(path))];
    // n_configs_left / right, first_k_left / right, best_cost_left / right
    for (int w = FSDP_RW(n_configs_left) + lane; w < FSDP_RW(path_fallback); w += 64) {
      int32_t v = 0;
      if (s) v = (w < FSDP_RW(best_cost_left)) ? s[FSDP_SW(n_configs_left) + (w - FSDP_RW(n_configs_left))] : s[FSDP_SW(best_cost_left) + (w - FSDP_RW(best_cost_left))];
      if (remap && w >= FSDP_RW(first_k_left) && w < FSDP_RW(best_cost_left) && v >= 0 && v < n_map) v = remap[remap_off[f] + v];
      r[w] = v;
    }
  }
  // a second, plain block of words on the same trip (skidpad steps: the planners' SkidInfo records to the caller's side)
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < extra_words; idx += (long long)gridDim.x * blockDim.x)
    extra_dst[idx] = extra_src[idx];
  if (blockIdx.x == 0 && threadIdx.x == 0 && trailer != nullptr) {
    // (the stage kernels finished before this kernel started: same stream)
    const int nb = big ? big[0] : 0, nr = retry ? retry[0] : 0;
    if (big) big[0] = 0;
    if (retry) retry[0] = 0;
    __hip_atomic_store(&trailer->n_big, nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&trailer->n_retry, nr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&trailer->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// The same trip with COMPACT records (include/fsdp.h fsdp_compact_result: path | sorted indices | status | counts — what a
// caller that only drives the car reads): 1384 instead of 2408 bytes per frame over PCIe.  One wavefront per frame; the
// path is one contiguous run of the path record, the indices one run of the sorting record.
constexpr int COMPACT_WORDS = (int)(sizeof(fsdp_compact_result) / 4);
static_assert(sizeof(fsdp_compact_result) % 8 == 0 && offsetof(fsdp_compact_result, left_idx) == sizeof(double) * 4 * PATH_POINTS &&
                  offsetof(fsdp_compact_result, right_idx) == offsetof(fsdp_compact_result, left_idx) + 4 * MAX_LEN &&
                  offsetof(fsdp_compact_result, status) == offsetof(fsdp_compact_result, right_idx) + 4 * MAX_LEN &&
                  sizeof(fsdp_compact_result) == offsetof(fsdp_compact_result, status) + 8,
              "compact record layout");
__global__ void __launch_bounds__(256) assemble_compact_kernel(int n_frames, const SortOut* __restrict__ sorted, const MatchOut* __restrict__ matched,
                                                                const PathOut* __restrict__ paths, fsdp_compact_result* __restrict__ results,
                                                                int* __restrict__ big, int* __restrict__ retry, PassTrailer* __restrict__ trailer, int seq,
                                                                const int32_t* __restrict__ remap, const int32_t* __restrict__ remap_off) {
  const int32_t* s32 = (const int32_t*)sorted;
  const int32_t* m32 = (const int32_t*)matched;
  const int32_t* p32 = (const int32_t*)paths;
  int32_t* r32 = (int32_t*)results;
  constexpr int SW = (int)(sizeof(SortOut) / 4), MW = (int)(sizeof(MatchOut) / 4), PW = (int)(sizeof(PathOut) / 4);
  constexpr int IDX0 = (int)(offsetof(fsdp_compact_result, left_idx) / 4), ST0 = (int)(offsetof(fsdp_compact_result, status) / 4);
  const int lane = (int)(threadIdx.x & 63);
  const int waves_per_block = (int)(blockDim.x >> 6);
  const long long wave0 = (long long)blockIdx.x * waves_per_block + (threadIdx.x >> 6);
  const long long n_waves = (long long)gridDim.x * waves_per_block;
  for (long long f = wave0; f < n_frames; f += n_waves) {
    const int32_t* s = s32 + (size_t)f * SW;
    const int32_t* m = m32 + (size_t)f * MW;
    const int32_t* p = p32 + (size_t)f * PW;
    int32_t* r = r32 + (size_t)f * COMPACT_WORDS;
    for (int w = lane; w < IDX0; w += 64) r[w] = p[FSDP_PW(path) + w];
    if (lane < 2 * MAX_LEN) {  // left_idx | right_idx are one run of the sorting record
      int32_t v = s[FSDP_SW(left_idx) + lane];
      if (remap && v >= 0 && v < remap_off[f + 1] - remap_off[f]) v = remap[remap_off[f] + v];  // (see assemble_kernel)
      r[IDX0 + lane] = v;
    }
    if (lane == 0) {
      int st = s[FSDP_SW(status)];
      if (m[FSDP_MW(status)] != 0) st = m[FSDP_MW(status)];
      if (p[FSDP_PW(status)] != 0) st = p[FSDP_PW(status)];
      r[ST0] = st;
      // n_left | n_right | path_fallback | n_dense, one byte each (little endian: the struct's field order)
      r[ST0 + 1] = (s[FSDP_SW(n_left)] & 255) | ((s[FSDP_SW(n_right)] & 255) << 8) | ((p[FSDP_PW(fallback)] & 255) << 16) | ((p[FSDP_PW(n_dense)] & 255) << 24);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && trailer != nullptr) {
    const int nb = big ? big[0] : 0, nr = retry ? retry[0] : 0;
    if (big) big[0] = 0;
    if (retry) retry[0] = 0;
    __hip_atomic_store(&trailer->n_big, nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&trailer->n_retry, nr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&trailer->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Host -> device of a batch whose buffers are page-locked, as a kernel: the batch comes over PCIe by plain loads from host
// memory (the GPU maps page-locked host memory into its address space) into the slot's device buffers — up to four
// segments (offsets, cones, poses, previous paths) in one launch.  Used by the skidpad steps and by contexts that filter
// the cones first; an ordinary ticket's batch is read by its sorting kernel itself (sort_kernel.h StageIn: +6 % frames/s
// over this kernel in front of the pass).  Why a kernel and not hipMemcpyAsync: the
// copies of the SDMA engines are ordered against the kernels of a stream through cross-engine signals, and with ten
// streams each alternating copy / kernels / copy the runtime's submission stalled for milliseconds at a time (measured:
// profiles/r03_streaming.txt); as kernels, a batch's transfers are ordinary packets of its own stream.  16 bytes per lane
// and load, four loads in flight per lane: 256 workgroups keep ~1 MB on the wire, enough for the link's latency x bandwidth.
struct CopySeg {
  const void* src;
  void* dst;
  unsigned long long bytes;  // multiple of 4
};
struct CopySegs {
  CopySeg seg[4];
  int n;
  int32_t rebase;  // subtracted from every 4-byte word of seg[0] (the CSR offsets of a slice of a larger batch: the device copy starts at 0)
};
__global__ void __launch_bounds__(256) stage_in_kernel(CopySegs S) {
  const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long nth = (unsigned long long)gridDim.x * blockDim.x;
  for (int k = 0; k < S.n; k++) {
    const char* src = (const char*)S.seg[k].src;
    char* dst = (char*)S.seg[k].dst;
    const unsigned long long bytes = S.seg[k].bytes;
    if (k == 0 && S.rebase != 0) {
      for (unsigned long long w = tid; w < bytes / 4; w += nth) ((int32_t*)dst)[w] = ((const int32_t*)src)[w] - S.rebase;
    } else if ((((unsigned long long)src | (unsigned long long)dst) & 15ull) == 0) {
      const unsigned long long n16 = bytes / 16;
      typedef unsigned int u4 __attribute__((ext_vector_type(4)));
      const u4* s4 = (const u4*)src;
      u4* d4 = (u4*)dst;
      unsigned long long i = tid;
      for (; i + 3 * nth < n16; i += 4 * nth) {
        const u4 a = __builtin_nontemporal_load(s4 + i), b = __builtin_nontemporal_load(s4 + i + nth),
                    c = __builtin_nontemporal_load(s4 + i + 2 * nth), d = __builtin_nontemporal_load(s4 + i + 3 * nth);
        d4[i] = a;
        d4[i + nth] = b;
        d4[i + 2 * nth] = c;
        d4[i + 3 * nth] = d;
      }
      for (; i < n16; i += nth) d4[i] = __builtin_nontemporal_load(s4 + i);
      for (unsigned long long w = n16 * 4 + tid; w < bytes / 4; w += nth) ((uint32_t*)dst)[w] = ((const uint32_t*)src)[w];
    } else {
      for (unsigned long long w = tid; w < bytes / 4; w += nth) ((uint32_t*)dst)[w] = ((const uint32_t*)src)[w];
    }
  }
}

// layout facts the word map above relies on
static_assert(offsetof(fsdp_frame_result, right_idx) - offsetof(fsdp_frame_result, n_left) == offsetof(SortOut, right_idx) - offsetof(SortOut, n_left), "sort run");
static_assert(offsetof(fsdp_frame_result, n_left_v) == offsetof(fsdp_frame_result, right_idx) + sizeof(int32_t) * MAX_LEN, "sort run end");
static_assert(offsetof(fsdp_frame_result, r2l) - offsetof(fsdp_frame_result, left_v) == offsetof(MatchOut, r2l) - offsetof(MatchOut, left_v), "match run");
static_assert(offsetof(fsdp_frame_result, path) == offsetof(fsdp_frame_result, r2l) + sizeof(int32_t) * MAX_MATCH, "match run end");
static_assert(offsetof(fsdp_frame_result, n_configs_left) == offsetof(fsdp_frame_result, path) + sizeof(double) * 4 * PATH_POINTS, "path run end");
static_assert(offsetof(fsdp_frame_result, first_k_right) - offsetof(fsdp_frame_result, n_configs_left) ==
                  offsetof(SortOut, first_k_right) - offsetof(SortOut, n_configs_left), "diagnostics run");
static_assert(offsetof(fsdp_frame_result, best_cost_left) == offsetof(fsdp_frame_result, first_k_right) + 8, "diagnostics run end");
static_assert(offsetof(fsdp_frame_result, best_cost_right) - offsetof(fsdp_frame_result, best_cost_left) ==
                  offsetof(SortOut, best_cost_right) - offsetof(SortOut, best_cost_left), "cost run");
static_assert(offsetof(fsdp_frame_result, path_fallback) == offsetof(fsdp_frame_result, best_cost_right) + 8, "cost run end");
static_assert(offsetof(fsdp_frame_result, n_dense) == offsetof(fsdp_frame_result, path_fallback) + 4 && sizeof(fsdp_frame_result) == offsetof(fsdp_frame_result, n_dense) + 4, "tail");

}  // namespace fsdp
