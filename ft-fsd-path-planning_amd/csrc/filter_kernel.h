// use_unknown_cones = False (config.py:40, sorting_cones/core_cone_sorting.py:113-115): the reference empties the UNKNOWN
// list before it flattens the cones, so the sorter never sees them.  Here: three small kernels in front of a pass, launched
// only for contexts created with that parameter — per frame the cones of a known type are compacted (in order) into a
// second cone buffer with its own CSR offsets, and a map compact index -> index in the caller's array lets assemble_kernel
// report the sorted indices in the caller's index space.  Every other kernel runs unchanged on the compacted batch.
#pragma once
#include "fsdp_device.h"

namespace fsdp {

// one wavefront per frame: number of cones whose type is not UNKNOWN
__global__ void __launch_bounds__(64) filter_count_kernel(int n_frames, const int32_t* __restrict__ off, const double* __restrict__ cones,
                                                         int32_t* __restrict__ counts) {
  const int f = blockIdx.x;
  if (f >= n_frames) return;
  const int lane = lane_id();
  const int lo = off[f], n = off[f + 1] - lo;
  int c = 0;
  for (int base = 0; base < n; base += WAVE) {
    const int i = base + lane;
    const bool keep = i < n && (int)cones[3 * (size_t)(lo + i) + 2] != T_UNKNOWN;
    c += __popcll(__ballot(keep));
  }
  if (lane == 0) counts[f] = c;
}

// one workgroup: exclusive prefix sum of the counts -> CSR offsets of the compacted batch (n_frames + 1 entries)
__global__ void __launch_bounds__(1024) filter_scan_kernel(int n_frames, const int32_t* __restrict__ counts, int32_t* __restrict__ new_off) {
  __shared__ int part[1024];
  const int t = threadIdx.x, T = blockDim.x;  // (T <= 1024)
  const int per = (n_frames + T - 1) / T;
  const int lo = t * per, hi = (lo + per < n_frames) ? lo + per : n_frames;
  int s = 0;
  for (int i = lo; i < hi; i++) s += counts[i];
  part[t] = s;
  __syncthreads();
  if (t == 0) {
    int acc = 0;
    for (int i = 0; i < T; i++) {
      const int v = part[i];
      part[i] = acc;
      acc += v;
    }
    new_off[n_frames] = acc;
  }
  __syncthreads();
  int acc = part[t];
  for (int i = lo; i < hi; i++) {
    new_off[i] = acc;
    acc += counts[i];
  }
}

// one wavefront per frame: rows of a known type, in order, to the compacted buffer; map[new index] = old index in the frame
__global__ void __launch_bounds__(64) filter_scatter_kernel(int n_frames, const int32_t* __restrict__ off, const double* __restrict__ cones,
                                                           const int32_t* __restrict__ new_off, double* __restrict__ new_cones,
                                                           int32_t* __restrict__ map) {
  const int f = blockIdx.x;
  if (f >= n_frames) return;
  const int lane = lane_id();
  const int lo = off[f], n = off[f + 1] - lo;
  const int dst0 = new_off[f];
  int done = 0;
  for (int base = 0; base < n; base += WAVE) {
    const int i = base + lane;
    double x = 0, y = 0, t = 0;
    bool keep = false;
    if (i < n) {
      const double* r = cones + 3 * (size_t)(lo + i);
      x = r[0];
      y = r[1];
      t = r[2];
      keep = (int)t != T_UNKNOWN;
    }
    const unsigned long long m = __ballot(keep);
    if (keep) {
      const int j = dst0 + done + __popcll(m & ((1ull << lane) - 1ull));
      double* w = new_cones + 3 * (size_t)j;
      w[0] = x;
      w[1] = y;
      w[2] = t;
      map[j] = i;
    }
    done += __popcll(m);
  }
}

}  // namespace fsdp
