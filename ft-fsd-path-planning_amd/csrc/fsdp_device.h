// Device-side common definitions for the MI355X (gfx950) PathPlanner kernels.
//
// Execution model: one wavefront per workgroup; a frame owns the whole wavefront (sorting, matching) or a lane
// group of it (path stage: Grp<G> below, up to eight frames per wavefront).  Frame state lives in LDS; group-uniform
// control flow drives the inherently serial parts (DFS stack, Givens QR of the smoothing spline) while the
// data-parallel parts (pairwise distances, kNN, side counting, B-spline evaluation, curvature windows) run one
// element per lane.  Cross-lane traffic uses ballot / shuffles; LDS hand-offs between lanes are fenced (a single-wave
// workgroup: s_barrier is a no-op in hardware, the fence orders LDS for the compiler).
//
// Arithmetic contract: float64 throughout, compiled with -ffp-contract=off so every + - * /
// sqrt is the IEEE operation the reference performs; fused multiply-adds appear only where
// the reference's NumPy->BLAS calls fuse them (explicit fma(), see blas_dot2).
#pragma once

#include <stdint.h>

#ifndef FSDP_EMU
#include <hip/hip_runtime.h>
#endif

namespace fsdp {

constexpr int WAVE = 64;
constexpr int MAX_CONES = 256;   // cones per frame handled in LDS (status OVERFLOW beyond)
// Structural capacities.  Two builds of these sources exist (include/fsdp.h): the standard shapes are the reference's
// defaults; -DFSDP_WIDE_SHAPES compiles the wide library for contexts whose parameters exceed them.
#ifdef FSDP_WIDE_SHAPES
constexpr int MAX_LEN = 16;      // reference config.py:36 max_length
constexpr int KNN = 8;           // config.py:34 max_n_neighbors
#else
constexpr int MAX_LEN = 12;
constexpr int KNN = 5;
#endif
constexpr int MAX_ENDS = 64;     // raw end configurations kept per side (OVERFLOW beyond)
constexpr int MAX_STACK = (KNN * MAX_LEN + 63) / 64 * 64;  // DFS stack bound: <= KNN pending siblings per depth * MAX_LEN depths
constexpr int MAX_MATCH = 2 * MAX_LEN;  // cones-with-virtual per side: a side's own (<= MAX_LEN) + inserted (<= MAX_LEN)
#ifdef FSDP_WIDE_SHAPES
constexpr int PATH_POINTS = 64;  // config.py:58 mpc_prediction_horizon (rows of every path array)
#else
constexpr int PATH_POINTS = 40;
#endif
constexpr int CHORD_POINTS = 40; // path_calculator_helpers.py:54-68: the almost-straight initial path has 40 points whatever the horizon

constexpr int T_UNKNOWN = 0, T_RIGHT = 1, T_LEFT = 2;

// per-frame status codes (mirrored in include/fsdp.h)
constexpr int ST_OK = 0;
constexpr int ST_REF_UNDEFINED_SET_DIFF = 101;
constexpr int ST_REF_UNDEFINED_DFS_OOB = 102;
constexpr int ST_REF_UNDEFINED_PATH = 103;
constexpr int ST_REF_UNDEFINED_MATCH_IDX = 104;
constexpr int ST_OVERFLOW_CONES = 201;
constexpr int ST_OVERFLOW_ENDS = 202;
constexpr int ST_OVERFLOW_PATH = 203;
constexpr int ST_OVERFLOW_KNOTS = 204;
constexpr int ST_RETRY = 299;  // internal: the fast kernels hand the frame to the exact one-frame-per-wavefront kernel (never leaves the library)

// The reference's configuration constants (fsd_path_planning/config.py:33-41,48,55-59,124-129), one device copy per
// context (fsdp_create): the kernels read them with scalar loads.  Structural ones are bounded by the compiled capacities
// (max_n_neighbors <= KNN, max_length <= MAX_LEN, mpc_prediction_horizon <= PATH_POINTS).
struct Params {
  // sorting_cones (config.py:33-41)
  int32_t max_n_neighbors, max_length;
  double max_dist, max_dist_to_first, threshold_directional_angle, threshold_absolute_angle;
  // cone_matching (config.py:124-129)
  double min_track_width, max_search_range, max_search_angle;
  // calculate_path (config.py:48,55-59)
  double smoothing, predict_every, maximal_distance_for_valid_path, mpc_path_length;
  // the ones that select a branch rather than a threshold
  int32_t max_deg;                      // config.py:48: degree of fits #1 / #2 is clip(points - 1, 1, max_deg) (utils/spline_fit.py:113)
  int32_t horizon;                      // config.py:58 mpc_prediction_horizon: rows of a path (<= PATH_POINTS, the stride of every path array)
  int32_t matches_should_be_monotonic;  // config.py:124-146, functional_cone_matching.py:164-171
  int32_t use_unknown_cones;            // config.py:40, core_cone_sorting.py:114 (0: cones of type UNKNOWN are dropped before sorting)
  // Optional side output of the path stage: the second value CalculatePath.run_path_calculation returns
  // (core_calculate_path.py:575 center_along_match_connection).  Null except during fsdp_path_batch_centers.
  int32_t centers_cap;   // points per frame the buffer holds
  double* centers;       // (n_frames, centers_cap, 2)
  int32_t* n_centers;    // (n_frames,) points the reference's array holds (may exceed centers_cap: only the first cap are stored)
  // path_retry_kernel: list length above which the frames of the exact route share wavefronts (four per wavefront); up to it a
  // frame has a wavefront to itself (path_kernel.h).  512 in the library (FSDP_RETRY_PACK_MIN); 0 in the host emulator's
  // parameter block, so that the CPU tests run the shared form.
  int32_t retry_pack_min;
};

#define FSDP_PI 3.14159265358979323846
#define FSDP_DEG (FSDP_PI / 180.0)

// ---- sorting stage output (one record per frame, HBM) ----
struct SortOut {
  int32_t status;
  int32_t n_left, n_right;
  int32_t left_idx[MAX_LEN];
  int32_t right_idx[MAX_LEN];
  int32_t n_configs_left, n_configs_right;
  int32_t first_k_left[2], first_k_right[2];
  double best_cost_left, best_cost_right;
};

// ---- matching stage output ----
struct MatchOut {
  int32_t status;
  int32_t n_left_v, n_right_v;
  int32_t pad;
  double left_v[MAX_MATCH][2];
  double right_v[MAX_MATCH][2];
  int32_t l2r[MAX_MATCH];
  int32_t r2l[MAX_MATCH];
};

// ---- path stage output ----
struct PathOut {
  double path[PATH_POINTS][4];
  int32_t status;
  int32_t fallback;
  int32_t n_dense;   // number of dense samples L the 40 outputs were drawn from
  int32_t pad;
};

// ------------------------------------------------------------------------------------------
// scalar math helpers (NumPy semantics; see DESIGN.md "arithmetic contract")
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double blas_dot2(double a0, double b0, double a1, double b1) {
  // np.dot / BLAS inner product of length 2 as OpenBLAS evaluates it: acc = a0*b0; acc = fma(a1,b1,acc)
  return fma(a1, b1, a0 * b0);
}
// np.dot of ONE point (a 1-D vector) with a 2 x 2 matrix goes to gemv, whose OpenBLAS kernel forms the products in the other
// order; from two rows on it is gemm, i.e. blas_dot2 (oracle/np_compat.h blas_dot2_single_row; pinned against NumPy by
// tests/test_oracle_numpy_semantics.py).  The skidpad mission rotates the car's position as a single point.
__device__ __forceinline__ double blas_dot2_single_row(double a0, double b0, double a1, double b1) { return fma(a0, b0, a1 * b1); }
__device__ __forceinline__ double norm_blas(double x, double y) { return sqrt(blas_dot2(x, x, y, y)); }
__device__ __forceinline__ double norm_axis(double x, double y) { return sqrt(x * x + y * y); }

// reference utils/math_utils.py:120-150 (expansion-form squared distance through dgemm, K=6)
__device__ __forceinline__ double cdist_sq(double ax, double ay, double bx, double by) {
  double acc = 1.0 * (bx * bx);
  acc = fma(1.0, by * by, acc);
  acc = fma(ax, -2 * bx, acc);
  acc = fma(ay, -2 * by, acc);
  acc = fma(ax * ax, 1.0, acc);
  acc = fma(ay * ay, 1.0, acc);
  return acc;
}

// reference utils/math_utils.py:70-100: the clipped cosine, then its arc cosine
__device__ __forceinline__ double cos_between(double ax, double ay, double bx, double by) {
  double c = ax * bx + ay * by;
  c /= sqrt(ax * ax + ay * ay) * sqrt(bx * bx + by * by);
  if (c < -1) c = -1;
  if (c > 1) c = 1;
  return c;
}
__device__ __forceinline__ double angle_between(double ax, double ay, double bx, double by) { return acos(cos_between(ax, ay, bx, by)); }
// angle_between(...) < thr and > thr for a constant threshold, from the clipped cosine c: the arc cosine is monotone, so
// the cosine decides unless it lies within 1e-9 of cos(thr) (nine orders of magnitude above acos' rounding) — only then
// is the arc cosine itself evaluated and compared, like the reference does.  NaN (a zero vector) compares false everywhere.
// The slow path is a CALL (FSDP_ACOS_COLD, default on): inlined, every predicate carried its own copy of the device library's acos, and
// the polynomial's coefficients — shared by the copies, hoisted to the top of the sorting kernel — stayed alive across the whole kernel
// and were spilled to scratch (18 registers, re-read by eight dependent scratch loads inside every acos of the cost phase).
#ifndef FSDP_ACOS_COLD
#define FSDP_ACOS_COLD 1
#endif
#if FSDP_ACOS_COLD && !defined(FSDP_EMU)
__device__ __attribute__((noinline)) inline double acos_cold(double c) { return acos(c); }
#else
__device__ __forceinline__ double acos_cold(double c) { return acos(c); }
#endif
__device__ __forceinline__ bool acos_less(double c, double thr, double cos_thr) {
  if (c > cos_thr + 1e-9) return true;
  if (c < cos_thr - 1e-9) return false;
  return acos_cold(c) < thr;
}
__device__ __forceinline__ bool acos_greater(double c, double thr, double cos_thr) {
  if (c < cos_thr - 1e-9) return true;
  if (c > cos_thr + 1e-9) return false;
  return acos_cold(c) > thr;
}
constexpr double COS_60DEG = 0.5, COS_85DEG = 0.087155742747658173558, COS_150DEG = -0.86602540378443864676;

// reference utils/math_utils.py:663-676: (a1 - a2 + 3pi) % (2pi) - pi  (NumPy floored modulo).  For arguments in
// [0, 6 pi) — every difference of two atan2 / acos values — the floating-point remainder is x, x - 2pi or x - 4pi, each an
// exact subtraction (Sterbenz), i.e. the very value fmod returns; anything else takes fmod.
__device__ __forceinline__ double angle_difference(double a1, double a2) {
  const double x = a1 - a2 + 3 * FSDP_PI;
  const double T = 2 * FSDP_PI;
  double m;
  if (x >= 0 && x < T) {
    m = x;
  } else if (x >= T && x < 2 * T) {
    m = x - T;
  } else if (x >= 2 * T && x < 3 * T) {
    m = x - 2 * T;
  } else {
    m = fmod(x, T);
    if (m != 0.0 && m < 0) m += T;
  }
  return m - FSDP_PI;
}

__device__ __forceinline__ double sign_of(double v) { return (v > 0) ? 1.0 : ((v < 0) ? -1.0 : 0.0); }

// rotation by theta the way utils/math_utils.py:103-117 does it: points @ [[c, s], [-s, c]]
struct Rot2 {
  double c, s;
};
__device__ __forceinline__ Rot2 make_rot(double theta) { return Rot2{cos(theta), sin(theta)}; }
__device__ __forceinline__ void rot_apply(const Rot2& r, double x, double y, double& ox, double& oy) {
  ox = blas_dot2(x, r.c, y, -r.s);
  oy = blas_dot2(x, r.s, y, r.c);
}

// cone_matching/match_directions.py:7-20: rotate the chord by +-pi/2 and normalise.
// cos(+-pi/2) = 6.123233995736766e-17 and sin(+-pi/2) = +-1 are the libm values NumPy uses.
__device__ __forceinline__ void search_direction(double x0, double y0, double x1, double y1, int cone_type, double& dx,
                                                 double& dy) {
  const double c = 6.123233995736766e-17;
  const double s = (cone_type == T_RIGHT) ? 1.0 : -1.0;
  double tx = x1 - x0, ty = y1 - y0;
  double rx = blas_dot2(tx, c, ty, -s);
  double ry = blas_dot2(tx, s, ty, c);
  double n = norm_blas(rx, ry);
  dx = rx / n;
  dy = ry / n;
}

// NumPy pairwise summation for short runs (n <= 128): n < 8 sequential, else 8 accumulators
__device__ __forceinline__ double np_sum_small(const double* a, int n) {
  if (n < 8) {
    double r = 0.0;
    for (int i = 0; i < n; i++) r += a[i];
    return r;
  }
  double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
  int i = 8;
  for (; i < n - (n % 8); i += 8) {
    r0 += a[i];
    r1 += a[i + 1];
    r2 += a[i + 2];
    r3 += a[i + 3];
    r4 += a[i + 4];
    r5 += a[i + 5];
    r6 += a[i + 6];
    r7 += a[i + 7];
  }
  double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
  for (; i < n; i++) res += a[i];
  return 0.0 + res;
}

// ------------------------------------------------------------------------------------------
// optional per-section cycle accounting (tools/section_profile.py builds with -DFSDP_PROFILE)
// ------------------------------------------------------------------------------------------
#if defined(FSDP_PROFILE) && !defined(FSDP_EMU)
__device__ long long* g_prof = nullptr;  // [n_blocks][32] cycle sums / counters
// Sections accumulate in LDS (a scope costs two s_memtime and one LDS read-modify-write) and are written to g_prof once
// per wavefront by PROF_FLUSH(); accumulating in global memory would charge every scope a memory round trip.
__device__ __forceinline__ long long* prof_lds() {
  __shared__ long long acc[32];
  return acc;
}
struct ProfScope {
  long long t0;
  int slot;
  __device__ ProfScope(int s) : t0(clock64()), slot(s) {}
  __device__ ~ProfScope() {
    long long t1 = clock64();
    if (g_prof && (threadIdx.x & 63) == 0) prof_lds()[slot] += (t1 - t0);
  }
};
#define PROF(slot) ProfScope prof_scope_##slot(slot)
// (a section that is not a scope of its own: PROF_T0(slot) ... PROF_T1(slot))
#define PROF_T0(slot) long long prof_t0_##slot = clock64()
#define PROF_T1(slot)                                                                        \
  do {                                                                                       \
    long long prof_t1_ = clock64();                                                          \
    if (g_prof && (threadIdx.x & 63) == 0) prof_lds()[slot] += (prof_t1_ - prof_t0_##slot); \
  } while (0)
// per-frame counters of the profiling build: slot + (group index of the frame inside its wavefront), lane 0 of the group
#define PROF_COUNT(slot, G, value)                                                                              \
  do {                                                                                                          \
    if (g_prof && (threadIdx.x & ((G) - 1)) == 0 && (threadIdx.x & 63) / (G) < 4)                                    \
      prof_lds()[(slot) + (threadIdx.x & 63) / (G)] += (value);                                                  \
  } while (0)
#define PROF_INIT()                                           \
  do {                                                        \
    if ((threadIdx.x & 63) < 32) prof_lds()[threadIdx.x & 63] = 0; \
    __syncthreads();                                          \
  } while (0)
#define PROF_FLUSH()                                                                                  \
  do {                                                                                                \
    __syncthreads();                                                                                  \
    if (g_prof && (threadIdx.x & 63) < 32) g_prof[(size_t)blockIdx.x * 32 + (threadIdx.x & 63)] += prof_lds()[threadIdx.x & 63]; \
  } while (0)
#else
#define PROF(slot)
#define PROF_T0(slot)
#define PROF_T1(slot)
#define PROF_COUNT(slot, G, value)
#define PROF_INIT()
#define PROF_FLUSH()
#endif
// finer sections of the fit itself (profiling builds of the fit kernel with -DFSDP_PROFILE_FIT_DETAIL: slots 1-6, which
// belong to the path stage in the prep / finish profiles)
#if defined(FSDP_PROFILE_FIT_DETAIL)
#define PROFX_T0(slot) PROF_T0(slot)
#define PROFX_T1(slot) PROF_T1(slot)
#else
#define PROFX_T0(slot)
#define PROFX_T1(slot)
#endif
// which kernel of the three-kernel path stage the profiling build accounts: 1 fit_kernel (default), 2 prep, 3 finish
#ifndef FSDP_PROFILE_KERNEL
#define FSDP_PROFILE_KERNEL 1
#endif
#define PROF_INIT_K(k)                          \
  do {                                          \
    if constexpr (FSDP_PROFILE_KERNEL == (k)) { \
      PROF_INIT();                              \
    }                                           \
  } while (0)
#define PROF_FLUSH_K(k)                         \
  do {                                          \
    if constexpr (FSDP_PROFILE_KERNEL == (k)) { \
      PROF_FLUSH();                             \
    }                                           \
  } while (0)

// ------------------------------------------------------------------------------------------
// wave primitives
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

template <class T>
__device__ __forceinline__ T wave_bcast(T v, int src) {
  return __shfl(v, src, WAVE);
}

// a value every lane of the wavefront holds, moved to scalar registers (it stays out of the vector registers for as long as
// it lives)
__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double wave_uniform(double v) {
  union {
    double d;
    int i[2];
  } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
  u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
  return u.d;
}

// argmin over (value, index) pairs with "first smallest" semantics (lowest index on ties);
// lanes holding no candidate pass idx = -1.
__device__ __forceinline__ void wave_argmin(double& v, int& idx) {
  for (int off = 32; off >= 1; off >>= 1) {
    double ov = __shfl_xor(v, off, WAVE);
    int oi = __shfl_xor(idx, off, WAVE);
    bool take = (oi >= 0) && (idx < 0 || ov < v || (ov == v && oi < idx));
    if (take) {
      v = ov;
      idx = oi;
    }
  }
}

__device__ __forceinline__ int wave_min_int(int v) {
  for (int off = 32; off >= 1; off >>= 1) {
    int o = __shfl_xor(v, off, WAVE);
    v = (o < v) ? o : v;
  }
  return v;
}

// ------------------------------------------------------------------------------------------
// lane groups: G lanes per frame, WAVE / G frames per wavefront
// ------------------------------------------------------------------------------------------
// The sorting / matching kernels give a frame the whole wavefront (G = 64).  The path stage is dominated by
// serial FP64 chains (spline QR) that keep 1-4 lanes busy, so it packs WAVE / G frames into one wavefront
// (G = 16: four frames, one per DPP row): a serial instruction then advances four frames at once.  Groups are
// aligned; control flow is uniform WITHIN a group and may diverge BETWEEN groups (the hardware runs the union of
// the paths, masked).  Cross-lane traffic never leaves a group.  sync() orders LDS / scratch hand-offs between the
// lanes of a group: with one wavefront per workgroup the lanes run in lock-step, so only the compiler and the memory
// counters need a fence (no s_barrier, which must not sit in divergent code).
template <int G>
struct Grp {
  static_assert(G == 4 || G == 8 || G == 16 || G == 32 || G == 64, "group size");
  static constexpr int SIZE = G;
  static constexpr int PER_WAVE = WAVE / G;
  static __device__ __forceinline__ int lane() { return (int)(threadIdx.x & (G - 1)); }
  static __device__ __forceinline__ int index() { return (int)((threadIdx.x & 63) / G); }
  static __device__ __forceinline__ void sync() {
#ifdef FSDP_EMU
    emu::gbarrier(G);
#else
    if constexpr (G == WAVE) {
      __syncthreads();
    } else {
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      __builtin_amdgcn_wave_barrier();
    }
#endif
  }
  // bit i = lane i of this group
  static __device__ __forceinline__ unsigned long long ballot(bool p) {
#ifdef FSDP_EMU
    return emu::gballot(p, G);
#else
    unsigned long long m = __ballot(p);
    if constexpr (G == WAVE)
      return m;
    else
      return (m >> (index() * G)) & ((1ull << G) - 1ull);
#endif
  }
  // value of group lane `src` (group-uniform src)
  template <class T>
  static __device__ __forceinline__ T bcast(T v, int src) {
#ifdef FSDP_EMU
    return emu::gexchange(v, (emu::B->cur & ~(G - 1)) | src, G);
#else
    return __shfl(v, (int)((threadIdx.x & 63) & ~(G - 1)) | src, WAVE);
#endif
  }
  template <class T>
  static __device__ __forceinline__ T shfl_xor(T v, int mask) {
#ifdef FSDP_EMU
    return emu::gexchange(v, emu::B->cur ^ mask, G);
#else
    return __shfl_xor(v, mask, WAVE);
#endif
  }
  // value of the previous lane of the group (lane 0 keeps its own)
  template <class T>
  static __device__ __forceinline__ T shfl_up1(T v) {
#ifdef FSDP_EMU
    return emu::gexchange(v, lane() > 0 ? emu::B->cur - 1 : emu::B->cur, G);
#else
    int me = (int)(threadIdx.x & 63);
    return __shfl(v, lane() > 0 ? me - 1 : me, WAVE);
#endif
  }
  // value of the next lane of the group (the last lane keeps its own)
  template <class T>
  static __device__ __forceinline__ T shfl_down1(T v) {
#ifdef FSDP_EMU
    return emu::gexchange(v, lane() < G - 1 ? emu::B->cur + 1 : emu::B->cur, G);
#else
    int me = (int)(threadIdx.x & 63);
    return __shfl(v, lane() < G - 1 ? me + 1 : me, WAVE);
#endif
  }
  // argmin over (value, index) pairs with "first smallest" semantics; lanes holding no candidate pass idx = -1
  static __device__ __forceinline__ void argmin(double& v, int& idx) {
    for (int off = G / 2; off >= 1; off >>= 1) {
      double ov = shfl_xor(v, off);
      int oi = shfl_xor(idx, off);
      bool take = (oi >= 0) && (idx < 0 || ov < v || (ov == v && oi < idx));
      if (take) {
        v = ov;
        idx = oi;
      }
    }
  }
};

}  // namespace fsdp
