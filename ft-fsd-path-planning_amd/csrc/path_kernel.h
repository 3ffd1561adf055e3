// calculate_path step as a HIP kernel with several frames per wavefront (gfx950).
//
// Replaces CalculatePath.run_path_calculation for independent frames (fresh-planner semantics: the
// "previous path" is the constant initial path), reference calculate_path/core_calculate_path.py:514-575:
//   centre points of the matches (:151-205)                      lane = cone
//   fit #1 + dense evaluation every 0.1 m (:207-223)              spline_device.h
//   too-far check, connect to car, circular/linear extension, trim behind the car (:225-334,:430-465)
//   fit #2 + 300-sample evaluation, cut at 20 m (:239-259,:467-499)
//   PathParameterizer.parameterize_path (path_parameterization.py:297-328): fit #3 (s = 0.01), dense
//   evaluation, circle-fit curvature over sliding windows (lane = window), uniform filter, 40-index resample.
// Every float that feeds the sample-count decision ceil(max_u / predict_every) is produced in the
// reference's rounding order (see DESIGN.md "arithmetic contract").
//
// Lanes: the stage is templated on the group size G (lanes per frame, fsdp_device.h Grp<G>) and instantiated for
// G = 8, 16 and 64 (eight, four, one frame per wavefront; the host picks per launch, see the end of this file): the
// stage is dominated by serial FP64 chains, and a serial instruction then advances all frames of the wavefront.
// Results do not depend on G (sums keep the reference's order).
// Memory: the working polyline (up to PATH_CAP points: x, y, parameter), the basis cache of the running fit, the rows
// of its smoothness matrix and the filtered curvature live in a per-frame HBM/L2 scratch arena (ARENA_DOUBLES
// doubles, lane-coalesced access); the serial sections consume it through LDS chunk buffers.  LDS per frame: the
// spline workspace, 4 KB (G = 8) to 6.7 KB (G = 64).
#pragma once
#include "fsdp_device.h"
#include "spline_device.h"
#include "det_math.h"

namespace fsdp {

constexpr int PATH_CAP = 1408;  // points of the working polyline (dense fit-#1 output + extension); the acceleration
                                // mission reaches ~1300 (outbound + return lane of its known path within 30 m)

constexpr int FIT_KNOTS = 16;  // knots the kernels of the three-kernel path stage keep per fit in LDS (more: exact kernel, 64)

// per-frame scratch in HBM/L2: working polyline x | y | parameter u, then the basis cache of the running fit
constexpr int ARENA_B = 1296;  // >= (NK_BIG + 2) * 5 rows of the smoothness matrix
constexpr int FITREC_DOUBLES = 112;  // >= sizeof(FitRec) / 8
constexpr int WIDE_KNOTS = 32;  // knots per fit of the WIDE instantiations of the three kernels (contexts with a global path: its fits need 17-32)
constexpr int BAND_DOUBLES = 256;    // >= 7 * (WIDE_KNOTS + 2): band triangle + right-hand sides + fpint of fit_kernel's fit
constexpr int ARENA_REC = 3 * PATH_CAP;                  // basis records (32 bytes per point), then one interval byte per point
constexpr int ARENA_BMAT = ARENA_REC + 4 * PATH_CAP + PATH_CAP / 8;  // rows of the smoothness matrix
constexpr int ARENA_FIT = ARENA_BMAT + ARENA_B + DENSE_CAP;          // FitRec
constexpr int ARENA_DOUBLES = ARENA_FIT + FITREC_DOUBLES + BAND_DOUBLES;  // x | y | u | records | intervals | b | filtered curvature | FitRec | band
// knots / coefficients of the refit (fit #2) on their way from fit_kernel to path_finish_kernel
struct FitRec {
  int32_t n, ier, status, pad;
  double fp;
  double t[34];
  double c[68];  // x coefficients [0, n), y coefficients [n, 2n)
};
static_assert(sizeof(FitRec) <= FITREC_DOUBLES * 8, "FitRec does not fit its arena region");
// what path_prep_kernel hands on: the polyline [off, off + n) of the arena is to be refitted (status 0), or the frame is
// finished / handed to the exact kernel (status != 0)
struct PathMid {
  int32_t status, fallback, off, n;
};
static_assert(ARENA_B >= DENSE_CAP, "raw curvature of a LEAN workspace lives in the smoothness-matrix rows");
static_assert(ARENA_DOUBLES % 8 == 0 && ARENA_REC % 8 == 0 && PATH_CAP % 64 == 0 && ARENA_BMAT % 8 == 0, "basis records must stay 32-byte aligned (the arena itself is 64-byte aligned)");
struct Arena {
  double* x;
  double* y;
  double* u;
  BasisCache bc;
  double* filt;  // filtered curvature of the dense samples
  double* curv;  // raw curvature of the dense samples where the LDS workspace has no room for it (LEAN): the rows of
                 // the smoothness matrix (bc.b), dead once the last fit is done
  FitRec* fit;
  double* band;  // fit_kernel's band triangle between observation passes (FitWS)
  const Params* prm;  // the context's configuration constants
  int frame;          // index of the frame within its batch
};

// Per-frame LDS = the spline workspace; while no fit is running the path stage uses its bytes for segment lengths (before
// fit #3), the <= 20 tail points of the extension, and after fit #3 the dense samples x | y (and, unless LEAN, the raw
// curvature; LEAN keeps that in the frame's scratch, Arena::curv).
template <int G, bool LEAN_ = false, int NKC = 0>  // NKC: knots per fit (0: the group size's default — 16 lean, 32 packed, 64 for a whole wavefront)
struct PathShared {
  static constexpr bool LEAN = LEAN_;
  using WS = SplineWS<G, NKC ? NKC : (LEAN_ ? FIT_KNOTS : knot_capacity<G>()), DENSE_CAP, LEAN_>;
  static constexpr int SEG_CAP = WS::DENSE_ARRAYS * DENSE_CAP;  // segment-length scratch in LDS
  WS ws;
  __device__ __forceinline__ double* seg() { return ws.dxyu; }
  __device__ __forceinline__ double* dx() { return ws.dxyu; }
  __device__ __forceinline__ double* dy() { return ws.dxyu + DENSE_CAP; }
};

// np.sum of a contiguous run (NumPy pairwise summation) — wave-uniform.  The recursion of
// DOUBLE_pairwise_sum (split n -> n/2 rounded down to a multiple of 8) is bounded at compile time.
__device__ __forceinline__ double pw_leaf(const double* b, int nn) {
  if (nn < 8) {
    double r = 0.0;
    for (int i = 0; i < nn; i++) r += b[i];
    return r;
  }
  double r0 = b[0], r1 = b[1], r2 = b[2], r3 = b[3], r4 = b[4], r5 = b[5], r6 = b[6], r7 = b[7];
  int i = 8;
  for (; i < nn - (nn % 8); i += 8) {
    r0 += b[i];
    r1 += b[i + 1];
    r2 += b[i + 2];
    r3 += b[i + 3];
    r4 += b[i + 4];
    r5 += b[i + 5];
    r6 += b[i + 6];
    r7 += b[i + 7];
  }
  double r = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
  for (; i < nn; i++) r += b[i];
  return r;
}
template <int DEPTH>
__device__ inline double pw_rec(const double* a, int n) {
  if (n <= 128) return pw_leaf(a, n);
  if constexpr (DEPTH == 0) {
    return pw_leaf(a, n);  // unreachable for n <= 128 * 2^DEPTH
  } else {
    int n2 = n / 2;
    n2 -= n2 % 8;
    return pw_rec<DEPTH - 1>(a, n2) + pw_rec<DEPTH - 1>(a + n2, n - n2);
  }
}
__device__ inline double np_sum_run(const double* a, int n) { return 0.0 + pw_rec<4>(a, n); }
// the same for runs of up to 8192 elements (the skidpad table's circle fits)
__device__ inline double np_sum_long(const double* a, int n) { return 0.0 + pw_rec<6>(a, n); }

// NumPy pairwise sums (n <= 128: 8 interleaved accumulators, then the tail) of NS series at once: f(i, v) fills v[0..NS)
// with the i-th element of every series.  Each series is summed exactly as np_sum_fn sums it; the operands are fetched
// and the shared sub-expressions evaluated once per i instead of once per series.
template <int NS, class F>
__device__ __forceinline__ void np_sum_multi(int n, F f, double (&out)[NS]) {
  double v[NS];
  if (n < 8) {
    double r[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) r[s] = 0.0;
    for (int i = 0; i < n; i++) {
      f(i, v);
#pragma unroll
      for (int s = 0; s < NS; s++) r[s] += v[s];
    }
#pragma unroll
    for (int s = 0; s < NS; s++) out[s] = 0.0 + r[s];
    return;
  }
  double acc[8][NS];
#pragma unroll
  for (int e = 0; e < 8; e++) {
    f(e, v);
#pragma unroll
    for (int s = 0; s < NS; s++) acc[e][s] = v[s];
  }
  int i = 8;
  for (; i < n - (n % 8); i += 8) {
#pragma unroll
    for (int e = 0; e < 8; e++) {
      f(i + e, v);
#pragma unroll
      for (int s = 0; s < NS; s++) acc[e][s] += v[s];
    }
  }
  double res[NS];
#pragma unroll
  for (int s = 0; s < NS; s++)
    res[s] = ((acc[0][s] + acc[1][s]) + (acc[2][s] + acc[3][s])) + ((acc[4][s] + acc[5][s]) + (acc[6][s] + acc[7][s]));
  for (; i < n; i++) {
    f(i, v);
#pragma unroll
    for (int s = 0; s < NS; s++) res[s] += v[s];
  }
#pragma unroll
  for (int s = 0; s < NS; s++) out[s] = 0.0 + res[s];
}

// utils/math_utils.py:579-646 circle_fit of points (px[idx0 + i], py[idx0 + i]), i < n (n <= 128)
__device__ inline void circle_fit(const double* px, const double* py, int idx0, int n, double& ocx, double& ocy,
                                  double& orad) {
  const double* X = px + idx0;
  const double* Y = py + idx0;
  const double dn = (double)n;
  double m2[2];
  np_sum_multi<2>(n, [&](int i, double (&v)[2]) {
    v[0] = X[i];
    v[1] = Y[i];
  }, m2);
  const double xm = m2[0] / dn, ym = m2[1] / dn;
  // Mxy, Mxx, Myy, Mxz, Myz, Mzz: six np.sum calls over Xi*Yi, Xi*Xi, Yi*Yi, Xi*Zi, Yi*Zi, Zi*Zi
  double m6[6];
  np_sum_multi<6>(n, [&](int i, double (&v)[6]) {
    const double a = X[i] - xm, b = Y[i] - ym;
    const double z = a * a + b * b;
    v[0] = a * b;
    v[1] = a * a;
    v[2] = b * b;
    v[3] = a * z;
    v[4] = b * z;
    v[5] = z * z;
  }, m6);
  double Mxy = m6[0] / dn, Mxx = m6[1] / dn, Myy = m6[2] / dn, Mxz = m6[3] / dn, Myz = m6[4] / dn, Mzz = m6[5] / dn;
  double Mz = Mxx + Myy;
  double Cov_xy = Mxx * Myy - Mxy * Mxy;
  double Var_z = Mzz - Mz * Mz;
  double A2 = 4 * Cov_xy - 3 * Mz * Mz - Mzz;
  double A1 = Var_z * Mz + 4.0 * Cov_xy * Mz - Mxz * Mxz - Myz * Myz;
  double A0 = Mxz * (Mxz * Myy - Myz * Mxy) + Myz * (Myz * Mxx - Mxz * Mxy) - Var_z * Cov_xy;
  double A22 = A2 + A2;
  double y = A0, x = 0.0;
  for (int it = 0; it < 99; it++) {
    double Dy = A1 + x * (A22 + 16.0 * x * x);
    double x_new = x - y / Dy;
    if (x_new == x || !isfinite(x_new)) break;
    double y_new = A0 + x_new * (A1 + x_new * (A2 + 4.0 * x_new * x_new));
    if (fabs(y_new) >= fabs(y)) break;
    x = x_new;
    y = y_new;
  }
  double det = x * x - x * Mz + Cov_xy;
  double Xc = (Mxz * (Myy - x) - Myz * Mxy) / det / 2.0;
  double Yc = (Myz * (Mxx - x) - Mxz * Mxy) / det / 2.0;
  ocx = Xc + xm;
  ocy = Yc + ym;
  orad = sqrt(fabs(Xc * Xc + Yc * Yc + Mz));
}

// sign-exact det([[1,x0,y0],[1,x1,y1],[1,x2,y2]]) as numpy.linalg.det computes it (LAPACK dgetrf of OpenBLAS)
__device__ inline double det3_lu(double x0, double y0, double x1, double y1, double x2, double y2) {
#define FABS fabs
#define FMA fma
  // OpenBLAS' unblocked LU (lapack/getf2/getf2.c: what dgetrf runs for n <= DTB_ENTRIES / 2), left-looking, on the
  // column-major copy NumPy hands to LAPACK; only the SIGN of the determinant is used by the callers.
  //   column 0 = (1,1,1): pivot row 0, multipliers 1 * (1 / 1) = 1.
  //   column 1: b_i = x_i - 1 * x_0 (gemv, alpha = -1); pivot = first largest |b_i|; rows swapped in columns 0..1;
  //             multiplier l21 = b_2 * (1 / b_1)  — scaled by the RECIPROCAL of the pivot (dscal), two roundings.
  //   column 2: pivots applied; u12 = y_1 - 1 * y_0 (forward substitution, ddot of one element);
  //             u22 = y_2 - t with t = fma(l21, u12, fma(l20, y_0, 0)) — dgemv_n's scalar tail for one row accumulates
  //             temp += a * x (contracted to an fma by the compiler the library is built with), then y += alpha * temp.
  // Checked against numpy.linalg.det on 64 000 exactly / nearly collinear and general triples: same sign on all of them
  // (tests/test_oracle_numpy_semantics.py); the textbook right-looking order agrees on 98 % only.
  double b1 = x1 - x0, b2 = x2 - x0;  // (l10 = l20 = 1)
  double ya = y1, yb = y2;
  int sign = 1;
  if (!(FABS(b1) >= FABS(b2))) {  // idamax: first of the largest
    double t = b1;
    b1 = b2;
    b2 = t;
    ya = y2;
    yb = y1;
    sign = -1;
  }
  if (b1 == 0.0) return 0.0;  // singular (info > 0): numpy returns 0
  const double l21 = b2 * (1.0 / b1);
  const double u12 = ya - y0;
  const double t = FMA(l21, u12, y0);  // fma(l20 = 1, y0, 0) = y0 exactly
  const double u22 = yb - t;
  return (double)sign * b1 * u22;
#undef FABS
#undef FMA
}

__device__ __forceinline__ double py_max(double a, double b) { return (b > a) ? b : a; }
__device__ __forceinline__ double py_min(double a, double b) { return (b < a) ? b : a; }

// The lane-parallel passes over a polyline in the frame's scratch are bound by the round trip of their loads (the prep
// and finish kernels run two wavefronts per SIMD, nothing else hides it): they fetch PASS_BATCH points per round — each
// lane its PASS_BATCH / G points, all loads in flight together — instead of one point per lane and round trip
// (profiles/r04_prep_finish_sections.txt).
constexpr int PASS_BATCH = 64;

// first closest point of the polyline (X, Y)[0, n) to (px, py) by norm_axis, "first smallest" (np.argmin): bi = -1 if n = 0
template <int G>
__device__ __forceinline__ void closest_point(const double* X, const double* Y, int n, double px, double py, double& bv, int& bi) {
  using GR = Grp<G>;
  constexpr int UN = PASS_BATCH / G;
  const int lane = GR::lane();
  bv = 0.0;
  bi = -1;
  for (int base = lane; base < n; base += PASS_BATCH) {
    double vx[UN], vy[UN];
#pragma unroll
    for (int q = 0; q < UN; q++) {
      int i = base + q * G;  // unconditional loads (index clamped): the round's fetches fly together
      i = i < n ? i : n - 1;
      vx[q] = X[i];
      vy[q] = Y[i];
    }
#pragma unroll
    for (int q = 0; q < UN; q++) {
      const int i = base + q * G;
      if (i < n) {
        double d = norm_axis(px - vx[q], py - vy[q]);
        if (bi < 0 || d < bv) {
          bv = d;
          bi = i;
        }
      }
    }
  }
  GR::argmin(bv, bi);
}

// segment lengths of the polyline (X, Y)[0, ns + 1) of one round: this lane's segments base + q * G + lane, q < UN
template <int G>
struct SegmentFetch {
  static constexpr int UN = PASS_BATCH / G;
  double ax[UN], ay[UN], bx[UN], by[UN];
  __device__ __forceinline__ void load(const double* X, const double* Y, int base, int ns) {
    const int lane = Grp<G>::lane();
#pragma unroll
    for (int q = 0; q < UN; q++) {
      int i = base + q * G + lane;  // (clamped: harmless past the end; ns >= 1)
      i = i < ns ? i : ns - 1;
      ax[q] = X[i];
      ay[q] = Y[i];
      bx[q] = X[i + 1];
      by[q] = Y[i + 1];
    }
  }
  __device__ __forceinline__ double length(int q) const {
    const double dx = bx[q] - ax[q], dy = by[q] - ay[q];
    return sqrt(dx * dx + dy * dy);
  }
};

// chord lengths -> parameter values: A.u[off + i] = cumulative length (np.cumsum: sequential order); returns max_u.
// PASS_BATCH segments per round: lengths lane-parallel into LDS (the dense-sample bytes: no fit is running), the next
// round's points fetched while every lane walks the sum in order and keeps the values of its own elements.
template <int G, class PS>
__device__ __forceinline__ double build_parameter(PS& S, const Arena& A, int off, int m) {
  PROF(19);
  using GR = Grp<G>;
  constexpr int NR = PASS_BATCH / G;
  static_assert(PASS_BATCH <= PS::SEG_CAP && G % 8 == 0 && PASS_BATCH % G == 0, "round buffer");
  double* const term = S.seg();
  const int lane = GR::lane();
  const double *X = A.x + off, *Y = A.y + off;
  const int ns = m - 1;
  double acc = 0.0;
  if (lane == 0) A.u[off] = 0.0;
  if (ns <= 0) return acc;
  SegmentFetch<G> sf;
  sf.load(X, Y, 0, ns);
  for (int base = 0; base < ns; base += PASS_BATCH) {
    const int cnt = (ns - base) < PASS_BATCH ? (ns - base) : PASS_BATCH;
#pragma unroll
    for (int q = 0; q < NR; q++)
      if (q * G + lane < cnt) term[q * G + lane] = sf.length(q);
    GR::sync();
    if (base + PASS_BATCH < ns) sf.load(X, Y, base + PASS_BATCH, ns);
    double mine[NR];
#pragma unroll
    for (int q = 0; q < NR; q++) {
      mine[q] = 0.0;
      if (q * G < cnt) {  // (group-uniform: the last round is short)
        for (int r0 = 0; r0 < G; r0 += 8) {  // operands eight at a time, additions in order
          double v[8];
#pragma unroll
          for (int e = 0; e < 8; e++) v[e] = term[q * G + r0 + e];
#pragma unroll
          for (int e = 0; e < 8; e++) {
            const int rr = r0 + e;
            if (q * G + rr < cnt) {
              acc += v[e];
              if (lane == rr) mine[q] = acc;  // lane rr keeps element q*G + rr
            }
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NR; q++)
      if (q * G + lane < cnt) A.u[off + base + q * G + lane + 1] = mine[q];  // coalesced stores
    GR::sync();
  }
  return acc;
}

// utils/spline_fit.py:95-128 on the arena polyline [off, off+m).  rc: 0 ok, 1 ValueError, >=200 overflow
// CUBIC (the kernels of the three-kernel path stage): only the degree-3 fit is compiled in; a polyline of fewer than
// four points sends the frame to the exact kernel (ST_RETRY).
template <int G, bool FAST, bool CUBIC = false, class PS>
__device__ __forceinline__ int fit_polyline(PS& S, const Arena& A, int off, int m, double smoothing, SplineFit& f,
                                   double& max_u, int max_deg = 3) {
  int k = m - 1;
  k = k < 1 ? 1 : (k > max_deg ? max_deg : k);  // np.clip(len(trace) - 1, 1, max_deg), utils/spline_fit.py:113
  if constexpr (CUBIC) {
    if (k < 3) return ST_RETRY;
    max_u = build_parameter<G>(S, A, off, m);
    f = spline_fit_k<3, FAST>(S.ws, A.bc, A.u + off, A.x + off, A.y + off, m, smoothing);
  } else {
    max_u = build_parameter<G>(S, A, off, m);
    f = spline_fit<FAST>(S.ws, A.bc, A.u + off, A.x + off, A.y + off, m, k, smoothing);
  }
  return f.status;
}
template <bool CUBIC, class WS>
__device__ __forceinline__ void eval_spline(const WS& ws, const SplineFit& f, double step, int count, double* OX, double* OY, double* OU) {
  if constexpr (CUBIC)
    spline_eval_k<3>(ws, f, step, count, OX, OY, OU);
  else
    spline_eval(ws, f, step, count, OX, OY, OU);
}

__device__ __forceinline__ int arange_len(double stop, double step) {
  double q = stop / step;
  if (!(q > 0)) return 0;
  return (int)ceil(q);
}

// calculate_path/path_parameterization.py:297-328 on the arena polyline [off, off+n).
// rc: 0 ok (out filled), 1 ValueError, ST_* otherwise.
template <int G, bool FAST, bool CUBIC = false, class PS>
__device__ __forceinline__ int parameterize_path(PS& S, const Arena& A, int off, int n, double (*out)[4], int* n_dense) {
  using GR = Grp<G>;
  const int lane = GR::lane();
  if (n < 2) return ST_REF_UNDEFINED_PATH;
  // _refit_spline :125-161 — segment lengths (LDS when they fit, else the arena's parameter array)
  double* seg = (n - 1 <= PS::SEG_CAP) ? S.seg() : (A.u + off);
  {
    SegmentFetch<G> sf;
    for (int base = 0; base < n - 1; base += PASS_BATCH) {
      sf.load(A.x + off, A.y + off, base, n - 1);
#pragma unroll
      for (int q = 0; q < SegmentFetch<G>::UN; q++) {
        const int i = base + q * G + lane;
        if (i < n - 1) seg[i] = sf.length(q);
      }
    }
  }
  GR::sync();
  double path_length = np_sum_run(seg, n - 1);
  int n10 = (n - 1) < 10 ? (n - 1) : 10;
  double mean_pd = np_sum_small(seg, n10) / (double)n10;
  const int H = A.prm->horizon;  // mpc_prediction_horizon: rows of the result (stride PATH_POINTS)
  double predict_every = path_length / H / 3;
  int skip;
  {
    double q = predict_every / mean_pd;
    if (isnan(q))
      skip = 1;
    else if (isinf(q))
      return ST_REF_UNDEFINED_PATH;
    else {
      skip = (int)q;
      if (skip < 1) skip = 1;
    }
  }
  GR::sync();
  int ns = n;
  if (skip > 1) {
    ns = (n + skip - 1) / skip;
    for (int base = 0; base < ns; base += G) {
      int i = base + lane;
      double vx = 0, vy = 0;
      if (i < ns) {
        vx = A.x[off + i * skip];
        vy = A.y[off + i * skip];
      }
      GR::sync();
      if (i < ns) {
        A.x[off + i] = vx;
        A.y[off + i] = vy;
      }
      GR::sync();
    }
  }
  SplineFit f;
  double max_u;
  int rc;
  {
    PROF(7);
    rc = fit_polyline<G, FAST, CUBIC>(S, A, off, ns, 0.01, f, max_u);
  }
  if (rc) return rc;
  // _calculate_path_curvature :163-193 — dense samples into LDS
  int L = arange_len(max_u, predict_every);
  if (L > DENSE_CAP) return ST_OVERFLOW_PATH;
  if (L == 0) return ST_REF_UNDEFINED_PATH;
  double* const DX = S.dx();
  double* const DY = S.dy();
  {
    PROF(8);
    eval_spline<CUBIC>(S.ws, f, predict_every, L, DX, DY, nullptr);
  }
  double* curv;
  if constexpr (PS::LEAN)
    curv = A.curv;
  else
    curv = S.ws.curv;
  double* filt = A.filt;
  int window = (L / 5) < 30 ? (L / 5) : 30;
  if (window % 2 == 0) window += 1;
  const int half = window / 2;
  {
  PROF(9);
  for (int i = lane; i < L; i += G) {
    // cyclic window cut at the wrap-around for an open path (path_parameterization.py:64-77)
    int lo = i - half, hi = i + half;
    int w0, wn;
    if (lo >= 0 && hi < L) {
      w0 = lo;
      wn = window;
    } else if (lo < 0) {  // wraps at the start (then i < window always): keep the part after the wrap
      w0 = 0;
      wn = hi + 1;
    } else if (i < window) {  // wraps at the end while i < window (only for very short paths): keeps the wrapped tail
      w0 = 0;
      wn = hi - L + 1;
    } else {  // wraps at the end: keep the part before the wrap
      w0 = lo;
      wn = L - lo;
    }
    double cx, cy, r;
    circle_fit(DX, DY, w0, wn, cx, cy, r);
    r = py_min(py_max(r, 1.0), 3000.0);
    double c = 1 / r;
    int i1 = wn / 2;
    double sg = det3_lu(DX[w0], DY[w0], DX[w0 + i1], DY[w0 + i1], DX[w0 + wn - 1], DY[w0 + wn - 1]);
    double cv = c * sign_of(sg);
    if (isnan(sg)) cv = sg;
    curv[i] = cv;
  }
  GR::sync();
  }
  PROF(18);
  // scipy.ndimage.uniform_filter1d(size = max(2, window // 2), mode = "nearest"): running sum in index order
  // (tmp += in[i + s2] - in[i - 1 - s1], filt[i] = tmp / size).  The differences are formed one per lane, the running sum
  // walks them in index order through the group's registers, and the lane that owns sample i stores filt[i].
  {
    const int size = (window / 2) > 2 ? (window / 2) : 2;
    const int s1 = size / 2, s2 = size - s1 - 1;
    auto clampi = [&](int q) { return q < 0 ? 0 : (q > L - 1 ? L - 1 : q); };
    double tmp = 0.0;
    for (int j = -s1; j <= s2; j++) tmp += curv[clampi(j)];
    if (lane == 0) filt[0] = tmp / size;
    constexpr int UN = PASS_BATCH / G;
    for (int base0 = 1; base0 < L; base0 += PASS_BATCH) {
      double dq[UN];  // the differences of UN rounds, fetched together (indices clamped: values past L are not used)
#pragma unroll
      for (int q = 0; q < UN; q++) {
        const int i = base0 + q * G + lane;
        dq[q] = curv[clampi(i + s2)] - curv[clampi(i - 1 - s1)];
      }
#pragma unroll
      for (int q = 0; q < UN; q++) {
        const int base = base0 + q * G;
        if (base < L) {
          const int i = base + lane;
          double mine = 0.0;
          const int cnt = (L - base) < G ? (L - base) : G;
          for (int r = 0; r < cnt; r++) {
            tmp += GR::bcast(dq[q], r);
            if (lane == r) mine = tmp;
          }
          if (i < L) filt[i] = mine / size;
        }
      }
    }
  }
  GR::sync();
  // _sample_path_parameters_for_prediction_horizon :252-295: np.linspace(0, L-1, 40, dtype=int)
  {
    const double step = ((double)(L - 1) - 0.0) / (double)(H - 1);
    auto sample_index = [&](int i) {
      if (H == 1) return 0;  // np.linspace(0, L - 1, 1) = [0.]
      double v = (double)i * step + 0.0;
      if (i == H - 1) v = (double)(L - 1);
      return (int)floor(v);
    };
    bool dup = false;
    for (int i = lane; i < H; i += G)
      if (i > 0 && sample_index(i - 1) == sample_index(i)) dup = true;
    if (GR::ballot(dup) != 0ull) return 1;  // "Indices of resampled path appear twice" (ValueError)
    for (int i = lane; i < PATH_POINTS; i += G) {
      if (i < H) {
        int idx = sample_index(i);
        out[i][0] = (double)idx * predict_every;  // np.arange(0, max_u, step)[idx]
        out[i][1] = DX[idx];
        out[i][2] = DY[idx];
        out[i][3] = filt[idx];
      } else {
        out[i][0] = out[i][1] = out[i][2] = out[i][3] = NAN;  // rows beyond the horizon do not exist in the reference
      }
    }
  }
  *n_dense = L;
  GR::sync();
#ifdef FSDP_DEBUG_DENSE
  // debug builds: the dense samples and the raw curvature where fsdp_debug_arena can read them (the polyline is dead here)
  for (int i = lane; i < L; i += G) {
    A.x[i] = DX[i];
    A.y[i] = DY[i];
    A.u[i] = curv[i];
  }
  GR::sync();
#endif
  return 0;
}

// sequential sum of segment lengths of the arena polyline [off, off+n) with optional early stop:
// returns the running total; *first_over = index of the first segment whose cumulative length exceeds
// `limit` (or n-1 if none).  Rounds of PASS_BATCH segments staged in LDS (as build_parameter); the additions keep
// np.cumsum's order.
template <int G, class PS>
__device__ __forceinline__ double cumulative_length(PS& S, const Arena& A, int off, int n, double limit, int* first_over) {
  using GR = Grp<G>;
  constexpr int NR = PASS_BATCH / G;
  double* const term = S.seg();
  const int lane = GR::lane();
  const double *X = A.x + off, *Y = A.y + off;
  const int ns = n - 1;
  double acc = 0.0;
  int first = n - 1;
  bool stop = false;
  if (ns > 0) {
    SegmentFetch<G> sf;
    sf.load(X, Y, 0, ns);
    for (int base = 0; base < ns && !stop; base += PASS_BATCH) {
      const int cnt = (ns - base) < PASS_BATCH ? (ns - base) : PASS_BATCH;
#pragma unroll
      for (int q = 0; q < NR; q++)
        if (q * G + lane < cnt) term[q * G + lane] = sf.length(q);
      GR::sync();
      if (base + PASS_BATCH < ns) sf.load(X, Y, base + PASS_BATCH, ns);
      for (int r0 = 0; r0 < cnt && !stop; r0 += 8) {  // operands eight at a time, additions in order
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = term[r0 + q];  // r0 + q < PASS_BATCH
#pragma unroll
        for (int q = 0; q < 8; q++) {
          if (!stop && r0 + q < cnt) {
            acc += v[q];
            if (acc > limit) {
              first = base + r0 + q;
              stop = true;
            }
          }
        }
      }
      GR::sync();
    }
  }
  if (first_over) *first_over = first;
  return acc;
}

// core_calculate_path.py:380-417 do_all_mpc_parameter_calculations on the polyline [1, 1+n) of the arena
// (slot 0 is reserved for the point connect_path_to_car may prepend), in three steps so that the refit can run in its
// own kernel: mpc_prepare (connect to the car, extend, trim behind the car -> polyline [off, off+n) to refit),
// the refit (utils/spline_fit.py, smoothing 0.2), mpc_finish (predict 30 m, cut at 20 m, parameterize).
// rc as parameterize_path.
template <int G, class PS>
__device__ __forceinline__ int mpc_prepare(PS& S, const Arena& A, int n, double px, double py, double dx, double dy,
                                           int* fallback, int* off_out, int* n_out) {
  using GR = Grp<G>;
  const int lane = GR::lane();
  if (n <= 0) return ST_REF_UNDEFINED_PATH;
  int off = 1;
  // connect_path_to_car :430-457
  {
    double fx = A.x[1], fy = A.y[1];
    double d = norm_blas(px - fx, py - fy);
    double cx = fx - px, cy = fy - py;
    double ang = angle_between(cx, cy, dx, dy);
    if (!(d < 0.5 || ang > FSDP_PI / 2)) {
      double nrm = norm_blas(cx, cy);
      GR::sync();
      if (lane == 0) {
        A.x[0] = px + (cx / nrm) * 0.2;
        A.y[0] = py + (cy / nrm) * 0.2;
      }
      off = 0;
      n += 1;
    }
  }
  GR::sync();
  // extend_path :261-334
  {
    // first index in front of the car (np.dot(car_to_path, direction) > 0), all later points count as in front
    int first = n;
    for (int base = 0; base < n; base += G) {
      int i = base + lane;
      bool fr = i < n && blas_dot2(A.x[off + i] - px, dx, A.y[off + i] - py, dy) > 0;
      unsigned long long m = GR::ballot(fr);
      if (m) {
        first = base + (__ffsll(m) - 1);
        break;
      }
    }
    int tail = n - 20 < 0 ? 0 : n - 20;
    int f0 = first < tail ? first : tail;  // mask[first:] = True, mask[-20:] = True  -> contiguous [f0, n)
    int nin = n - f0;
    if (nin >= 1) {
      if (nin < 2) return ST_REF_UNDEFINED_PATH;  // cumsum([])[-1] -> IndexError
      // path length in front of the car: np.cumsum of the segment lengths (sequential)
      double plen = cumulative_length<G>(S, A, off + f0, nin, INFINITY, nullptr);
      if (!(plen > A.prm->mpc_path_length)) {
        int nrel = nin < 20 ? nin : 20;
        int r0 = off + n - nrel;
        // the last <= 20 points through LDS for the circle fit
        double* const TX = S.dx();
        double* const TY = S.dy();
        for (int i = lane; i < nrel; i += G) {
          TX[i] = A.x[r0 + i];
          TY[i] = A.y[r0 + i];
        }
        GR::sync();
        double ccx, ccy, radius;
        circle_fit(TX, TY, 0, nrel, ccx, ccy, radius);
        double r_use = py_min(py_max(radius, 10), 100);
        const double lastx = TX[nrel - 1], lasty = TY[nrel - 1];
        int n_new;
        if (r_use < 80) {
          *fallback |= 16;
          int i1 = nrel / 2;
          double t0x = TX[0] - ccx, t0y = TY[0] - ccy;
          double t1x = TX[i1] - ccx, t1y = TY[i1] - ccy;
          double t2x = TX[nrel - 1] - ccx, t2y = TY[nrel - 1] - ccy;
          double sg = sign_of(det3_lu(t0x, t0y, t1x, t1y, t2x, t2y));
          // the only libm values that enter the float chain: deterministic correctly-rounded versions (det_math.h)
          double start = detm::det_atan2(t0y, t0x);
          double end = start + sg * FSDP_PI;
          const int NP = 50;
          double step = (end - start) / (double)(NP - 1);
          double s0, c0;
          detm::det_sincos(start, s0, c0);
          double raw0x = c0 * r_use, raw0y = s0 * r_use;  // i = 0: 0*step + start
          n_new = NP - 1;
          if (off + n + n_new > PATH_CAP) return ST_OVERFLOW_PATH;
          for (int i = 1 + lane; i < NP; i += G) {
            double a = (double)i * step + start;
            if (i == NP - 1) a = end;
            double sa, ca;
            detm::det_sincos(a, sa, ca);
            double rx = ca * r_use, ry = sa * r_use;
            A.x[off + n + i - 1] = rx - raw0x + lastx;
            A.y[off + n + i - 1] = ry - raw0y + lasty;
          }
        } else {
          *fallback |= 32;
          double ddx = lastx - TX[nrel - 2], ddy = lasty - TY[nrel - 2];
          double nrm = norm_blas(ddx, ddy);
          ddx /= nrm;
          ddy /= nrm;
          n_new = 29;
          if (off + n + n_new > PATH_CAP) return ST_OVERFLOW_PATH;
          for (int i = 1 + lane; i < 30; i += G) {
            A.x[off + n + i - 1] = lastx + ddx * (double)i;
            A.y[off + n + i - 1] = lasty + ddy * (double)i;
          }
        }
        n += n_new;
        GR::sync();
      }
    }
  }
  // remove_path_behind_car :459-465
  {
    double bv;
    int bi;
    closest_point<G>(A.x + off, A.y + off, n, px, py, bv, bi);
    off += bi;
    n -= bi;
  }
  GR::sync();
  *off_out = off;
  *n_out = n;
  return 0;
}

// after refit_path_for_mpc_with_safety_factor's fit :239-259 (knots / coefficients in S.ws, `fitted` false when the
// polyline had fewer than 2 points): predict to 1.5 * 20 m, cut at 20 m :467-499, parameterize
template <int G, bool FAST, bool CUBIC = false, class PS>
__device__ __forceinline__ int mpc_finish(PS& S, const Arena& A, bool fitted, const SplineFit& f, double (*out)[4],
                                          int* n_dense) {
  int n5;
  {
    int n4 = 0;
    if (fitted) {
      PROF(5);
      n4 = arange_len(A.prm->mpc_path_length * 1.5, A.prm->predict_every);
      if (n4 > PATH_CAP) return ST_OVERFLOW_PATH;  // (fsdp_create refuses such parameters; never write past the polyline)
      eval_spline<CUBIC>(S.ws, f, A.prm->predict_every, n4, A.x, A.y, nullptr);
    }
    int nseg = n4 - 1;
    if (nseg <= 1) return 1;  // previous (40,4) array handed on -> LinAlgError (a ValueError) downstream
    int first = nseg;
    cumulative_length<G>(S, A, 0, n4, A.prm->mpc_path_length, &first);
    n5 = first;
  }
  return parameterize_path<G, FAST, CUBIC>(S, A, 0, n5, out, n_dense);
}

template <int G, bool FAST, class PS>
__device__ __forceinline__ int do_all_mpc(PS& S, const Arena& A, int n, double px, double py, double dx, double dy,
                                 double (*out)[4], int* fallback, int* n_dense) {
  int off = 0;
  int rc = mpc_prepare<G>(S, A, n, px, py, dx, dy, fallback, &off, &n);
  if (rc) return rc;
  SplineFit f;
  f.k = 3;
  f.n = 0;
  const bool fitted = n >= 2;
  if (fitted) {
    double max_u;
    PROF(4);
    rc = fit_polyline<G, FAST>(S, A, off, n, A.prm->smoothing, f, max_u, A.prm->max_deg);
    if (rc) return rc;
  }
  return mpc_finish<G, FAST>(S, A, fitted, f, out, n_dense);
}

// overwrite_path_if_it_is_too_far_away :225-237 on the dense path update [1, 1+n1) of the arena; returns the new n1
template <int G>
__device__ __forceinline__ int overwrite_if_too_far(const Arena& A, int n1, double px, double py, const double* prev, int* fallback) {
  using GR = Grp<G>;
  const int lane = GR::lane();
  double bv;
  int bi;
  closest_point<G>(A.x + 1, A.y + 1, n1, px, py, bv, bi);
  if (bv > A.prm->maximal_distance_for_valid_path) {
    *fallback |= 4;
    GR::sync();
    for (int i = lane; i < A.prm->horizon; i += G) {
      A.x[1 + i] = prev[4 * i + 1];
      A.y[1 + i] = prev[4 * i + 2];
    }
    n1 = A.prm->horizon;
    GR::sync();
  }
  return n1;
}

// core_calculate_path.py:555-575: too-far check + MPC step with its ValueError retry, on the dense path update stored
// in the arena at [1, 1+n1); prev = previous path (40,4) rows [s, x, y, curvature].  Returns the frame status.
template <int G, bool FAST, class PS>
__device__ __forceinline__ int finish_path(PS& S, const Arena& A, int n1, double px, double py, double dx, double dy,
                                           const double* prev, double (*out)[4], int* fallback, int* n_dense) {
  using GR = Grp<G>;
  const int lane = GR::lane();
  if (n1 == 0) return ST_REF_UNDEFINED_PATH;  // min() of an empty array
  n1 = overwrite_if_too_far<G>(A, n1, px, py, prev, fallback);
  // one call site (the stage is inlined into the kernel so that every LDS access is a DS instruction): the second
  // round of the loop is the ValueError retry with the previous path (:564-570)
  int rc = 1;
  for (int attempt = 0; attempt < 2 && rc == 1; attempt++) {
    if (attempt == 1) {
      *fallback |= 8;
      GR::sync();
      for (int i = lane; i < A.prm->horizon; i += G) {
        A.x[1 + i] = prev[4 * i + 1];
        A.y[1 + i] = prev[4 * i + 2];
      }
      n1 = A.prm->horizon;
      GR::sync();
    }
    rc = do_all_mpc<G, FAST>(S, A, n1, px, py, dx, dy, out, fallback, n_dense);
  }
  if (rc == 1) rc = ST_REF_UNDEFINED_PATH;
  return rc;
}

// path_calculator_helpers.py:26-68 calculate_almost_straight_path (host side, libm = what NumPy uses)
inline void default_chord_points(double (*chord)[2]) {
  const int NP = CHORD_POINTS;
  const double max_angle = FSDP_PI / 50;
  const double step = (fabs(max_angle) - 0.0) / (double)(NP - 1);
  const double c = cos(-(FSDP_PI / 2)), s = sin(-(FSDP_PI / 2));
  for (int i = 0; i < NP; i++) {
    double a = (double)i * step + 0.0;
    if (i == NP - 1) a = fabs(max_angle);
    double px = (cos(a) - 1.0) * 1000.0, py = (sin(a) - 0.0) * 1000.0;
    double qx = fma(py, -s, px * c), qy = fma(py, c, px * s);
    chord[i][0] = qx;
    chord[i][1] = qy * 1.0;
  }
}

__device__ __forceinline__ Arena frame_arena(double* arena, int frame, const Params* prm) {
#ifdef FSDP_ARENA_ALIAS
  // experiment builds only (tools/build_variant.sh -DFSDP_ARENA_ALIAS=64 with a batch whose frames repeat with that
  // period): frames share scratch arenas, so the path stage's scratch stream stays in the L2 — what the kernels would
  // take if that stream cost nothing (profiles/r04_fit_memory_bound.txt)
  double* b = arena + (size_t)(frame % FSDP_ARENA_ALIAS) * ARENA_DOUBLES;
#else
  double* b = arena + (size_t)frame * ARENA_DOUBLES;
#endif
  Arena A;
  A.x = b;
  A.y = b + PATH_CAP;
  A.u = b + 2 * PATH_CAP;
  A.bc.rec = (BRec*)(b + ARENA_REC);  // 32-byte records on a 64-byte boundary: ARENA_DOUBLES and ARENA_REC are multiples of 8
  A.bc.l = (uint8_t*)(b + ARENA_REC + 4 * PATH_CAP);
  A.bc.b = b + ARENA_BMAT;
  A.filt = A.bc.b + ARENA_B;
  A.curv = A.bc.b;
  A.fit = (FitRec*)(A.filt + DENSE_CAP);
  A.band = A.filt + DENSE_CAP + FITREC_DOUBLES;
  A.prm = prm;
  A.frame = frame;
  return A;
}

// core_calculate_path.py:103-121: previous_paths[0] = parameterize_path(fit(chord).predict())
__global__ void __launch_bounds__(64) default_path_kernel(const double* __restrict__ chord, double* __restrict__ arena,
                                                          double* __restrict__ out, const Params* __restrict__ prm) {
  constexpr int G = WAVE;
  __shared__ PathShared<G> S;
  const int lane = lane_id();
  const Arena A = frame_arena(arena, 0, prm);
  if (lane < CHORD_POINTS) {
    A.x[lane] = chord[2 * lane];
    A.y[lane] = chord[2 * lane + 1];
  }
  __syncthreads();
  SplineFit f;
  double max_u;
  constexpr bool FAST = false;  // one-off per context: plain divisions
  int rc = fit_polyline<G, FAST>(S, A, 0, CHORD_POINTS, A.prm->smoothing, f, max_u, A.prm->max_deg);
  int n1 = arange_len(max_u, A.prm->predict_every);
  if (rc == 0 && n1 <= PATH_CAP) {
    spline_eval(S.ws, f, A.prm->predict_every, n1, A.x, A.y, nullptr);
    int nd = 0;
    double(*o)[4] = (double(*)[4])out;
    rc = parameterize_path<G, FAST>(S, A, 0, n1, o, &nd);
  }
  if (rc != 0 && lane < PATH_POINTS)
    for (int q = 0; q < 4; q++) out[4 * lane + q] = NAN;
}

// Lanes per frame of the product path kernel (see the header comment).  Three instantiations, chosen per launch by the
// host (fsdp_lib.hip launch_path): G = 8 (eight frames per wavefront) when passes overlap or a pass has more than
// 4096 frames — the regime is throughput and a serial instruction should advance as many frames as the LDS allows;
// G = 16 for a single pass of up to 4096 frames (one wavefront per SIMD, lowest latency); G = 64 for small batches
// (one frame per wavefront).  Measured: tools/batch_sweep.py, FSDP_PATH_G=8|16|64 pins the choice.
constexpr int PATH_G_THROUGHPUT = 8;
constexpr int PATH_G_LATENCY = 16;
constexpr int PATH_G_SMALL = 64;
constexpr int PATH_SMALL_BATCH = 1024;    // frames at or below which every frame gets its own wavefront
constexpr int PATH_G_SPLIT = 8;           // lanes per frame of path_prep_kernel / path_finish_kernel (three-kernel path stage)
constexpr int PATH_LATENCY_BATCH = 4096;  // largest single pass that G = 16 serves with one wavefront per SIMD

// First half of run_path_calculation (core_calculate_path.py:514-553): centre points (or the global path window), fit #1
// and its dense evaluation.  Leaves the dense path update in the arena [1, 1 + *n1_out); returns the frame status.
template <int G, bool FAST, bool CUBIC = false, class PS>
__device__ __forceinline__ int path_front(PS& S, const Arena& A, const MatchOut* mo, double px, double py,
                                          const double* prev, const double* __restrict__ gpath, int n_gpath, int* fallback_out,
                                          int* n1_out) {
  using GR = Grp<G>;
  const int lane = GR::lane();
  PROF_T0(2);
  int status = mo->status;
  int fallback = 0;
  const int nl = mo->n_left_v, nr = mo->n_right_v;
  int nc = 0;  // centre points, written to the arena polyline [0, nc)
  // A car position that is not finite: whatever branch the reference takes below, the position joins the path it fits
  // (connect_path_to_car, :430-457), the fit raises ValueError, and so does the retry with the previous path (:564-570).
  // Decided here, for all lanes alike, rather than by arg-mins over NaN distances (which lanes would not agree on).
  // (Pinned by tests/golden/nonfinite_poses.npz; a direction that is not finite plans normally, as in the reference.)
  if (status == ST_OK && !(fabs(px) < INFINITY && fabs(py) < INFINITY)) status = ST_REF_UNDEFINED_PATH;
  if (status == ST_OK && gpath != nullptr) {
    // PathPlanner.global_path is set (full_pipeline.py:81-82,181-183; core_calculate_path.py:514-529): the basis of the
    // path is the part of the global path within 30 m of the car, rolled so that it starts a third of the table before
    // the closest point (np.roll); the matches are ignored
    if (n_gpath <= 0) {
      status = ST_REF_UNDEFINED_PATH;  // argmin of an empty array
    } else {
      double bv = 0.0;
      int bi = -1;
      for (int i = lane; i < n_gpath; i += G) {
        double d = norm_axis(px - gpath[2 * i], py - gpath[2 * i + 1]);
        if (bi < 0 || d < bv) {
          bv = d;
          bi = i;
        }
      }
      GR::argmin(bv, bi);  // first smallest
      const int roll = -bi + n_gpath / 3;
      bool overflow = false;
      for (int base = 0; base < n_gpath; base += G) {
        const int k = base + lane;
        bool keep = false;
        double gx = 0.0, gy = 0.0;
        if (k < n_gpath) {
          int src = (k - roll) % n_gpath;  // np.roll: out[k] = in[(k - roll) mod n]
          if (src < 0) src += n_gpath;
          gx = gpath[2 * src];
          gy = gpath[2 * src + 1];
          keep = norm_axis(px - gx, py - gy) < 30;
        }
        const unsigned long long km = GR::ballot(keep);
        if (keep) {
          const int p = nc + __popcll(km & ((1ull << lane) - 1ull));
          if (p < PATH_CAP) {
            A.x[p] = gx;
            A.y[p] = gy;
          }
        }
        nc += __popcll(km);
      }
      if (nc > PATH_CAP) overflow = true;
      if (overflow)
        status = ST_OVERFLOW_PATH;
      else if (nc < 2)
        status = ST_REF_UNDEFINED_PATH;  // NullSplineEvaluator -> empty path update -> min() of an empty array (:232)
    }
    GR::sync();
  } else if (status == ST_OK) {
    bool use_prev = false;
    if (nl < 3 && nr < 3) {
      use_prev = true;
    } else {
      // select_side_to_use :151-183 (<= 24 entries per side: group-uniform scalar loops)
      int cntl = 0, cntr = 0, sl = 0, sr = 0;
      for (int i = 0; i < nl; i++) {
        int v = mo->l2r[i];
        if (v != -1) {
          cntl++;
          sl += v;
        }
      }
      for (int i = 0; i < nr; i++) {
        int v = mo->r2l[i];
        if (v != -1) {
          cntr++;
          sr += v;
        }
      }
      bool use_left = !((cntr > cntl) || (cntr == cntl && sr > sl));
      const int ns = use_left ? nl : nr, no = use_left ? nr : nl;
      if (ns > 0 && no == 0) {
        status = ST_REF_UNDEFINED_MATCH_IDX;
      } else {
        const int32_t* mv = use_left ? mo->l2r : mo->r2l;
        const double(*sv)[2] = use_left ? mo->left_v : mo->right_v;
        const double(*ov)[2] = use_left ? mo->right_v : mo->left_v;
        bool bad = false;
        int p = 0;
        for (int i = 0; i < ns; i++) {
          const int mine = mv[i];
          const int j = mine < 0 ? no + mine : mine;
          if (j < 0 || j >= no) {
            bad = true;
          } else if (mine != -1 && !bad && lane == 0) {
            A.x[p] = (sv[i][0] + ov[j][0]) / 2;
            A.y[p] = (sv[i][1] + ov[j][1]) / 2;
          }
          if (mine != -1) p++;
        }
        if (bad) status = ST_REF_UNDEFINED_MATCH_IDX;
        nc = p;
        if (nc < 2) use_prev = true;
      }
    }
    GR::sync();
    if (use_prev) {
      fallback |= 1;
      for (int i = lane; i < A.prm->horizon; i += G) {
        A.x[i] = prev[4 * i + 1];
        A.y[i] = prev[4 * i + 2];
      }
      nc = A.prm->horizon;
    }
    GR::sync();
  }
  if (A.prm->centers != nullptr) {
    // run_path_calculation's second return value (:575): the points the first fit is given, whichever branch chose them
    const int nk = status == ST_OK ? nc : 0, cap = A.prm->centers_cap;
    double* co = A.prm->centers + (size_t)A.frame * 2 * (size_t)cap;
    for (int i = lane; i < nk && i < cap; i += G) {
      co[2 * i] = A.x[i];
      co[2 * i + 1] = A.y[i];
    }
    if (lane == 0) A.prm->n_centers[A.frame] = nk;
  }
  PROF_T1(2);
  // fit_matches_as_spline :207-223 -> dense path update in arena [1, 1+n1)
  int n1 = 0;
  if (status == ST_OK) {
    for (int attempt = 0; attempt < 2; attempt++) {
      SplineFit f;
      double max_u;
      PROF(1);
      int rc = fit_polyline<G, FAST, CUBIC>(S, A, 0, nc, A.prm->smoothing, f, max_u, A.prm->max_deg);
      if (rc == 0) {
        n1 = arange_len(max_u, A.prm->predict_every);
        if (n1 + 1 + 50 > PATH_CAP) {
          status = ST_OVERFLOW_PATH;
        } else {
          PROF_T0(3);
          eval_spline<CUBIC>(S.ws, f, A.prm->predict_every, n1, A.x + 1, A.y + 1, nullptr);
          PROF_T1(3);
        }
        break;
      }
      if (rc != 1) {
        status = rc;
        break;
      }
      if (attempt == 1) {
        status = ST_REF_UNDEFINED_PATH;  // the fallback fit raised as well
        break;
      }
      fallback |= 2;
      for (int i = lane; i < A.prm->horizon; i += G) {
        A.x[i] = prev[4 * i + 1];
        A.y[i] = prev[4 * i + 2];
      }
      nc = A.prm->horizon;
      GR::sync();
    }
  }
  *fallback_out = fallback;
  *n1_out = n1;
  return status;
}

template <int G>
__device__ __forceinline__ void write_path_status(PathOut* o, int status, int fallback, int n_dense) {
  using GR = Grp<G>;
  const int lane = GR::lane();
  GR::sync();
  if (status != ST_OK)
    for (int i = lane; i < PATH_POINTS; i += G)
      for (int q = 0; q < 4; q++) o->path[i][q] = NAN;
  if (lane == 0) {
    o->status = status;
    o->fallback = fallback;
    o->n_dense = n_dense;
    o->pad = 0;
  }
}

// the whole path stage of one frame on one lane group
template <int G, bool FAST, class PS>
__device__ __forceinline__ int path_frame(PS& S, int frame, const double* __restrict__ poses,
                                  const MatchOut* __restrict__ matched, const double* __restrict__ default_path,
                                  const double* __restrict__ prev_paths, const double* __restrict__ gpath, int n_gpath,
                                  double* __restrict__ arena, PathOut* __restrict__ out, const Params* __restrict__ prm) {
  PROF(0);
  const Arena A = frame_arena(arena, frame, prm);
  PathOut* o = &out[frame];
  const double px = poses[4 * frame + 0], py = poses[4 * frame + 1], dx = poses[4 * frame + 2], dy = poses[4 * frame + 3];
  // previous_paths[-1]: the constant initial path (fresh planner) or, for sequential replays, the caller-supplied
  // previous output of this planner (core_calculate_path.py:572-573)
  const double* prev = prev_paths ? prev_paths + (size_t)frame * (PATH_POINTS * 4) : default_path;
  int fallback = 0, n_dense = 0, n1 = 0;
  int status = path_front<G, FAST>(S, A, &matched[frame], px, py, prev, gpath, n_gpath, &fallback, &n1);
  if (status == ST_OK) status = finish_path<G, FAST>(S, A, n1, px, py, dx, dy, prev, o->path, &fallback, &n_dense);
  write_path_status<G>(o, status, fallback, n_dense);
  return status;
}

// ---- the path stage as three kernels (large batches) --------------------------------------------------------------------
// 72 % of the one-kernel stage is the refit of the dense path update (fit #2: 450-500 points, ~10 knots, 2-5 observation
// passes, 3-19 smoothing iterations).  Split off, it becomes a kernel that holds nothing but one degree-3 fit — few enough
// registers and LDS for two wavefronts per SIMD, 16 fits per SIMD in flight — while the lane-parallel rest runs before
// (path_prep_kernel) and after it (path_finish_kernel).  Whatever is not the plain case (a fit that raises, fewer than 4
// points, a knot set beyond the packed capacity, an operand outside the fast division's exponent band, the ValueError
// retry of core_calculate_path.py:564-570) is appended to the retry list and planned from scratch by the exact
// one-frame-per-wavefront kernel (path_retry_kernel), so results do not depend on the route a frame took.
__device__ __forceinline__ void push_retry(int* retry, int frame) { retry[1 + atomicAdd(&retry[0], 1)] = frame; }

template <int G, int NKC = FIT_KNOTS>  // NKC: knots per fit the frame's LDS workspace keeps (WIDE_KNOTS: contexts with a global path)
// (two wavefronts per SIMD: 256 registers + 44 bytes of spill instead of 259 registers and one wavefront)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) path_prep_kernel(int n_frames, const double* __restrict__ poses, const MatchOut* __restrict__ matched,
                                                       const double* __restrict__ default_path, const double* __restrict__ prev_paths,
                                                       const double* __restrict__ gpath, int n_gpath, double* __restrict__ arena,
                                                       PathOut* __restrict__ out, PathMid* __restrict__ mid, int* __restrict__ retry,
                                                       const Params* __restrict__ prm) {
  using GR = Grp<G>;
  __shared__ PathShared<G, true, NKC> S_all[WAVE / G];
  const int frame = blockIdx.x * (WAVE / G) + GR::index();
  PROF_INIT_K(2);
  if (frame < n_frames) {
  PROF(0);
  PathShared<G, true, NKC>& S = S_all[GR::index()];
  const Arena A = frame_arena(arena, frame, prm);
  const double px = poses[4 * frame + 0], py = poses[4 * frame + 1], dx = poses[4 * frame + 2], dy = poses[4 * frame + 3];
  const double* prev = prev_paths ? prev_paths + (size_t)frame * (PATH_POINTS * 4) : default_path;
  int fallback = 0, n1 = 0, off = 0, n = 0;
  int status = path_front<G, true, true>(S, A, &matched[frame], px, py, prev, gpath, n_gpath, &fallback, &n1);
  bool plain = false;
  if (status == ST_OK && n1 > 0) {
    PROF_T0(6);
    n1 = overwrite_if_too_far<G>(A, n1, px, py, prev, &fallback);
    PROF_T1(6);
    PROF_T0(30);
    const int rc = mpc_prepare<G>(S, A, n1, px, py, dx, dy, &fallback, &off, &n);
    PROF_T1(30);
    plain = rc == 0 && n >= 4;  // degree 3 needs 4 points; everything else takes the exact route
    if (plain) build_parameter<G>(S, A, off, n);
  }
  const bool final_status = status != ST_OK && status != ST_RETRY && status != ST_OVERFLOW_KNOTS;
  if (final_status) write_path_status<G>(&out[frame], status, fallback, 0);  // sorting / matching / fit #1 decided the frame
  if (GR::lane() == 0) {
    PathMid m;
    m.status = plain ? ST_OK : (final_status ? status : ST_RETRY);
    m.fallback = fallback;
    m.off = off;
    m.n = n;
    mid[frame] = m;
    if (!plain && !final_status) push_retry(retry, frame);
  }
  }
  PROF_FLUSH_K(2);
}

#ifndef FSDP_FIT_WAVES
#define FSDP_FIT_WAVES 3
#endif
#ifndef FSDP_FIT4_WAVES
#define FSDP_FIT4_WAVES 2
#endif
#ifdef FSDP_EMU
#define FSDP_WAVES_PER_EU(n)  // (the host emulator's compiler does not parse an expression in an attribute it does not know)
#else
#define FSDP_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n)))
#endif
// the refit: utils/spline_fit.py:95-128 (splprep, k = 3, s = 0.2) of the arena polyline [off, off + n), G lanes per frame
template <int G, int NKC>
// (8 / 16 lanes per frame: three wavefronts per SIMD at 168 registers, measured +2 % frames/s over two; 4 lanes per frame:
// sixteen frames' workspaces are 19.2 KB, i.e. two wavefronts per SIMD — a third one at 14.2 KB / 168 registers was measured
// to change nothing, profiles/r04_ab_variants.txt 1)
__global__ void __launch_bounds__(64) FSDP_WAVES_PER_EU(G == 4 ? FSDP_FIT4_WAVES : FSDP_FIT_WAVES) fit_kernel(int n_frames, double* __restrict__ arena, PathMid* __restrict__ mid,
                                                 int* __restrict__ retry, const Params* __restrict__ prm,
                                                 unsigned long long* __restrict__ clock_first, unsigned long long* __restrict__ clock_last) {
  using GR = Grp<G>;
  using WS = FitWS<G, NKC>;
#ifndef FSDP_EMU
  // optional (fsdp_time_runs): when did the launch's first wavefront start and its last one end, on the device's constant-rate
  // clock — the kernel's duration as a kernel trace reports it, without the wait of its queue that an event bracket includes
  if (clock_first && threadIdx.x == 0) atomicMin(clock_first, (unsigned long long)wall_clock64());
#endif
  static_assert(7 * (NKC + 2) <= BAND_DOUBLES, "band region of the arena");
  __shared__ WS ws_all[WAVE / G];
  const int frame = blockIdx.x * (WAVE / G) + GR::index();
  PROF_INIT_K(1);
  if (frame < n_frames && mid[frame].status == ST_OK) {
    PROF(0);
    WS& ws = ws_all[GR::index()];
    const Arena A = frame_arena(arena, frame, prm);
    if (GR::lane() == 0) ws.band = A.band;
    GR::sync();
    const int off = mid[frame].off, m = mid[frame].n;
    const SplineFit f = spline_fit_k<3, true>(ws, A.bc, A.u + off, A.x + off, A.y + off, m, A.prm->smoothing);
    const int lane = GR::lane();
    if (f.status != 0) {
      if (lane == 0) {
        mid[frame].status = ST_RETRY;
        push_retry(retry, frame);
      }
    } else {
      FitRec* fr = A.fit;
      for (int i = lane; i < f.n; i += G) {
        fr->t[i] = ws.t[1 + i];
        fr->c[i] = ws.c[1 + i];
        fr->c[f.n + i] = ws.c[1 + f.n + i];
      }
      if (lane == 0) {
        fr->n = f.n;
        fr->ier = f.ier;
        fr->status = 0;
        fr->fp = f.fp;
      }
    }
  }
  PROF_FLUSH_K(1);
#ifndef FSDP_EMU
  if (clock_last && threadIdx.x == 0) atomicMax(clock_last, (unsigned long long)wall_clock64());
#endif
}

template <int G, int NKC = FIT_KNOTS>
__global__ void __launch_bounds__(64) path_finish_kernel(int n_frames, double* __restrict__ arena, PathMid* __restrict__ mid,
                                                         PathOut* __restrict__ out, int* __restrict__ retry,
                                                         const Params* __restrict__ prm) {
  using GR = Grp<G>;
  __shared__ PathShared<G, true, NKC> S_all[WAVE / G];
  const int frame = blockIdx.x * (WAVE / G) + GR::index();
  PROF_INIT_K(3);
  if (frame < n_frames && mid[frame].status == ST_OK) {
  PROF(0);
  PathShared<G, true, NKC>& S = S_all[GR::index()];
  const Arena A = frame_arena(arena, frame, prm);
  const int lane = GR::lane();
  const FitRec* fr = A.fit;
  SplineFit f;
  f.k = 3;
  f.n = fr->n;
  f.ier = fr->ier;
  f.fp = fr->fp;
  f.status = 0;
  for (int i = lane; i < f.n; i += G) {
    S.ws.t[1 + i] = fr->t[i];
    S.ws.c[1 + i] = fr->c[i];
    S.ws.c[1 + f.n + i] = fr->c[f.n + i];
  }
  GR::sync();
  int n_dense = 0;
  const int rc = mpc_finish<G, true, true>(S, A, true, f, out[frame].path, &n_dense);
  if (rc == 0) {
    write_path_status<G>(&out[frame], ST_OK, mid[frame].fallback, n_dense);
  } else if (lane == 0) {
    mid[frame].status = ST_RETRY;
    push_retry(retry, frame);
  }
  }
  PROF_FLUSH_K(3);
}

// grid = ceil(n_frames / (64 / G)) workgroups of one wavefront; group g of block b plans frame b * (64 / G) + g.
// retry (optional): [0] = counter, [1..] = frames that ended with ST_OVERFLOW_KNOTS (the packed kernels keep 32 knots per
// fit); path_retry_kernel plans those again with the one-frame-per-wavefront instantiation (256 knots).
#ifndef FSDP_PATH_WAVES
#define FSDP_PATH_WAVES 1
#endif
template <int G, bool FAST = true>
__global__ void __launch_bounds__(64, FSDP_PATH_WAVES) path_kernel(int n_frames, const double* __restrict__ poses,
                                                     const MatchOut* __restrict__ matched,
                                                     const double* __restrict__ default_path,
                                                     const double* __restrict__ prev_paths,
                                                     const double* __restrict__ gpath, int n_gpath,
                                                     double* __restrict__ arena, PathOut* __restrict__ out,
                                                     int* __restrict__ retry, const Params* __restrict__ prm) {
  __shared__ PathShared<G, false, G == WAVE ? NK_BIG : 0> S_all[WAVE / G];  // (a frame with the wavefront to itself: 256 knots)
  const int frame = blockIdx.x * (WAVE / G) + Grp<G>::index();
  PROF_INIT();
  if (frame < n_frames) {
    path_frame<G, FAST>(S_all[Grp<G>::index()], frame, poses, matched, default_path, prev_paths, gpath, n_gpath, arena, out, prm);
    if (retry != nullptr && Grp<G>::lane() == 0 && (out[frame].status == ST_OVERFLOW_KNOTS || out[frame].status == ST_RETRY))
      retry[1 + atomicAdd(&retry[0], 1)] = frame;
  }  // (retry list on the device: used by the emulator harness; the library collects the list on the host)
  PROF_FLUSH();
}

// The frames the packed kernels handed on (device list: more than 16 knots in a fit, a degree below 3, the ValueError retry,
// an operand outside the fast division's exponent band, a global-path slice): the whole path stage of such a frame from
// scratch.  Two levels inside one kernel: first FOUR frames per wavefront with 16 lanes each (64 knots per fit like the
// whole-wavefront form — with 32 most frames of a noisy batch came back for the second level — and the scaling-free divisions) — a frame then costs a quarter of a wavefront instead of a whole SIMD (round 3: one frame per
// wavefront at one wavefront per SIMD: 3 % of the frames of a noisy batch took 60 % of the chip's time); what that form
// cannot finish (an exponent outside the fast division's band) is planned once more by the whole wavefront with plain IEEE
// divisions.  The route never changes a result (tests: test_every_path_kernel_instantiation_equals_oracle).
#ifndef FSDP_RETRY_WAVES
#define FSDP_RETRY_WAVES 1  // (two per SIMD: 107 registers spilled to scratch, slower: profiles/r04_ab_variants.txt 4)
#endif
__global__ void __launch_bounds__(64, FSDP_RETRY_WAVES) path_retry_kernel(const double* __restrict__ poses, const MatchOut* __restrict__ matched,
                                                           const double* __restrict__ default_path,
                                                           const double* __restrict__ prev_paths,
                                                           const double* __restrict__ gpath, int n_gpath,
                                                           double* __restrict__ arena, PathOut* __restrict__ out,
                                                           const int* __restrict__ retry, const Params* __restrict__ prm) {
  constexpr int G1 = 16, PER = WAVE / G1;
  union Shared {
    PathShared<G1, false, NK_MAX> quad[PER];
    PathShared<WAVE, false, NK_BIG> whole;
    __device__ Shared() {}
  };
  __shared__ Shared S;
  const int n = retry[0];
  // A short list (a few frames of a noisy batch): the chip is idle next to them and the pass waits for the slowest one —
  // a whole wavefront per frame is the quickest way through (config 4r: 60 frames, 4.9 ms instead of 5.8).  A long one (every
  // frame of a global-path / acceleration batch): chip time counts — four frames per wavefront (+ 90 % frames/s there).
  if (n <= prm->retry_pack_min) {
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
      path_frame<WAVE, false>(S.whole, retry[1 + i], poses, matched, default_path, prev_paths, gpath, n_gpath, arena, out, prm);
      __syncthreads();
    }
    return;
  }
  const int g = Grp<G1>::index();
  for (int base = blockIdx.x * PER; base < n; base += gridDim.x * PER) {
    const int i = base + g;
    const int frame = i < n ? retry[1 + i] : -1;
    int st = ST_OK;
    if (frame >= 0) st = path_frame<G1, true>(S.quad[g], frame, poses, matched, default_path, prev_paths, gpath, n_gpath, arena, out, prm);
    __syncthreads();
    for (int q = 0; q < PER; q++) {
      const int fq = __shfl(frame, q * G1), sq = __shfl(st, q * G1);
      if (fq >= 0 && (sq == ST_OVERFLOW_KNOTS || sq == ST_RETRY)) {
        path_frame<WAVE, false>(S.whole, fq, poses, matched, default_path, prev_paths, gpath, n_gpath, arena, out, prm);
        __syncthreads();
      }
    }
  }
}

}  // namespace fsdp
