// Smoothing B-spline fit + evaluation on one lane group (G lanes of a wavefront, gfx950).
//
// What the reference reaches through SciPy (utils/spline_fit.py:117 splprep, :61 splev) is
// Dierckx's FITPACK parcur/fppara: adaptive knot placement, Givens-QR least squares on a banded
// observation matrix, and a rational iteration on the smoothing parameter p.  The path stage's
// discrete decisions (how many dense samples the output is drawn from) hang on the last bit of
// these results, so this implementation keeps FITPACK's operation ORDER wherever a rounding
// could differ, and uses the group's lanes only where the result is order-independent:
//   * per-data-point work (knot-interval search, de Boor basis values, residual terms, curve
//     evaluation) runs one point per lane, staged through LDS in chunks of CH rows;
//   * the row-by-row Givens rotations, back-substitution, knot bookkeeping and all running sums
//     are group-uniform (every lane carries the same scalars; LDS reads broadcast).
// Everything is templated on the group size G (fsdp_device.h Grp<G>): G = 64 is one frame per wavefront,
// G = 16 packs four frames into a wavefront so that the serial sections advance four fits per instruction.
// Arrays are 1-based like the published algorithm.  idim = 2, unit weights, iopt = 0.
#pragma once
#include "fsdp_device.h"

namespace fsdp {

// Max number of knots kept in LDS (FITPACK's nest is m + 2k; OVERFLOW_KNOTS beyond): 32 where several frames share a
// wavefront's LDS, 64 where a frame has the wavefront to itself.  The host re-runs frames that overflow the packed
// kernels through the G = 64 kernel (fsdp_lib.hip launch_path), so results do not depend on the packing.
template <int G>
constexpr int knot_capacity() {
  return G == 64 ? 64 : 32;
}
constexpr int NK_MAX = 64;
// The exact kernels' last level — one frame, the whole wavefront, plain IEEE divisions (path_kernel<64>, the second level of
// path_retry_kernel) — keeps 256 knots: the most a knot interval stored in one byte per data point allows (BasisCache::l holds
// l <= n - 4 = 252), 29 KB of LDS for a kernel that runs one wavefront per SIMD anyway.  FITPACK itself allows nest = m + 2k knots
// (utils/spline_fit.py:117); the noisiest frames of the fuzz sets end with 68 ... 171 (profiles/r05_fuzz_gpu_vs_oracle_wide.txt), which
// 64 knots refused with FSDP_OVERFLOW_KNOTS.
constexpr int NK_BIG = 256;

// dense samples of the final spline: 3 x mpc_prediction_horizon, or one more (path_parameterization.py:163-193: 120 or 121
// at the default horizon of 40; up to 193 in the wide build's 64 rows)
constexpr int DENSE_CAP = (3 * PATH_POINTS + 1 <= 128) ? 128 : 200;

// LDS workspace of one frame: 4.8 KB at G = 16/32, 4 KB at G = 8 (eight frames of a wavefront: five workgroups per CU).  Three
// lifetimes share the bytes: what a running fit always needs (knots, coefficients, the band triangle and its
// right-hand sides), what only its observation / residual passes need (chunk buffers, knot bookkeeping), what only the
// smoothing iteration needs (the extended triangle g; the f(p) term buffer overlays it while it is dead), and — when no
// fit is running — the path stage's dense samples.  The rows of the smoothness matrix b live in the frame's scratch.
//
// LEAN (the prep / finish kernels of the three-kernel path stage, NKC = 16): the no-fit view is the knots | coefficients of
// the last fit left untouched, then the dense samples x | y only — the raw curvature goes through the frame's scratch —
// which makes a frame 2 480 bytes, eight frames of a wavefront 19.4 KB (eight workgroups per CU instead of five).
template <int G, int NKC = knot_capacity<G>(), int DCAP = DENSE_CAP, bool LEAN_ = false>  // DCAP = 1: a fit-only workspace
struct SplineWS {
  static constexpr int GRP = G;
  static constexpr int NK = NKC;
  static constexpr bool LEAN = LEAN_;
  static constexpr int DENSE_ARRAYS = LEAN_ ? 2 : 3;
  static constexpr int CH = (G >= 32) ? G : (G >= 16 ? 32 : 16);  // data rows staged per chunk
  union {
    struct {  // ---- a fit in progress ----
      double t[NK + 2];
      double c[2 * (NK + 2)];
      double z[2 * (NK + 2)];
      double a_[NK + 2][4];  // band columns 1..4 of FITPACK's a (see A())
      union {
        struct {  // part 1: observation passes, residual pass, knot selection
          double hq[CH][4];       // per-chunk basis values; residual terms of a super-chunk in the residual / f(p) passes
          double xq[CH], yq[CH];  // per-chunk data points (K < 3: rotated-out right-hand sides); "new interval" flags
          double term[CH];        // per-chunk segment lengths (path stage)
          int32_t lq[CH];         // per-chunk knot interval
          double fpint[NK + 2];
          int32_t nrdata[NK + 2];
        };
        double g_[NK + 2][5];  // part 2: columns 1..5 of the extended triangle (see Gm())
      };
    };
    struct {  // ---- no fit running (path stage) ----
      // raw curvature; overlays t | c: written only after the last spline evaluation (LEAN: never written, the bytes of t | c).
      // At least as long as t | c, so that the dense samples behind it — written WHILE the final spline is evaluated — never
      // lie on the knots and coefficients being read (NK = 256: t | c are 774 doubles, the curvature 128 / 200)
      double curv[(LEAN_ || 3 * (NK + 2) > DCAP) ? 3 * (NK + 2) : DCAP];
      double dxyu[DENSE_ARRAYS * DCAP];  // dense samples x | y of the final spline (third array: more room for the segment
                                         // lengths before fit #3); in the extension the tail points of the polyline
    };
  };
  static constexpr bool BAND_GLOBAL = false;
  static constexpr bool RECOMPUTE = false;  // basis values travel through the BasisCache records
  __device__ __forceinline__ double& FPI(int i) { return fpint[i]; }
  __device__ __forceinline__ double& FPI_W(int i) { return fpint[i]; }  // (where the residual pass writes it)
  __device__ __forceinline__ void fpint_commit(int) {}
  __device__ __forceinline__ int32_t& NRD(int i) { return nrdata[i]; }
  __device__ __forceinline__ double& A(int i, int j) { return a_[i][j - 1]; }
  __device__ __forceinline__ double& Z(int i) { return z[i]; }
  __device__ __forceinline__ double& Gm(int i, int j) { return g_[i][j - 1]; }
};

// The fit-only workspace of fit_kernel: 1160 bytes at 16 knots, so that sixteen frames of a wavefront take 18.6 KB and two
// wavefronts fit a SIMD's share of the LDS.  The band triangle and its right-hand sides are not here: during an
// observation pass they live in the registers of the Givens quad, between passes in the frame's scratch (`band`:
// (NK + 2) x 4 rows, then 2 (NK + 2) right-hand sides), where back-substitution and the smoothing iteration fetch them.
#ifndef FSDP_FIT_CH8
#define FSDP_FIT_CH8 16
#endif
template <int G, int NKC>
struct FitWS {
  static constexpr int GRP = G;
  static constexpr int NK = NKC;
#ifndef FSDP_FIT_CH4
#define FSDP_FIT_CH4 4
#endif
  static constexpr int CH = (G >= 32) ? G : (G >= 16 ? 32 : (G >= 8 ? FSDP_FIT_CH8 : FSDP_FIT_CH4));
  static constexpr bool BAND_GLOBAL = true;
  // 1-based arrays sized for what a fit of NK knots touches: t(1..n), c(1..2n), nrdata(1..n), and the rows 1..n-4 of the
  // extended triangle.  At four lanes per frame (CH = 4) a frame is 928 bytes: sixteen of them are 14 848 bytes, eleven
  // workgroups per CU.
  double t[NK + 1];
  double c[2 * NK + 1];
  double* band;             // the frame's scratch: band triangle | right-hand sides | fpint(1..n)
  int16_t nrdata[NK + 4];   // data points inside a knot interval (< PATH_CAP); lives across the passes of a fit
  union {
    struct {
      double hq[CH][4];
      double xq[CH], yq[CH];
      int32_t lq[CH];
      // refined reciprocals of the knot differences t(a + j) - t(a), j = 1..3, of the current knot set (knot_reciprocals):
      // what the six divisions of a point's basis values (fpbspl3_rd) divide by — a = l-2 .. l for a point in knot
      // interval l = 4 .. n-4, so rows a = RD_A0 .. n-4 are kept.  The rows of g_ overwrite its head in the smoothing
      // iteration, which therefore rebuilds it before every f(p) pass.
      double rd[3 * (NK - 5)];
    };
    double g_[NK - 4][5];
  };
  static constexpr int RD_A0 = 2;  // first row of rd
  // The basis values of a data point are computed again in every pass over the data (observation, residual, f(p)) from
  // its parameter value and the reciprocal table instead of being written to / read from the frame's scratch (BasisCache
  // records): 25 instead of 49 bytes per point and pass through HBM — the fit kernel was bound by that stream
  // (profiles/r04_fit_memory_bound.txt) — for ~45 more FP64 instructions per point and pass.
  static constexpr bool RECOMPUTE = true;
  // fpint (residual sum per knot interval: written once per interval by the residual pass, read and updated by the knot
  // selection that follows it) has no bytes of its own: the residual pass — whose term / flag buffers are hq | xq | yq —
  // writes it behind the band in the frame's scratch, fpint_commit() fetches it into the then dead chunk buffers.
  __device__ __forceinline__ double& FPI_W(int i) { return band[6 * (NK + 2) + i]; }
  __device__ __forceinline__ double& FPI(int i) { return (&hq[0][0])[i]; }
  __device__ __forceinline__ void fpint_commit(int nrint) {  // (after a group sync that follows the last FPI_W)
    static_assert(sizeof(hq) + sizeof(xq) + sizeof(yq) + sizeof(lq) >= sizeof(double) * (NK + 1), "fpint over the chunk buffers");
    for (int i = 1 + Grp<G>::lane(); i <= nrint; i += G) FPI(i) = FPI_W(i);
    Grp<G>::sync();
  }
  __device__ __forceinline__ int16_t& NRD(int i) { return nrdata[i]; }
  __device__ __forceinline__ double& A(int i, int j) { return band[4 * i + j - 1]; }
  __device__ __forceinline__ double& Z(int i) { return band[4 * (NK + 2) + i]; }
  __device__ __forceinline__ double& Gm(int i, int j) { return g_[i - 1][j - 1]; }
};

// Per-point basis cache in the frame's HBM/L2 scratch: one 32-byte record per data point — the K+1 non-zero B-spline
// values — and one byte per point for its knot interval, written by the observation pass of the current knot set and
// re-read by the residual pass and by every f(p) evaluation of the smoothing iteration (same knots => same values; saves
// the interval search and the de Boor recursion with its six divisions per point and pass).  A lane moves a record with
// two 16-byte accesses, the lanes of a group touch consecutive records; the data point itself is read from the polyline
// arrays (8 bytes per lane and coordinate, consecutive lanes on consecutive doubles).  49 bytes per point and pass
// instead of the 64-byte records {h, x, y, interval} of the earlier layout: the refit streams ~10 such passes over 480
// points per frame, and with the chip full of fit wavefronts that stream is a few TB/s.
struct alignas(16) D2 {
  double a, b;
};
struct alignas(32) BRec {
  D2 h01, h23;  // basis values h[0..3] (unused entries of lower degrees are 0)
};
static_assert(sizeof(BRec) == 32, "basis record");
struct BasisCache {
  BRec* rec;
  uint8_t* l;  // knot interval per data point (<= NK_BIG - 4)
  double* b;   // (NK + 2) x 5 rows of the smoothness matrix (fpdisc) of the running fit, row-major, element (i, j) at 5 i + j - 1
};

struct SplineFit {
  int k, n, ier;
  double fp;
  int status;  // 0 ok, 1 = scipy would raise ValueError, ST_OVERFLOW_KNOTS
};

// fpbspl: (k+1) non-zero B-splines at t(l) <= x < t(l+1); h is 1-based [1..k+1]
template <int K>
__device__ __forceinline__ void fpbspl(const double* t, double x, int l, double* h) {
  double hh[K + 2];
  h[1] = 1.0;
#pragma unroll
  for (int j = 1; j <= K; j++) {
#pragma unroll
    for (int i = 1; i <= j; i++) hh[i] = h[i];
    h[1] = 0.0;
#pragma unroll
    for (int i = 1; i <= j; i++) {
      int li = l + i;
      int lj = li - j;
      if (t[li] == t[lj]) {
        h[i + 1] = 0.0;
        continue;
      }
      double f = hh[i] / (t[li] - t[lj]);
      h[i] = h[i] + f * (t[li] - x);
      h[i + 1] = f * (x - t[lj]);
    }
  }
}

__device__ __forceinline__ void fpgivs(double piv, double& ww, double& cs, double& sn) {
  // fpgivs.f: dd = |piv| * sqrt(1 + (ww/piv)^2) if |piv| >= ww else ww * sqrt(1 + (piv/ww)^2) — written with selects so
  // that one division and one square root are issued (same operations, same operands, same bits)
  double store = fabs(piv);
  bool big = store >= ww;
  double num = big ? ww : piv, den = big ? piv : ww, scale = big ? store : ww;
  double r = num / den;
  double dd = scale * sqrt(1.0 + r * r);
  cs = ww / dd;
  sn = piv / dd;
  ww = dd;
}

__device__ __forceinline__ void fprota(double cs, double sn, double& a, double& b) {
  double stor1 = a, stor2 = b;
  b = cs * stor2 + sn * stor1;
  a = cs * stor1 - sn * stor2;
}

// Sequential (data-order) accumulations over an LDS chunk buffer.  The operands are fetched eight at a time so the LDS
// round trip is paid once per eight elements rather than once per element; the additions keep their order.
__device__ __forceinline__ double seq_sum(const double* buf, int cnt, double acc) {
  int r = 0;
  for (; r + 8 <= cnt; r += 8) {
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) v[q] = buf[r + q];
#pragma unroll
    for (int q = 0; q < 8; q++) acc = acc + v[q];
  }
  for (; r < cnt; r++) acc = acc + buf[r];
  return acc;
}
// acc = acc + x[r]^2; acc = acc + y[r]^2 for r = 0..cnt-1
__device__ __forceinline__ double seq_sum_squares2(const double* x, const double* y, int cnt, double acc) {
  int r = 0;
  for (; r + 4 <= cnt; r += 4) {
    double v[4], w[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      v[q] = x[r + q];
      w[q] = y[r + q];
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      acc = acc + v[q] * v[q];
      acc = acc + w[q] * w[q];
    }
  }
  for (; r < cnt; r++) {
    double v = x[r], w = y[r];
    acc = acc + v * v;
    acc = acc + w * w;
  }
  return acc;
}

// knot interval of x: largest l in [k1, nk1] with t(l) <= x (FITPACK's forward search)
__device__ __forceinline__ int find_interval(const double* t, int k1, int nk1, double x) {
  int l = k1;
  while (!(x < t[l + 1] || l == nk1)) l++;
  return l;
}

// the same search started at a known lower bound (t[lstart] <= x): data are processed in increasing order, so a lane
// resumes at its previous interval instead of rescanning the knot vector
__device__ __forceinline__ int find_interval_from(const double* t, int lstart, int nk1, double x) {
  int l = lstart;
  while (!(x < t[l + 1] || l == nk1)) l++;
  return l;
}

// back-substitution, band width k (fpback); el(i, j) = element j (1-based) of band row i
template <class EL>
__device__ __forceinline__ void fpback(EL el, const double* z, int n, int k, double* c) {
  int k1 = k - 1;
  c[n] = z[n] / el(n, 1);
  int i = n - 1;
  if (i == 0) return;
  for (int j = 2; j <= n; j++) {
    double store = z[i];
    int i1 = k1;
    if (j <= k1) i1 = j - 1;
    int m = i;
    for (int l = 1; l <= i1; l++) {
      m = m + 1;
      store = store - c[m] * el(i, l + 1);
    }
    c[i] = store / el(i, 1);
    i = i - 1;
  }
}

// ---- exact division without the range scaling ---------------------------------------------------------
// An IEEE double division on gfx950 is a software sequence: v_div_scale (x2), v_rcp_f64, two Newton steps, a quotient
// with one correction (v_div_fmas) and v_div_fixup.  The scaling and the fix-up only act when an exponent sits near the
// limits of the format; for operands in a safe band the sequence below is the same arithmetic on the same operands and
// returns the same (correctly rounded) bits with 8 instead of 11 instructions — and two quotients over one denominator
// share the refined reciprocal (11 instead of 22).  The guard: callers flag operands outside [2^-255, 2^255]
// (float compares on the operands; the knot differences once per knot set) and such a frame is re-planned with plain
// divisions (ST_RETRY, path_kernel.h).
// the exponent band the guards of the scaling-free division accept (what fsdp_selftest_math checks the sequence on)
__device__ __forceinline__ bool in_div_band(double v) { return fabs(v) >= 0x1p-255 && fabs(v) <= 0x1p255; }
// max(|a|, b) / min(|a|, b) of numbers that are never NaN: one v_max_f64 / v_min_f64 each (the absolute value is a source
// modifier; fmax() would first quiet both operands)
__device__ __forceinline__ double max_abs_nn(double a, double b) {
#ifdef FSDP_EMU
  return fabs(a) >= b ? fabs(a) : b;
#else
  double r;
  asm("v_max_f64 %0, |%1|, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
#endif
}
__device__ __forceinline__ double min_abs_nn(double a, double b) {
#ifdef FSDP_EMU
  return fabs(a) >= b ? b : fabs(a);
#else
  double r;
  asm("v_min_f64 %0, |%1|, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
#endif
}
__device__ __forceinline__ double rcp_refined(double d) {
#ifdef FSDP_EMU
  return d;  // (the emulator divides directly, see div_rcp)
#else
  double r = __builtin_amdgcn_rcp(d);
  double e = fma(-d, r, 1.0);
  r = fma(r, e, r);
  e = fma(-d, r, 1.0);
  r = fma(r, e, r);
  return r;
#endif
}
// n / d given r = rcp_refined(d)
__device__ __forceinline__ double div_rcp(double n, double d, double r) {
#ifdef FSDP_EMU
  (void)r;
  return n / d;
#else
  const double q = n * r;
  const double rem = fma(-d, q, n);
  return fma(rem, r, q);
#endif
}

// sqrt for arguments in [1, 2] (1 + r^2 with |r| <= 1): the correctly rounded result, i.e. what sqrt() returns; on the
// device this is the compiler's own v_rsq_f64 + Goldschmidt sequence without the range scaling that [1, 2] never needs
// (checked against sqrt() on the GPU: tests/test_gpu_parity.py::test_device_math_helpers)
__device__ __forceinline__ double sqrt_1_2(double x) {
#ifdef FSDP_EMU
  return sqrt(x);
#else
  double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = y * 0.5;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  double d = fma(-g, g, x);
  g = fma(d, h, g);
  d = fma(-g, g, x);
  g = fma(d, h, g);
  return g;
#endif
}

// The middle of fpgivs for operands in the divisions' safe band: den = max(|piv|, ww), num = min(|piv|, ww) ->
//   dd = den * sqrt(1 + (num / den)^2)   and   rd = a refined reciprocal of dd for the two quotients cs = ww / dd, sn = piv / dd.
// The reciprocal is the head of the second half of the step's dependent chain (rcp_refined(dd): v_rcp_f64 and two Newton steps, five
// links after dd is known).  Its seed need not wait for dd: 1 / dd = (1 / den) * (1 / sqrt(x)), and both factors exist while the square
// root is still being corrected — rq = rcp_refined(den) from the first quotient, and h, the half reciprocal square root the
// Goldschmidt iteration refines next to g (relative error ~2^-45 after its coupled step, what v_rcp_f64 + ONE Newton step gives).
// r0 = (2 rq) h is formed in the shadow of sqrt's last two corrections, and ONE Newton step against dd itself (error^2 ~ 2^-90, then
// the rounding of the fma) makes it the reciprocal rcp_refined returns for all the quotients care: two links after dd instead of
// five, three instructions less per step.  The quotients are div_rcp's (product, exact remainder, correction): correctly rounded
// with either reciprocal (fsdp_selftest_givens holds cs / sn / dd against the IEEE operations on the device: tests/test_gpu_parity.py).
// FSDP_GIVENS_RSQ_SEED=0 (A/B builds): the reciprocal from v_rcp_f64 again.
#ifndef FSDP_GIVENS_RSQ_SEED
#define FSDP_GIVENS_RSQ_SEED 1
#endif
__device__ __forceinline__ void givens_dd_rd(double den, double num, double& dd, double& rd) {
#ifdef FSDP_EMU
  const double q = num / den;
  dd = den * sqrt(1.0 + q * q);
  rd = dd;  // (the emulator's div_rcp divides directly)
#else
  const double rq = rcp_refined(den);
  const double q = div_rcp(num, den, rq);
#if FSDP_GIVENS_RSQ_SEED
  const double x = 1.0 + q * q;
  // sqrt_1_2(x), keeping h
  double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = y * 0.5;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  const double r0 = (rq + rq) * h;  // ~ 1 / (den sqrt(x)), off the chain
  double d = fma(-g, g, x);
  g = fma(d, h, g);
  d = fma(-g, g, x);
  g = fma(d, h, g);
  dd = den * g;
  const double e = fma(-dd, r0, 1.0);
  rd = fma(r0, e, r0);
#else
  dd = den * sqrt_1_2(1.0 + q * q);
  rd = rcp_refined(dd);
#endif
#endif
}

// fpgivs with the scaling-free divisions (the arithmetic of giv_step<true>: max / min instead of the branch, two refined reciprocals,
// sqrt_1_2) for the rotations of the smoothing rows; operands outside the divisions' exponent band set `bad` (the frame is then
// planned again with plain divisions).  FAST = false: fpgivs itself.
template <bool FAST>
__device__ __forceinline__ void fpgivs_guarded(double piv, double& ww, double& cs, double& sn, int& bad) {
  if constexpr (FAST) {
    const double w = ww;
    const double den = max_abs_nn(piv, w), num = min_abs_nn(piv, w);
    bad |= (int)!((den >= 0x1p-255) & (den <= 0x1p+255) & ((num == 0.0) | (num >= 0x1p-255)));
    double dd, rd;
    givens_dd_rd(den, num, dd, rd);
    cs = div_rcp(w, dd, rd);
    sn = div_rcp(piv, dd, rd);
    ww = dd;
  } else {
    fpgivs(piv, ww, cs, sn);
  }
}

// fpbspl for degree 3 as straight-line code: the six knots around the interval are fetched together (one LDS round
// trip instead of a dependent read per term), coincident knots are handled by selects (the quotient of the skipped
// branch is computed and dropped), and with FAST the six divisions use the exact scaled-free sequence above (operands
// outside its exponent band set `bad`).  Same operations on the same operands as fpbspl<3>: same bits.
template <bool FAST>
__device__ __forceinline__ void fpbspl3(const double* t, double x, int l, double* h /*[0..3]*/, int& bad) {
  const double tm2 = t[l - 2], tm1 = t[l - 1], t0 = t[l], tp1 = t[l + 1], tp2 = t[l + 2], tp3 = t[l + 3];
  // FAST: the denominators are differences of knots — their exponent band is checked once per knot set by the caller
  // (knot_differences_safe) — so only the numerators (partial basis values: 0 or in [2^-255, 2^255]) are checked per point
  bool ok = true;
  auto quot = [&](double num, double den) {
    if constexpr (FAST) {
      ok = ok & ((num == 0.0) | ((num >= 0x1p-255) & (num <= 0x1p255)));
      return div_rcp(num, den, rcp_refined(den));
    } else {
      return num / den;
    }
  };
  // level 1
  double h1, h2, h3, h4;
  {
    const double den = tp1 - t0;
    const bool same = tp1 == t0;
    const double f = quot(1.0, den);
    h1 = same ? 0.0 : 0.0 + f * (tp1 - x);
    h2 = same ? 0.0 : f * (x - t0);
  }
  // level 2
  {
    const double a1 = h1, a2 = h2;
    h1 = 0.0;
    {
      const double den = tp1 - tm1;
      const bool same = tp1 == tm1;
      const double f = quot(a1, den);
      h1 = same ? h1 : h1 + f * (tp1 - x);
      h2 = same ? 0.0 : f * (x - tm1);
    }
    {
      const double den = tp2 - t0;
      const bool same = tp2 == t0;
      const double f = quot(a2, den);
      h2 = same ? h2 : h2 + f * (tp2 - x);
      h3 = same ? 0.0 : f * (x - t0);
    }
  }
  // level 3
  {
    const double a1 = h1, a2 = h2, a3 = h3;
    h1 = 0.0;
    {
      const double den = tp1 - tm2;
      const bool same = tp1 == tm2;
      const double f = quot(a1, den);
      h1 = same ? h1 : h1 + f * (tp1 - x);
      h2 = same ? 0.0 : f * (x - tm2);
    }
    {
      const double den = tp2 - tm1;
      const bool same = tp2 == tm1;
      const double f = quot(a2, den);
      h2 = same ? h2 : h2 + f * (tp2 - x);
      h3 = same ? 0.0 : f * (x - tm1);
    }
    {
      const double den = tp3 - t0;
      const bool same = tp3 == t0;
      const double f = quot(a3, den);
      h3 = same ? h3 : h3 + f * (tp3 - x);
      h4 = same ? 0.0 : f * (x - t0);
    }
  }
  h[0] = h1;
  h[1] = h2;
  h[2] = h3;
  h[3] = h4;
  if constexpr (FAST) bad |= (int)!ok;
}

// The denominators of fpbspl3 are t(a + j) - t(a), j = 1..3: every one of the current knot vector t(1..n) must be 0
// (coincident knots: that term is skipped) or inside the exponent band of the scaling-free division.  One lane per knot.
template <int G>
__device__ __forceinline__ bool knot_differences_safe(const double* t, int n) {
  bool badk = false;
  for (int a = 1 + Grp<G>::lane(); a <= n; a += G) {
#pragma unroll
    for (int j = 1; j <= 3; j++) {
      const int b = a + j <= n ? a + j : n;
      const double d = t[b] - t[a];
      badk |= !((d == 0.0) | ((d >= 0x1p-255) & (d <= 0x1p255)));
    }
  }
  return Grp<G>::ballot(badk) == 0ull;
}

// ---- basis values from the reciprocal table (workspaces with RECOMPUTE) ---------------------------------------
// rd(a, j) = rcp_refined(t(a + j) - t(a)) at rd[3 a + j - 1].  The six denominators of a point in knot interval l are
// t(l+1)-t(l), t(l+1)-t(l-1), t(l+2)-t(l), t(l+1)-t(l-2), t(l+2)-t(l-1), t(l+3)-t(l): each spans [t(l), t(l+1)], which
// is not empty for an interval that holds a data point — so none of them is zero and fpbspl's "coincident knots" branch
// (the selects of fpbspl3) is never taken; knot_reciprocals checks exactly that once per knot set, and that every one of
// them lies in the exponent band of the scaling-free division.
template <int G, int A0 = 1>  // A0: first row the table keeps (rows A0 .. n-4 at rd[3 (a - A0) + j - 1])
__device__ __forceinline__ bool knot_reciprocals(const double* t, int n, double* rd) {
  bool badk = false;
  for (int idx = Grp<G>::lane(); idx < 3 * n; idx += G) {
    const int a = 1 + idx / 3, j = 1 + idx % 3;
    const int b = a + j <= n ? a + j : n;
    const double d = t[b] - t[a];
    const bool zero = d == 0.0;
    badk |= !(zero | ((d >= 0x1p-255) & (d <= 0x1p255)));
    // interior knot intervals must not be empty: a = k1 .. nk1 = 4 .. n - 4 (then all six spans of fpbspl3_rd are > 0)
    if (j == 1 && a >= 4 && a <= n - 4) badk |= zero;
    // (fpbspl3_rd reads rows l-2 .. l of a point in knot interval l = 4 .. n-4)
    if (a >= A0 && a <= n - 4) rd[3 * (a - A0) + j - 1] = zero ? 0.0 : rcp_refined(d);
  }
  return Grp<G>::ballot(badk) == 0ull;
}

// fpbspl3 without the coincident-knot selects and with the denominators' reciprocals from the table: the operations on
// the operands of fpbspl3<true> in its order — same bits (the quotients are div_rcp(num, den, rcp_refined(den))).
// CHECK: the numerators' exponent band (once per point and knot set is enough: the observation pass).
template <bool CHECK, int A0 = 1>
__device__ __forceinline__ void fpbspl3_rd(const double* t, const double* rdt, double x, int l, double* h /*[0..3]*/, int& bad) {
  const double tm2 = t[l - 2], tm1 = t[l - 1], t0 = t[l], tp1 = t[l + 1], tp2 = t[l + 2], tp3 = t[l + 3];
  const double* const rd = rdt + 3 * (l - A0) - 1;  // rd[3 (a - l) + j] = row a, span j
  const double r01 = rd[1], r02 = rd[2], r03 = rd[3];  // t(l+1..3) - t(l)
  const double rm12 = rd[-3 + 2], rm13 = rd[-3 + 3];   // t(l+1..2) - t(l-1)
  const double rm23 = rd[-6 + 3];                      // t(l+1) - t(l-2)
  bool ok = true;
  auto quot = [&](double num, double den, double r) {
    if constexpr (CHECK) ok = ok & ((num == 0.0) | ((num >= 0x1p-255) & (num <= 0x1p255)));
    return div_rcp(num, den, r);
  };
  double h1, h2, h3, h4;
  {
    const double f = quot(1.0, tp1 - t0, r01);
    h1 = 0.0 + f * (tp1 - x);
    h2 = f * (x - t0);
  }
  {
    const double a1 = h1, a2 = h2;
    h1 = 0.0;
    {
      const double f = quot(a1, tp1 - tm1, rm12);
      h1 = h1 + f * (tp1 - x);
      h2 = f * (x - tm1);
    }
    {
      const double f = quot(a2, tp2 - t0, r02);
      h2 = h2 + f * (tp2 - x);
      h3 = f * (x - t0);
    }
  }
  {
    const double a1 = h1, a2 = h2, a3 = h3;
    h1 = 0.0;
    {
      const double f = quot(a1, tp1 - tm2, rm23);
      h1 = h1 + f * (tp1 - x);
      h2 = f * (x - tm2);
    }
    {
      const double f = quot(a2, tp2 - tm1, rm13);
      h2 = h2 + f * (tp2 - x);
      h3 = f * (x - tm1);
    }
    {
      const double f = quot(a3, tp3 - t0, r03);
      h3 = h3 + f * (tp3 - x);
      h4 = f * (x - t0);
    }
  }
  h[0] = h1;
  h[1] = h2;
  h[2] = h3;
  h[3] = h4;
  if constexpr (CHECK) bad |= (int)!ok;
}

// ---- 4-stage systolic Givens pipeline (degree 3) ------------------------------------------------------
// A data row of knot interval l touches the 4 consecutive band rows l-3..l, one rotation each, in that order;
// consecutive data rows fall (almost always) into the same interval.  Lane p of the group's first quad owns the band
// row j with j mod 4 = p and keeps it in registers; a data row travels lane -> lane (quad rotate, DPP) one stage per
// step, so up to 4 data rows are in flight.  Every band row still sees the data rows in data order and every data row
// still visits its band rows in order: the arithmetic (fpgivs / fprota) and its sequence per element are exactly
// FITPACK's, only independent rotations overlap in time.
//
// The rows of one knot interval form a *run*: inside a run a lane's stage is fixed (lane - (l-3) mod 4), a row's
// validity follows from the step counter, and the sum of squared rotated-out right-hand sides (fp, accumulated in data
// order) stays in the stage-4 lane.  When the interval changes (a handful of times per pass: intervals <= knots) the
// three rows in flight are drained, fp moves to the new stage-4 lane, the lane whose band row falls out of the window
// writes it back to LDS and starts the next one from zero.  A run may pause between the chunks of an observation pass
// (rows in flight wait in registers).
//
// Entries of a data row beyond its band (and of a band row beyond the interval reached so far) are exact zeros and rotate
// to exact zeros, and a step that must not rotate (no row at this stage yet, or pivot 0: fpgivs is skipped) runs the
// rotation with cs = 1, sn = 0, which returns its finite inputs unchanged — so the step has no per-element selects.
struct GivLane {
  double a1, a2, a3, a4, z1, z2;       // the band row this lane owns
  int j;                                // its index (0 = none yet)
  double o_piv, o_r0, o_r1, o_x1, o_x2; // data row leaving this lane (rotated to the next lane at the next step)
  double fp;                            // fp as accumulated by the stage-4 lane of the current run (uniform between runs)
  int l;                                // knot interval of the current run (0 = no run)
  int stage;                            // this lane's stage (1..4) in the current run
  int t;                                // steps of the current run so far
  int fed;                              // rows fed into the current run so far
  int bad;                              // FAST division met an operand outside the safe exponent band
};

__device__ __forceinline__ void giv_init(GivLane& st) {
  st.a1 = st.a2 = st.a3 = st.a4 = st.z1 = st.z2 = 0.0;
  st.j = 0;
  st.o_piv = st.o_r0 = st.o_r1 = st.o_x1 = st.o_x2 = 0.0;
  st.fp = 0.0;
  st.l = 0;
  st.stage = 1;
  st.t = 0;
  st.fed = 0;
  st.bad = 0;
}

template <class WS>
__device__ __forceinline__ void giv_flush(WS& ws, const GivLane& st, int lane, int n) {
  if (lane < 4 && st.j > 0) {
    ws.A(st.j, 1) = st.a1;
    ws.A(st.j, 2) = st.a2;
    ws.A(st.j, 3) = st.a3;
    ws.A(st.j, 4) = st.a4;
    ws.Z(st.j) = st.z1;
    ws.Z(st.j + n) = st.z2;
  }
}

__device__ __forceinline__ double quad_prev(double d) {  // value of the previous lane of the quad (lane 0 <- lane 3)
#ifdef FSDP_EMU
  int l = emu::B->cur;
  return emu::gexchange(d, (l & ~3) | ((l + 3) & 3), 4);
#else
  int lo = __double2loint(d), hi = __double2hiint(d);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x93, 0xf, 0xf, true);  // quad_perm:[3,0,1,2]
  hi = __builtin_amdgcn_mov_dpp(hi, 0x93, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
#endif
}

// One pipeline step on every lane.  feed: the stage-1 lane takes the data row (h0..h3, x, y); otherwise nothing enters.
template <bool FAST>
__device__ __forceinline__ void giv_step(GivLane& st, bool feed, double h0, double h1, double h2, double h3, double dx, double dy) {
  double piv = quad_prev(st.o_piv), r0 = quad_prev(st.o_r0), r1 = quad_prev(st.o_r1), x1 = quad_prev(st.o_x1),
         x2 = quad_prev(st.o_x2);
  double r2 = 0.0;
  const bool s1 = st.stage == 1;
  if (s1 && feed) {
    piv = h0;
    r0 = h1;
    r1 = h2;
    r2 = h3;
    x1 = dx;
    x2 = dy;
  }
  const int idx = st.t - (st.stage - 1);
  const bool valid = (unsigned)idx < (unsigned)st.fed;  // 0 <= idx < fed
  const bool rot = valid && piv != 0.0;
  // fpgivs.f: dd = |piv| * sqrt(1 + (ww/piv)^2) if |piv| >= ww else ww * sqrt(1 + (piv/ww)^2) — written with selects so
  // that one division and one square root are issued (same operations, same operands, same bits)
  // The quotient only enters as its square, and |ww / piv| = ww / |piv| bit for bit (IEEE division is sign-symmetric), so
  // the two branches are den = max(|piv|, ww) = scale, num = min(|piv|, ww) (ww >= 0): two instructions instead of a
  // compare and six selects.
  const double ww = st.a1;
  const double den = max_abs_nn(piv, ww), num = min_abs_nn(piv, ww), scale = den;
  double dd, cs, sn;
  if constexpr (FAST) {
    // operands of the scaling-free divisions: den (and dd in [den, sqrt(2) den]) inside the safe band, num inside it or 0
    st.bad |= (int)(rot & !((den >= 0x1p-255) & (den <= 0x1p+255) & ((num == 0.0) | (num >= 0x1p-255))));
    // (Tried in round 5 and not kept: den is the diagonal itself unless the pivot is larger, and the diagonal's refined reciprocal
    // is the rd of the step that formed it — same bits, five instructions and a reciprocal's latency less.  The wave-uniform branch
    // around the rare other case cost more than that: p50 894 -> 924 us, 7.00 -> 6.87 M frames/s; profiles/r05_givens_step.txt.)
    double rd;
    givens_dd_rd(scale, num, dd, rd);
    cs = div_rcp(ww, dd, rd);
    sn = div_rcp(piv, dd, rd);
  } else {
    const double q = num / den;
    dd = scale * sqrt(1.0 + q * q);
    cs = ww / dd;
    sn = piv / dd;
  }
  cs = rot ? cs : 1.0;
  sn = rot ? sn : 0.0;
  st.a1 = rot ? dd : st.a1;
  // fprota on the right-hand sides and the rest of the band row
  const double nz1 = cs * st.z1 + sn * x1, nx1 = cs * x1 - sn * st.z1;
  const double nz2 = cs * st.z2 + sn * x2, nx2 = cs * x2 - sn * st.z2;
  const double na2 = cs * st.a2 + sn * r0, nr0 = cs * r0 - sn * st.a2;
  const double na3 = cs * st.a3 + sn * r1, nr1 = cs * r1 - sn * st.a3;
  const double na4 = cs * st.a4 + sn * r2, nr2 = cs * r2 - sn * st.a4;
  st.z1 = nz1;
  st.z2 = nz2;
  st.a2 = na2;
  st.a3 = na3;
  st.a4 = na4;
  // retire at stage 4: the rotated-out right-hand sides enter fp in data order
  {
    double f = st.fp + nx1 * nx1;
    f = f + nx2 * nx2;
    st.fp = (valid && st.stage == 4) ? f : st.fp;
  }
  st.o_piv = nr0;
  st.o_r0 = nr1;
  st.o_r1 = nr2;
  st.o_x1 = nx1;
  st.o_x2 = nx2;
  st.t++;
}

// drain the rows in flight and close the run: fp becomes uniform in the group again
template <int G, bool FAST>
__device__ __forceinline__ void giv_end_run(GivLane& st) {
  if (st.l == 0) return;
  for (int q = 0; q < 3; q++) giv_step<FAST>(st, false, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0);
  st.fp = Grp<G>::bcast(st.fp, st.l & 3);  // the stage-4 lane of interval l: band row l, lane l mod 4
  st.l = 0;
}

// open the run of knot interval l: stages follow from the lane, the lane whose band row left the window moves on
template <class WS>
__device__ __forceinline__ void giv_begin_run(WS& ws, GivLane& st, int l, int lane, int n) {
  const int j0 = l - 3;
  st.stage = ((lane - j0) & 3) + 1;
  const int my_j = j0 + st.stage - 1;
  if (st.j != my_j) {
    // the old band row is complete; the new one has not been touched yet in this pass (rows arrive in increasing
    // interval order), i.e. it is still all zero
    giv_flush(ws, st, lane, n);
    st.j = my_j;
    st.a1 = st.a2 = st.a3 = st.a4 = st.z1 = st.z2 = 0.0;
  }
  st.l = l;
  st.t = 0;
  st.fed = 0;
}

// rows [r0, r1) of the chunk buffers (all of the run's interval) into the pipeline; rows still in flight at the end stay
// in `st`.  Two rows per loop trip: a row's registers are refilled (group-uniform LDS reads) right after the step that
// consumed them, so the read flies during the other row's step and the loop carries no register copies.
template <bool FAST, class WS>
__device__ __forceinline__ void giv_feed(WS& ws, GivLane& st, int r0, int r1) {
  st.fed += r1 - r0;
  struct Row {
    double h0, h1, h2, h3, x, y;
  };
  auto load = [&](int r) {
    const int q = r < r1 ? r : r1 - 1;
    return Row{ws.hq[q][0], ws.hq[q][1], ws.hq[q][2], ws.hq[q][3], ws.xq[q], ws.yq[q]};
  };
  Row a = load(r0), b = load(r0 + 1);
  int r = r0;
  for (; r + 1 < r1; r += 2) {
    giv_step<FAST>(st, true, a.h0, a.h1, a.h2, a.h3, a.x, a.y);
    a = load(r + 2);
    giv_step<FAST>(st, true, b.h0, b.h1, b.h2, b.h3, b.x, b.y);
    b = load(r + 3);
  }
  if (r < r1) giv_step<FAST>(st, true, a.h0, a.h1, a.h2, a.h3, a.x, a.y);
}

// ---- the same pipeline on a 4 x 4 lane grid (groups of 16 lanes and more) ------------------------------------------------
// Round 5.  In the quad form every lane carries a whole step — fpgivs AND five plane rotations — while the other lanes of a 16- or
// 64-lane group idle.  Here the first sixteen lanes of the group form four quads: quad p owns the band rows j with j mod 4 = p (as
// lane p did), and inside a quad lane e holds ONE column pair of that row: e = 0 (a2 | r0), e = 1 (a3 | r1), e = 2 (a4 | r2),
// e = 3 both right-hand sides (z1 | x1), (z2 | x2).  Every lane of a quad keeps the diagonal a1 and receives the pivot, so every lane
// forms cs / sn itself (fpgivs is the critical path anyway) and then rotates its own pair(s): two rotation slots instead of five,
// ~88 instead of 109 instructions per step, same operations on the same operands in the same order per element.  A data row moves
// quad -> quad by a row rotation (DPP row_ror:4); inside the target quad the pivot (element 0 of the row that leaves) goes to every
// lane (quad_perm 0,0,0,0) and every other element one column to the left (quad_perm 1,2,2,3: r0 <- r1, r1 <- r2, r2 <- 0).
struct GivGridLane {
  double a1;      // the band row's diagonal (every lane of the quad)
  double bA, bB;  // this lane's band element(s): e = 0..2: a2 / a3 / a4 (bB = 0); e = 3: z1, z2
  int j;          // the band row the quad owns (0 = none yet)
  double oA, oB;  // this lane's element(s) of the data row leaving the quad
  double fp;      // as GivLane::fp, in the lane e = 3 of the stage-4 quad
  int l, stage, t, fed, bad;
};
__device__ __forceinline__ void giv_init(GivGridLane& st) {
  st.a1 = st.bA = st.bB = 0.0;
  st.j = 0;
  st.oA = st.oB = 0.0;
  st.fp = 0.0;
  st.l = 0;
  st.stage = 1;
  st.t = 0;
  st.fed = 0;
  st.bad = 0;
}
template <class WS>
__device__ __forceinline__ void giv_flush(WS& ws, const GivGridLane& st, int lane, int n) {
  if (lane < 16 && st.j > 0) {
    const int e = lane & 3;
    if (e == 0) {
      ws.A(st.j, 1) = st.a1;
      ws.A(st.j, 2) = st.bA;
    } else if (e == 1) {
      ws.A(st.j, 3) = st.bA;
    } else if (e == 2) {
      ws.A(st.j, 4) = st.bA;
    } else {
      ws.Z(st.j) = st.bA;
      ws.Z(st.j + n) = st.bB;
    }
  }
}
// DPP moves of a double inside a row of 16 lanes.  CTRL: 0x124 = row_ror:4 (lane i <- lane i - 4 mod 16), 0x00 = quad_perm
// [0,0,0,0], 0xE9 = quad_perm [1,2,2,3]
template <int CTRL>
__device__ __forceinline__ double dpp_row16(double d) {
#ifdef FSDP_EMU
  const int l = emu::B->cur, r = l & 15, base = l & ~15;
  int src;
  if (CTRL == 0x124)
    src = base | ((r + 12) & 15);
  else if (CTRL == 0x00)
    src = l & ~3;
  else
    src = (l & ~3) | ((l & 3) == 0 ? 1 : ((l & 3) == 3 ? 3 : 2));
  return emu::gexchange(d, src, 16);
#else
  int lo = __double2loint(d), hi = __double2hiint(d);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
#endif
}
// One pipeline step.  feed: the stage-1 quad takes the data row: h0 = its first element (pivot), fA / fB = this lane's element(s)
// of it (e = 0..2: h1 / h2 / h3, 0; e = 3: x, y).
template <bool FAST>
__device__ __forceinline__ void giv_step(GivGridLane& st, int e, bool feed, double h0, double fA, double fB) {
  const double tA = dpp_row16<0x124>(st.oA), tB = dpp_row16<0x124>(st.oB);
  double piv = dpp_row16<0x00>(tA);
  double eA = dpp_row16<0xE9>(tA);
  eA = (e == 2) ? 0.0 : eA;  // (a row's fourth element is zero behind its first stage)
  double eB = tB;
  const bool take = (st.stage == 1) && feed;
  piv = take ? h0 : piv;
  eA = take ? fA : eA;
  eB = take ? fB : eB;
  const int idx = st.t - (st.stage - 1);
  const bool valid = (unsigned)idx < (unsigned)st.fed;
  const bool rot = valid && piv != 0.0;
  const double ww = st.a1;
  const double den = max_abs_nn(piv, ww), num = min_abs_nn(piv, ww);
  double dd, cs, sn;
  if constexpr (FAST) {
    st.bad |= (int)(rot & !((den >= 0x1p-255) & (den <= 0x1p+255) & ((num == 0.0) | (num >= 0x1p-255))));
    double rd;
    givens_dd_rd(den, num, dd, rd);
    cs = div_rcp(ww, dd, rd);
    sn = div_rcp(piv, dd, rd);
  } else {
    const double q = num / den;
    dd = den * sqrt(1.0 + q * q);
    cs = ww / dd;
    sn = piv / dd;
  }
  cs = rot ? cs : 1.0;
  sn = rot ? sn : 0.0;
  st.a1 = rot ? dd : st.a1;
  const double nbA = cs * st.bA + sn * eA, nA = cs * eA - sn * st.bA;
  const double nbB = cs * st.bB + sn * eB, nB = cs * eB - sn * st.bB;
  st.bA = nbA;
  st.bB = nbB;
  {
    double f = st.fp + nA * nA;
    f = f + nB * nB;
    st.fp = (valid && st.stage == 4 && e == 3) ? f : st.fp;
  }
  st.oA = nA;
  st.oB = nB;
  st.t++;
}
template <int G, bool FAST>
__device__ __forceinline__ void giv_end_run(GivGridLane& st) {
  if (st.l == 0) return;
  const int e = Grp<G>::lane() & 3;
  for (int q = 0; q < 3; q++) giv_step<FAST>(st, e, false, 0.0, 0.0, 0.0);
  st.fp = Grp<G>::bcast(st.fp, 4 * (st.l & 3) + 3);  // lane e = 3 of the quad that owns band row l
  st.l = 0;
}
template <class WS>
__device__ __forceinline__ void giv_begin_run(WS& ws, GivGridLane& st, int l, int lane, int n) {
  const int j0 = l - 3;
  const int p = (lane >> 2) & 3;
  st.stage = ((p - j0) & 3) + 1;
  const int my_j = j0 + st.stage - 1;
  if (st.j != my_j) {
    giv_flush(ws, st, lane, n);
    st.j = my_j;
    st.a1 = st.bA = st.bB = 0.0;
  }
  st.l = l;
  st.t = 0;
  st.fed = 0;
}
template <bool FAST, class WS>
__device__ __forceinline__ void giv_feed(WS& ws, GivGridLane& st, int r0, int r1) {
  st.fed += r1 - r0;
  const int e = Grp<WS::GRP>::lane() & 3;
  // this lane's element of a chunk row: column 1 + e of the basis values, or x (e = 3); y for the second slot of e = 3
  const double* const baseA = (e < 3) ? (&ws.hq[0][0] + 1 + e) : &ws.xq[0];
  const int strideA = (e < 3) ? 4 : 1;
  struct Row {
    double h0, a, b;
  };
  auto load = [&](int r) {
    const int q = r < r1 ? r : r1 - 1;
    const double y = ws.yq[q];
    return Row{ws.hq[q][0], baseA[q * strideA], (e == 3) ? y : 0.0};
  };
  Row a = load(r0), b = load(r0 + 1);
  int r = r0;
  for (; r + 1 < r1; r += 2) {
    giv_step<FAST>(st, e, true, a.h0, a.a, a.b);
    a = load(r + 2);
    giv_step<FAST>(st, e, true, b.h0, b.a, b.b);
    b = load(r + 3);
  }
  if (r < r1) giv_step<FAST>(st, e, true, a.h0, a.a, a.b);
}

template <bool GRID>
struct GivState {
  using type = GivLane;
};
template <>
struct GivState<true> {
  using type = GivGridLane;
};

// Residual terms sum_d (s_d(u_i) - x_d,i)^2 of one "super-chunk" of points [base, base + cnt) (cnt <= 4 * CH), one point
// per lane and round, from the basis cache.  load() issues every scratch load of the super-chunk (registers), compute()
// does the arithmetic and writes the terms to tbuf[0..cnt) (the chunk's basis buffer, idle here) and, when FLAGS, the
// "a new knot interval starts at this point" flags to fbuf.  The caller issues the next super-chunk's load() before
// the serial accumulation of the current one, so the scratch round trip hides behind it.
#ifndef RB_MAX_ROUNDS
#define RB_MAX_ROUNDS 2  // rounds of a residual super-chunk held in registers (2: fit_kernel fits three wavefronts per SIMD)
#endif
template <int K, int G, bool FLAGS, int CHV, bool RC = false>  // RC: basis values from the parameter value and the workspace's reciprocal table
struct ResidualBatch {
  // (4 lanes per frame: fit_kernel<4> runs at two wavefronts per SIMD and has the registers for four rounds in flight)
#ifndef FSDP_RB_G4
#define FSDP_RB_G4 (2 * RB_MAX_ROUNDS)
#endif
  static constexpr int RB_CAP = (G == 4) ? FSDP_RB_G4 : RB_MAX_ROUNDS;
  static constexpr int ROUNDS = (4 * CHV / G > RB_CAP) ? RB_CAP : 4 * CHV / G;
  static constexpr int k1 = K + 1, k2 = K + 2;
  double hv[ROUNDS][RC ? 1 : K + 1], xv[ROUNDS], yv[ROUNDS];
  int lv[ROUNDS], lpv[ROUNDS];

  // Loads are unconditional (rows past the end re-read the last record, index clamped) so that all of a super-chunk's
  // loads sit in one basic block and fly together; only the stores of compute() are predicated.
  __device__ __forceinline__ void load(const BasisCache& bc, const double* U, const double* X, const double* Y, int base, int cnt, int m) {
    (void)cnt;
    const int lane = Grp<G>::lane();
#pragma unroll
    for (int q = 0; q < ROUNDS; q++) {
      int it = base + q * G + lane;
      it = it < m ? it : m - 1;
      if constexpr (RC) {
        hv[q][0] = U[it];  // (the parameter value; compute() turns it into the basis values)
      } else {
        const BRec* p = &bc.rec[it];
        const D2 a = p->h01, b = p->h23;
        const double hh[4] = {a.a, a.b, b.a, b.b};
#pragma unroll
        for (int j = 0; j < k1; j++) hv[q][j] = hh[j];
      }
      // FITPACK tracks l sequentially (one step per data point); with knots at data points this is
      // l = k2 + #{interior knots <= u(it)} = (interval of u(it)) + 1, "new" when it grew at this point
      lv[q] = (int)bc.l[it] + 1;
      if constexpr (FLAGS) {
        const int lp = (int)bc.l[it > 0 ? it - 1 : 0] + 1;
        lpv[q] = it > 0 ? lp : k2;
      }
      xv[q] = X[it];
      yv[q] = Y[it];
#ifdef FSDP_PAD_LOADS
      {  // experiment: 32 more bytes per point and pass from lines nobody else touches (is the kernel bound by its scratch stream?)
        const D2 e0 = bc.rec[it + 704].h01, e1 = bc.rec[it + 704].h23;
        asm volatile("" ::"v"(e0.a), "v"(e0.b), "v"(e1.a), "v"(e1.b));
      }
#endif
    }
  }

  template <class WS>
  __device__ __forceinline__ void compute(WS& ws, int cnt, int n, double* tbuf, int32_t* fbuf) const {
    const int lane = Grp<G>::lane();
#pragma unroll
    for (int q = 0; q < ROUNDS; q++) {
      const int r = q * G + lane;
      const int l0 = lv[q] - k2;
      double hb[K + 1];
      if constexpr (RC) {
        int unused = 0;
        fpbspl3_rd<false, WS::RD_A0>(ws.t, ws.rd, hv[q][0], lv[q] - 1, hb, unused);  // (lv = FITPACK's l = interval + 1)
      } else {
#pragma unroll
        for (int j = 0; j < k1; j++) hb[j] = hv[q][j];
      }
      double term = 0.0;
#pragma unroll
      for (int d = 0; d < 2; d++) {
        double fac = 0.0;
        int j1 = l0 + d * n;
#pragma unroll
        for (int j = 1; j <= k1; j++) {
          j1++;
          fac = fac + ws.c[j1] * hb[j - 1];
        }
        double dv = 1.0 * (fac - (d == 0 ? xv[q] : yv[q]));  // w = 1
        term = term + dv * dv;
      }
      if (r < cnt) {
        tbuf[r] = term;
        if constexpr (FLAGS) fbuf[r] = (lv[q] > lpv[q]) ? 1 : 0;
      }
    }
  }
};

// parcur/fppara for idim=2, w=1, iopt=0.  Data (0-based arrays U = parameter, X, Y; m points) in LDS or HBM.
// All lanes of the group call; result (t, c) left in ws; returns the group-uniform SplineFit.
template <int K, bool FAST, class WS>
__device__ __forceinline__ SplineFit spline_fit_k(WS& ws, const BasisCache& bc, const double* U, const double* X, const double* Y,
                                         int m, double s) {
  constexpr int G = WS::GRP;
  using GR = Grp<G>;
  constexpr int CH = WS::CH;
  constexpr int SC = ResidualBatch<K, G, false, CH>::ROUNDS * G;  // points per residual half "super-chunk"
  constexpr int HALVES = (2 * SC <= 4 * CH) ? 2 : 1;             // the chunk's basis buffer holds 4 * CH terms
  constexpr int k = K;
  const int lane = GR::lane();
  SplineFit R;
  R.k = k;
  R.n = 0;
  R.ier = 0;
  R.fp = 0.0;
  R.status = 0;
  constexpr int k1 = K + 1, k2 = K + 2;
  constexpr int nmin = 2 * k1;
  int nest = m + 2 * k;
  constexpr int NK = WS::NK;
  if (nest > NK) nest = NK;
  if (m < k1 || nest < nmin) {
    R.status = 1;
    return R;
  }
  // parcur: u must be strictly increasing (ier = 10 -> ValueError in SciPy)
  {
    bool badl = false;
    for (int i = 1 + lane; i < m; i += G)
      if (!(U[i - 1] < U[i])) badl = true;
    if (GR::ballot(badl) != 0ull) {
      R.status = 1;
      return R;
    }
  }
  const double ub = U[0], ue = U[m - 1];
  const double tol = 0.001;
  const int maxit = 20;
  // fppara.f sets these from single-precision literals (0.1e0, 0.9e0, 0.4e-01) stored in real*8
  const double one = 1.0, con1 = (double)0.1f, con9 = (double)0.9f, con4 = (double)0.04f, half = 0.5;
  const double acc = tol * s;
  const int nmax = m + k1;
  int n = nmin, ier = 0, nplus = 0, nrint = 0, nk1 = 0;
  double fp = 0, fpold = 0, fp0 = 0, fpms = 0;
  if (lane == 0) ws.NRD(1) = m - 2;
  GR::sync();

  bool done = false, to_part2 = false, interp_knots = false;
  while (!done && !to_part2) {
    if (interp_knots) {
      // knots for interpolation (fppara label 10); k odd: t(i) = u(j), k even: midpoints
      interp_knots = false;
      int mk1 = m - k1;
      if (mk1 != 0 && lane == 0) {
        int k3 = k / 2;
        int i = k2;
        int j = k3 + 2;
        for (int l = 1; l <= mk1; l++) {
          ws.t[i] = (k3 * 2 == k) ? (U[j - 1] + U[j - 2]) * half : U[j - 1];
          i++;
          j++;
        }
      }
      GR::sync();
    }
    bool restart = false;
    for (int iter = 1; iter <= m && !restart; iter++) {
      PROFX_T0(1);
      if (n == nmin) ier = -2;
      nrint = n - nmin + 1;
      nk1 = n - k1;
      if (lane < k1) {
        ws.t[1 + lane] = ub;
        ws.t[n - lane] = ue;
      }
      for (int i = 1 + lane; i <= 2 * (NK + 1); i += G) ws.Z(i) = 0.0;
      for (int i = 1 + lane; i <= nk1; i += G)
        for (int j = 1; j <= k1; j++) ws.A(i, j) = 0.0;
      GR::sync();
      fp = 0.0;
      // (groups of 16 lanes and more: the step on a 4 x 4 lane grid, two rotation slots per lane instead of five)
#ifndef FSDP_NO_GIV_GRID
      typename GivState<(G >= 16)>::type gst;
#else
      GivLane gst;
#endif
      giv_init(gst);
      if constexpr (WS::RECOMPUTE) {
        static_assert(!WS::RECOMPUTE || (FAST && K == 3), "the reciprocal table serves the scaling-free cubic fit only");
        if (!knot_reciprocals<G, WS::RD_A0>(ws.t, n, ws.rd)) gst.bad = 1;
        GR::sync();
      } else if constexpr (FAST && K == 3) {
        if (!knot_differences_safe<G>(ws.t, n)) gst.bad = 1;
      }
      int lvq[CH / G] = {};  // knot intervals of this lane's rows of the current chunk
      int lres = k1;  // this lane's previous knot interval (data are increasing: the search resumes there)
      // ---- observation rows: basis values per lane, Givens rotations group-uniform in data order ----
      // the chunk's points (parameter, x, y) are fetched one chunk ahead: the scratch round trip of chunk c + 1 hides
      // behind the Givens pipeline of chunk c
      constexpr int NRC = CH / G;
      double pu[NRC] = {}, pxv[NRC] = {}, pyv[NRC] = {};
      auto fetch_chunk = [&](int base) {
#pragma unroll
        for (int q = 0; q < NRC; q++) {
          int it = base + q * G + lane;  // unconditional loads (index clamped): the chunk's fetches fly together
          it = it < m ? it : m - 1;
          pu[q] = U[it];
          pxv[q] = X[it];
          pyv[q] = Y[it];
        }
      };
      fetch_chunk(0);
      PROFX_T1(1);
      for (int base = 0; base < m; base += CH) {
        const int cnt = m - base < CH ? m - base : CH;
        {
          PROF(10);
#pragma unroll
          for (int q = 0; q < NRC; q++) {
            const int r = q * G + lane;
            const int it = base + r;
            if (r < cnt) {
              double ui = pu[q];
              int l = find_interval_from(ws.t, lres, nk1, ui);
              lres = l;
              double h[K + 2];
              if constexpr (WS::RECOMPUTE)
                fpbspl3_rd<true, WS::RD_A0>(ws.t, ws.rd, ui, l, &h[1], gst.bad);
              else if constexpr (K == 3)
                fpbspl3<FAST>(ws.t, ui, l, &h[1], gst.bad);
              else
                fpbspl<K>(ws.t, ui, l, h);
              double hf[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
              for (int j = 0; j < k1; j++) {
                ws.hq[r][j] = h[j + 1];
                hf[j] = h[j + 1];
              }
              ws.lq[r] = l;
              lvq[q] = l;
              ws.xq[r] = pxv[q];
              ws.yq[r] = pyv[q];
              if constexpr (!WS::RECOMPUTE) {
                BRec* p = &bc.rec[it];
                p->h01 = D2{hf[0], hf[1]};
                p->h23 = D2{hf[2], hf[3]};
              }
              bc.l[it] = (uint8_t)l;  // (RECOMPUTE: the interval is all a later pass reads back, one byte per point)
            }
          }
          GR::sync();
          if (base + CH < m) fetch_chunk(base + CH);
        }
        if constexpr (K == 3) {
          PROF(11);
          // the chunk's rows, one run per knot interval (rows are sorted by interval: a run is contiguous)
          int r = 0;
          while (r < cnt) {
            const int l = ws.lq[r];
            int same = 0;
#pragma unroll
            for (int q = 0; q < NRC; q++) same += __popcll(GR::ballot(q * G + lane < cnt && lvq[q] == l));
            if (l != gst.l) {
              giv_end_run<G, FAST>(gst);
              giv_begin_run(ws, gst, l, lane, n);
            }
            giv_feed<FAST>(ws, gst, r, r + same);
            r += same;
          }
          PROF_COUNT(20, G, cnt);
        } else {
          if (lane == 0) {  // serial section: rows enter the triangle in data order (single writer of a / z)
            for (int r = 0; r < cnt; r++) {
              double h[K + 2];
#pragma unroll
              for (int q = 0; q < k1; q++) h[q + 1] = ws.hq[r][q];
              int l = ws.lq[r];
              double xi1 = ws.xq[r], xi2 = ws.yq[r];
              int j = l - k1;
#pragma unroll
              for (int i = 1; i <= k1; i++) {
                j++;
                double piv = h[i];
                if (piv == 0.0) continue;
                double cs, sn;
                double ww = ws.A(j, 1);
                fpgivs(piv, ww, cs, sn);
                ws.A(j, 1) = ww;
                double z1 = ws.Z(j), z2 = ws.Z(j + n);
                fprota(cs, sn, xi1, z1);
                fprota(cs, sn, xi2, z2);
                ws.Z(j) = z1;
                ws.Z(j + n) = z2;
                if (i == k1) break;
                int i2 = 1;
#pragma unroll
                for (int i1 = i + 1; i1 <= k1; i1++) {
                  i2++;
                  double av = ws.A(j, i2);
                  fprota(cs, sn, h[i1], av);
                  ws.A(j, i2) = av;
                }
              }
              ws.xq[r] = xi1;
              ws.yq[r] = xi2;
            }
          }
          GR::sync();
          fp = seq_sum_squares2(ws.xq, ws.yq, cnt, fp);
        }
        GR::sync();
      }
      PROFX_T0(2);
      if constexpr (K == 3) {
        PROF(12);
        giv_end_run<G, FAST>(gst);
        giv_flush(ws, gst, lane, n);
        fp = gst.fp;
        if (GR::ballot(gst.bad != 0) != 0ull) R.status = ST_RETRY;
        GR::sync();
      }
      if constexpr (WS::BAND_GLOBAL) {
        GR::sync();  // the quad's band rows are in the scratch
        if (lane == 0) {
          PROF(13);
          // fpback for both coordinates in one sweep over the band rows (scratch), the next row fetched while the
          // current one is solved; per coordinate the operations and their order are fpback's
          double a1 = ws.A(nk1, 1), a2 = ws.A(nk1, 2), a3 = ws.A(nk1, 3), a4 = ws.A(nk1, 4), z1 = ws.Z(nk1), z2 = ws.Z(nk1 + n);
          for (int i = nk1; i >= 1; i--) {
            const double b1 = a1, b2 = a2, b3 = a3, b4 = a4, y1 = z1, y2 = z2;
            if (i > 1) {
              a1 = ws.A(i - 1, 1);
              a2 = ws.A(i - 1, 2);
              a3 = ws.A(i - 1, 3);
              a4 = ws.A(i - 1, 4);
              z1 = ws.Z(i - 1);
              z2 = ws.Z(i - 1 + n);
            }
            const int i1 = (nk1 - i) < (k1 - 1) ? (nk1 - i) : (k1 - 1);
            double s1 = y1, s2 = y2;
            const double bl[3] = {b2, b3, b4};
#pragma unroll
            for (int l = 1; l <= 3; l++)
              if (l <= i1) {
                s1 = s1 - ws.c[i + l] * bl[l - 1];
                s2 = s2 - ws.c[n + i + l] * bl[l - 1];
              }
            ws.c[i] = s1 / b1;
            ws.c[n + i] = s2 / b1;
          }
        }
      } else if (lane == 0) {
        PROF(13);
        // back substitution (both coordinates)
        auto ael = [&](int i, int j) { return ws.A(i, j); };
        fpback(ael, &ws.Z(0), nk1, k1, &ws.c[0]);
        fpback(ael, &ws.Z(n), nk1, k1, &ws.c[n]);
      }
      GR::sync();
      if (ier == -2) fp0 = fp;
      if (lane == 0) {
        ws.FPI(n) = fp0;
        ws.FPI(n - 1) = fpold;
        ws.NRD(n) = nplus;
      }
      GR::sync();
      PROFX_T1(2);
      fpms = fp - s;
      if (fabs(fpms) < acc) {
        done = true;
        break;
      }
      if (fpms < 0.) {
        to_part2 = true;
        break;
      }
      if (n == nmax) {
        ier = -1;
        done = true;
        break;
      }
      if (n == nest) {
        ier = 1;
        if (nest == NK && m + 2 * k > NK) R.status = ST_OVERFLOW_KNOTS;
        done = true;
        break;
      }
      if (ier == 0) {
        int npl1 = nplus * 2;
        double rn = nplus;
        if (fpold - fp > acc) npl1 = (int)(rn * fpms / (fpold - fp));
        int mx = npl1 > nplus / 2 ? npl1 : nplus / 2;
        mx = mx > 1 ? mx : 1;
        nplus = (nplus * 2 < mx) ? nplus * 2 : mx;
      } else {
        nplus = 1;
        ier = 0;
      }
      fpold = fp;
      // ---- residual sums per knot interval (terms per lane, accumulation in data order) ----
      {
        PROF(14);
        double fpart = 0.0;
        int ii = 1;
        double* const tbuf = &ws.hq[0][0];     // 4 * CH terms
        int32_t* const fbuf = (int32_t*)ws.xq;  // xq | yq: 4 * CH flags
        ResidualBatch<K, G, true, CH, WS::RECOMPUTE> ra, rb;  // two half super-chunks: one is computed while the other's loads fly
        auto clampc = [&](int left) { return left < 0 ? 0 : (left < SC ? left : SC); };
        ra.load(bc, U, X, Y, 0, clampc(m), m);
        for (int base = 0; base < m; base += HALVES * SC) {
          const int cnt_a = clampc(m - base);
          const int cnt_b = (HALVES == 2) ? clampc(m - base - SC) : 0;
          const int cnt = cnt_a + cnt_b;
          if constexpr (HALVES == 2) rb.load(bc, U, X, Y, base + SC, cnt_b, m);
          ra.compute(ws, cnt_a, n, tbuf, fbuf);
          ra.load(bc, U, X, Y, base + HALVES * SC, 0, m);  // (clamped: harmless past the end)
          if constexpr (HALVES == 2) rb.compute(ws, cnt_b, n, tbuf + SC, fbuf + SC);
          GR::sync();
          for (int r0 = 0; r0 < cnt; r0 += 8) {  // operands eight at a time (one LDS round trip), order kept
            double tv[8];
            int fl[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
              tv[q] = tbuf[r0 + q];  // r0 + q < HALVES * SC always (a multiple of 8)
              fl[q] = fbuf[r0 + q];
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
              if (r0 + q < cnt) {
                double term = tv[q];
                // fppara.f: fpart = fpart+term; on a new interval: store = term*half; fpint(i) = fpart-store; fpart = store
                // ((fpart + term) - term/2 rounds differently from fpart + term/2, and fpknot compares these sums)
                fpart = fpart + term;
                if (fl[q]) {
                  double store = term * half;
                  if (lane == 0) ws.FPI_W(ii) = fpart - store;
                  ii++;
                  fpart = store;
                }
              }
            }
          }
          GR::sync();
        }
        if (lane == 0) ws.FPI_W(nrint) = fpart;
        GR::sync();
        ws.fpint_commit(nrint);
      }
      // ---- add nplus knots (fpknot), group-uniform ----
      for (int lq = 1; lq <= nplus; lq++) {
        {
          PROF(15);
          int kk = (n - nrint - 1) / 2;
          double fpmax = 0.;
          int jbegin = 1;
          int number = 0, maxpt = 0, maxbeg = 0;
          for (int j = 1; j <= nrint; j++) {
            int jpoint = ws.NRD(j);
            if (!(fpmax >= ws.FPI(j) || jpoint == 0)) {
              fpmax = ws.FPI(j);
              number = j;
              maxpt = jpoint;
              maxbeg = jbegin;
            }
            jbegin = jbegin + jpoint + 1;
          }
          int ihalf = maxpt / 2 + 1;
          int nrx = maxbeg + ihalf;
          int next = number + 1;
          GR::sync();
          if (lane == 0) {
            if (next <= nrint) {
              for (int j = next; j <= nrint; j++) {
                int jj = next + nrint - j;
                ws.FPI(jj + 1) = ws.FPI(jj);
                ws.NRD(jj + 1) = ws.NRD(jj);
                int jk = jj + kk;
                ws.t[jk + 1] = ws.t[jk];
              }
            }
            ws.NRD(number) = ihalf - 1;
            ws.NRD(next) = maxpt - ihalf;
            double am = maxpt;
            double an = ws.NRD(number);
            ws.FPI(number) = fpmax * an / am;
            an = ws.NRD(next);
            ws.FPI(next) = fpmax * an / am;
            int jk = next + kk;
            ws.t[jk] = U[nrx - 1];
          }
          n = n + 1;
          nrint = nrint + 1;
          GR::sync();
        }
        if (n == nmax) {
          interp_knots = true;
          restart = true;
          break;
        }
        if (n == nest) {
          // FITPACK stops adding knots at nest = m + 2k.  Where nest is this workspace's capacity instead and knots of
          // this round are still to come, the reference goes on to more knots than fit here: hand the frame on (the
          // next observation pass would otherwise run — and possibly converge — on a knot set the reference never has).
          if (lq < nplus && nest == NK && m + 2 * k > NK) {
            R.status = ST_OVERFLOW_KNOTS;
            done = true;
            restart = true;  // leaves the iteration loop
          }
          break;
        }
      }
    }
    if (!restart && !done && !to_part2) to_part2 = true;
  }

  if (to_part2 && ier != -2) {
    // ---- part 2: smoothing spline, root of f(p) = s ----
    // fpdisc: discontinuity jumps of the k-th derivative at the interior knots (lane = knot)
    PROFX_T0(4);
    {
      int nrint2 = nk1 - k;
      double an = nrint2;
      double fac = an / (ws.t[nk1 + 1] - ws.t[k1]);
      for (int l = k2 + lane; l <= nk1; l += G) {
        double h[2 * K + 3];
        int lmk = l - k1;
#pragma unroll
        for (int j = 1; j <= k1; j++) {
          int ik = j + k1;
          int lj = l + j;
          int lk = lj - k2;
          h[j] = ws.t[l] - ws.t[lk];
          h[ik] = ws.t[l] - ws.t[lj];
        }
        int lp = lmk;
#pragma unroll
        for (int j = 1; j <= k2; j++) {
          int jk = j;
          double prod = h[j];
#pragma unroll
          for (int i = 1; i <= k; i++) {
            jk = jk + 1;
            prod = prod * h[jk] * fac;
          }
          int lk = lp + k1;
          bc.b[5 * lmk + j - 1] = (ws.t[lk] - ws.t[lp]) / prod;
          lp = lp + 1;
        }
      }
      GR::sync();
    }
    PROFX_T1(4);
    PROFX_T0(3);
    int bad2 = 0;  // (FAST: an operand of the smoothing rows' rotations left the exponent band of the scaling-free divisions)
    double p1 = 0., f1 = fp0 - s, p3 = -one, f3 = fpms, p = 0.;
    for (int i = 1; i <= nk1; i++) p = p + ws.A(i, 1);
    PROFX_T1(3);
    double rn = nk1;
    p = rn / p;
    int ich1 = 0, ich3 = 0;
    const int n8 = n - nmin;
    for (int iter = 1; iter <= maxit; iter++) {
      double pinv = one / p;
      PROF_COUNT(24, G, 1);
      PROFX_T0(5);
      GR::sync();
      if constexpr (WS::BAND_GLOBAL) {
        // the triangle and its right-hand sides come from the frame's scratch: every lane's loads in one batch (indices
        // clamped), not one round trip per row
        constexpr int NZ = (2 * NK + G - 1) / G, NRW = (NK - 4 + G - 1) / G;
        double zv[NZ], av[NRW][4];
#pragma unroll
        for (int q = 0; q < NZ; q++) {
          const int i = 1 + lane + q * G;
          zv[q] = ws.Z(i <= 2 * n ? i : 2 * n);
        }
#pragma unroll
        for (int q = 0; q < NRW; q++) {
          const int i = 1 + lane + q * G, ic = i <= nk1 ? i : nk1;
#pragma unroll
          for (int j = 1; j <= 4; j++) av[q][j - 1] = ws.A(ic, j);
        }
#pragma unroll
        for (int q = 0; q < NZ; q++) {
          const int i = 1 + lane + q * G;
          if (i <= 2 * n) ws.c[i] = zv[q];
        }
#pragma unroll
        for (int q = 0; q < NRW; q++) {
          const int i = 1 + lane + q * G;
          if (i <= nk1) {
            ws.Gm(i, k2) = 0.;
#pragma unroll
            for (int j = 1; j <= 4; j++) ws.Gm(i, j) = av[q][j - 1];
          }
        }
      } else {
        for (int i = 1 + lane; i <= 2 * n; i += G) ws.c[i] = ws.Z(i);
        for (int i = 1 + lane; i <= nk1; i += G) {
          ws.Gm(i, k2) = 0.;
          for (int j = 1; j <= k1; j++) ws.Gm(i, j) = ws.A(i, j);
        }
      }
      GR::sync();
      PROFX_T1(5);
      if constexpr (G >= K + 4) {
        // Column-parallel rotations (fppara's smoothing rows b / p into the triangle g): lane q < k2 holds column q + 1 of
        // the incoming row and, at band row j, the element g(j, q + 1); lanes k2 and k2 + 1 hold the two right-hand
        // sides.  Every lane derives the rotation (cs, sn) from the pivot (lane 0's element) and g(j, 1), applies it to
        // its own pair, and the row moves one lane down — the same operations on the same operands as the column loop
        // of the published algorithm.  (Reads of a step come before its lane exchange, writes after it.)
        PROF(16);
        const bool is_rhs = lane == k2 || lane == k2 + 1;
        double hq = 0.0;
        double bnext = (lane < k2 && n8 >= 1) ? bc.b[5 * 1 + lane] : 0.0;
        for (int it = 1; it <= n8; it++) {
          hq = (lane < k2) ? bnext * pinv : 0.0;  // (right-hand-side lanes: xi = 0)
          if (it < n8 && lane < k2) bnext = bc.b[5 * (it + 1) + lane];
          for (int j = it; j <= nk1; j++) {
            double* const pb = is_rhs ? &ws.c[j + (lane - k2) * n] : &ws.Gm(j, lane < k2 ? lane + 1 : 1);
            double b = *pb;
            double ww = ws.Gm(j, 1);
            const double piv = GR::bcast(hq, 0);
            double cs, sn;
            fpgivs_guarded<FAST>(piv, ww, cs, sn, bad2);
            int i2 = k1;
            if (j > n8) i2 = nk1 - j;
            const bool col_on = lane >= 1 && lane < k2 && j != nk1 && lane <= i2;  // columns 2 .. i2 + 1
            double a = hq;
            fprota(cs, sn, a, b);
            if (lane == 0)
              *pb = ww;
            else if (col_on || is_rhs)
              *pb = b;
            if (j == nk1) break;
            const double mine = (col_on || is_rhs) ? a : hq;
            const double nxt = GR::shfl_down1(mine);
            if (lane < k2) {
              const int col = lane + 1;
              hq = (col <= i2) ? nxt : (col == i2 + 1 ? 0.0 : mine);
            } else {
              hq = mine;
            }
          }
          GR::sync();
        }
        if (lane < 2) {  // the two right-hand sides side by side
          auto gel = [&](int i, int jj) { return ws.Gm(i, jj); };
          fpback(gel, &ws.c[lane * n], nk1, k2, &ws.c[lane * n]);
        }
      } else if (lane == 0) {  // serial section (single writer of g / c): groups of fewer than k2 + 2 lanes
        PROF(16);
        double bn[K + 3];  // next row of b, fetched from scratch one row ahead
#pragma unroll
        for (int i = 1; i <= k2; i++) bn[i] = (n8 >= 1) ? bc.b[5 * 1 + i - 1] : 0.0;
        for (int it = 1; it <= n8; it++) {
          double h[K + 4];
#pragma unroll
          for (int i = 1; i <= k2; i++) h[i] = bn[i] * pinv;
          if (it < n8) {
#pragma unroll
            for (int i = 1; i <= k2; i++) bn[i] = bc.b[5 * (it + 1) + i - 1];
          }
          double xi1 = 0., xi2 = 0.;
          for (int j = it; j <= nk1; j++) {
            double piv = h[1];
            double cs, sn;
            double ww = ws.Gm(j, 1);
            fpgivs_guarded<FAST>(piv, ww, cs, sn, bad2);
            ws.Gm(j, 1) = ww;
            double c1 = ws.c[j], c2 = ws.c[j + n];
            fprota(cs, sn, xi1, c1);
            fprota(cs, sn, xi2, c2);
            ws.c[j] = c1;
            ws.c[j + n] = c2;
            if (j == nk1) break;
            int i2 = k1;
            if (j > n8) i2 = nk1 - j;
#pragma unroll
            for (int i = 1; i <= k1; i++) {
              if (i <= i2) {
                int i1 = i + 1;
                double gv = ws.Gm(j, i1);
                fprota(cs, sn, h[i1], gv);
                ws.Gm(j, i1) = gv;
                h[i] = h[i1];
              }
            }
#pragma unroll
            for (int i = 1; i <= k2; i++)
              if (i == i2 + 1) h[i] = 0.;
          }
        }
        auto gel = [&](int i, int j) { return ws.Gm(i, j); };
        fpback(gel, &ws.c[0], nk1, k2, &ws.c[0]);
        fpback(gel, &ws.c[n], nk1, k2, &ws.c[n]);
      }
      GR::sync();
      PROFX_T0(6);
      if constexpr (WS::RECOMPUTE) {
        (void)knot_reciprocals<G, WS::RD_A0>(ws.t, n, ws.rd);  // (checked when this knot set's observation pass built it)
        GR::sync();
      }
      PROFX_T1(6);
      // f(p): terms per lane, accumulation in data order
      PROF(17);
      fp = 0.;
      {
        double* const tbuf = &ws.hq[0][0];  // 4 * CH terms
        ResidualBatch<K, G, false, CH, WS::RECOMPUTE> ra, rb;  // two half super-chunks: one is computed while the other's loads fly
        auto clampc = [&](int left) { return left < 0 ? 0 : (left < SC ? left : SC); };
        ra.load(bc, U, X, Y, 0, clampc(m), m);
        for (int base = 0; base < m; base += HALVES * SC) {
          const int cnt_a = clampc(m - base);
          const int cnt_b = (HALVES == 2) ? clampc(m - base - SC) : 0;
          {
            PROF(28);
            if constexpr (HALVES == 2) rb.load(bc, U, X, Y, base + SC, cnt_b, m);
            ra.compute(ws, cnt_a, n, tbuf, nullptr);
            ra.load(bc, U, X, Y, base + HALVES * SC, 0, m);  // (clamped: harmless past the end)
            if constexpr (HALVES == 2) rb.compute(ws, cnt_b, n, tbuf + SC, nullptr);
            GR::sync();
          }
          PROF(29);
          fp = seq_sum(tbuf, cnt_a + cnt_b, fp);  // w = 1: term * w^2 is the term itself
          GR::sync();
        }
      }
      fpms = fp - s;
      if (fabs(fpms) < acc) break;
      if (iter == maxit) {
        ier = 3;
        break;
      }
      double p2 = p, f2 = fpms;
      bool do_rati = true;
      if (ich3 == 0) {
        if ((f2 - f3) > acc) {
          if (f2 < 0.) ich3 = 1;
        } else {
          p3 = p2;
          f3 = f2;
          p = p * con4;
          if (p <= p1) p = p1 * con9 + p2 * con1;
          do_rati = false;
        }
      }
      if (do_rati && ich1 == 0) {
        if ((f1 - f2) > acc) {
          if (f2 > 0.) ich1 = 1;
        } else {
          p1 = p2;
          f1 = f2;
          p = p / con4;
          if (!(p3 < 0.)) {
            if (p >= p3) p = p2 * con1 + p3 * con9;
          }
          do_rati = false;
        }
      }
      if (do_rati) {
        if (f2 >= f1 || f2 <= f3) {
          ier = 2;
          break;
        }
        // fprati
        double pn;
        if (p3 > 0.) {
          double h1 = f1 * (f2 - f3);
          double h2 = f2 * (f3 - f1);
          double h3 = f3 * (f1 - f2);
          pn = -(p1 * p2 * h3 + p2 * p3 * h1 + p3 * p1 * h2) / (p1 * h1 + p2 * h2 + p3 * h3);
        } else {
          pn = (p1 * (f1 - f3) * f2 - p2 * (f2 - f3) * f1) / ((f1 - f2) * f3);
        }
        if (f2 < 0.) {
          p3 = p2;
          f3 = f2;
        } else {
          p1 = p2;
          f1 = f2;
        }
        p = pn;
      }
    }
    if (GR::ballot(bad2 != 0) != 0ull) R.status = ST_RETRY;
  }
  GR::sync();
  R.n = n;
  R.ier = ier;
  R.fp = fp;
  return R;
}

template <bool FAST, class WS>
__device__ __forceinline__ SplineFit spline_fit(WS& ws, const BasisCache& bc, const double* U, const double* X, const double* Y,
                                       int m, int k, double s) {
  if (k == 3) return spline_fit_k<3, FAST>(ws, bc, U, X, Y, m, s);
  if (k == 2) return spline_fit_k<2, FAST>(ws, bc, U, X, Y, m, s);
  return spline_fit_k<1, FAST>(ws, bc, U, X, Y, m, s);
}

// splev (der = 0, ext = 0) at arg = i * step for i in [0, count): one evaluation point per lane.
// Outputs to OX/OY (LDS or global), optional parameter values to OU.  Reads only t / c, so the outputs may alias the
// rest of the fit workspace (the dense samples of the final spline do).
template <int K, class WS>
__device__ __forceinline__ void spline_eval_k(const WS& ws, const SplineFit& f, double step, int count, double* OX, double* OY,
                                     double* OU) {
  constexpr int G = WS::GRP;
  const int lane = Grp<G>::lane();
  const int n = f.n;
  constexpr int k1 = K + 1;
  const int nk1 = n - k1;
  int lcur = k1;  // a lane's evaluation points increase: the interval search resumes where the previous one ended
  for (int i = lane; i < count; i += G) {
    double arg = (double)i * step;
    int l = find_interval_from(ws.t, lcur, nk1, arg);
    lcur = l;
    double h[K + 2];
    if constexpr (K == 3) {
      int unused = 0;
      fpbspl3<false>(ws.t, arg, l, &h[1], unused);
    } else {
      fpbspl<K>(ws.t, arg, l, h);
    }
    double sx = 0., sy = 0.;
    int ll = l - k1;
#pragma unroll
    for (int j = 1; j <= k1; j++) {
      ll++;
      sx = sx + ws.c[ll] * h[j];
      sy = sy + ws.c[ll + n] * h[j];
    }
    OX[i] = sx;
    OY[i] = sy;
    if (OU) OU[i] = arg;
  }
  Grp<G>::sync();
}

template <class WS>
__device__ __forceinline__ void spline_eval(const WS& ws, const SplineFit& f, double step, int count, double* OX, double* OY,
                                   double* OU) {
  if (f.k == 3)
    spline_eval_k<3>(ws, f, step, count, OX, OY, OU);
  else if (f.k == 2)
    spline_eval_k<2>(ws, f, step, count, OX, OY, OU);
  else
    spline_eval_k<1>(ws, f, step, count, OX, OY, OU);
}

}  // namespace fsdp
