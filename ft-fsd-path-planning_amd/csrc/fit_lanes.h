// The refit (fit #2 of the path stage: utils/spline_fit.py:95-128 -> SciPy splprep = FITPACK parcur / fppara, k = 3, s = 0.2)
// with ONE FRAME PER LANE: sixty-four fits per wavefront, every lane running FITPACK's own scalar sequence on its frame.
//
// Why this form (round 5; DESIGN.md "What bounds the pipeline"): with four lanes per frame (fit_kernel<4>, spline_device.h) a
// data row costs one step of the Givens quad pipeline — 109 wave-instructions for sixteen frames, every lane carrying a full
// fpgivs and five plane rotations whatever its stage needs — plus the chunk staging through LDS: 8.5 instructions per row and
// frame.  A lane that rotates its own row through its own four band rows issues the 4 fpgivs and exactly the 3 + 2 + 1 + 0 band
// and 4 x 2 right-hand-side rotations FITPACK has, for sixty-four frames at once: ~4.9 per row and frame.  The serial sections
// (back substitution, knot selection, the smoothing rows, every running sum) are free in this form: each lane's own scalars.
// What it costs: 64 frames per wavefront means sixty-four times fewer wavefronts (4096 frames = 64 of them), so this kernel is
// for batches in flight that still fill the chip (the host picks it from the frames in flight, fsdp_lib.hip launch_path), and
// a wavefront lives as long as the slowest of its 64 fits (consecutive frames of a replay: 0.75-0.82 lane efficiency measured
// on the bench workload, profiles/r05_fit_lanes.txt).
//
// Memory: the polyline (parameter u, x, y) comes from a TILE of 64 frames, point-major — T[array][point][frame mod 64] — so
// that "point i of my frame" is one coalesced 512-byte access per array (path_prep_kernel writes the tile next to the frame's
// own arena copy).  Everything else of a fit — knots, coefficients, the band triangle, the smoothing rows — lives in the
// lane's private memory (scratch: lane-interleaved, i.e. coalesced where the lanes agree on the index); the hot loops keep what
// a knot interval needs (six knots, five refined reciprocals, the first-level quotient, the four coefficients per coordinate,
// the four band rows of the Givens window) in registers and touch private memory only when the interval changes — a handful of
// times per pass.  No LDS at all.
//
// Arithmetic: the operation sequence per element is FITPACK's (and spline_fit_k<3, true>'s: same scaling-free divisions, same
// guards — a frame whose operands leave their exponent band is handed to the exact kernel); results are bit-identical to the
// other instantiations (tests/test_kernel_logic_emulated.py, tests/test_gpu_parity.py).
#pragma once
#include "spline_device.h"

namespace fsdp {

constexpr int LANES_KNOTS = 32;  // knots per fit the lane form keeps (twice the packed kernels': costs private memory only)

template <int NK>
struct LaneWS {
  double t[NK + 2];
  double cx[NK + 2], cy[NK + 2];
  double a[NK + 2][4];
  double zx[NK + 2], zy[NK + 2];
  double fpint[NK + 2];
  int nrd[NK + 2];
  double g[NK + 2][5];
  double b[NK + 2][5];
};

// what a data point in knot interval l needs from the knot vector (fpbspl3_rd's operands), held in registers
struct IntervalCtx {
  double tm2, tm1, t0, tp1, tp2, tp3;
  double rm12, r02, rm23, rm13, r03;  // refined reciprocals of t(l+1)-t(l-1), t(l+2)-t(l), t(l+1)-t(l-2), t(l+2)-t(l-1), t(l+3)-t(l)
  double f1;                          // 1 / (t(l+1) - t(l)) as fpbspl forms it: the level-1 quotient, the same for every point of the interval
};

__device__ __forceinline__ void load_interval(const double* t, int l, IntervalCtx& c) {
  c.tm2 = t[l - 2];
  c.tm1 = t[l - 1];
  c.t0 = t[l];
  c.tp1 = t[l + 1];
  c.tp2 = t[l + 2];
  c.tp3 = t[l + 3];
  const double d01 = c.tp1 - c.t0;
  c.f1 = div_rcp(1.0, d01, rcp_refined(d01));
  c.rm12 = rcp_refined(c.tp1 - c.tm1);
  c.r02 = rcp_refined(c.tp2 - c.t0);
  c.rm23 = rcp_refined(c.tp1 - c.tm2);
  c.rm13 = rcp_refined(c.tp2 - c.tm1);
  c.r03 = rcp_refined(c.tp3 - c.t0);
}

// fpbspl3_rd with the interval's operands in registers: the operations of fpbspl3<true> on its operands, in its order
template <bool CHECK>
__device__ __forceinline__ void basis_lane(const IntervalCtx& c, double x, double& h1, double& h2, double& h3, double& h4, bool& ok) {
  auto chk = [&](double num) {
    if constexpr (CHECK) ok = ok & ((num == 0.0) | ((num >= 0x1p-255) & (num <= 0x1p255)));
  };
  const double e1 = c.tp1 - x, e2 = c.tp2 - x, e3 = c.tp3 - x;
  const double g0 = x - c.t0, g1 = x - c.tm1, g2 = x - c.tm2;
  {
    const double f = c.f1;
    h1 = 0.0 + f * e1;
    h2 = f * g0;
  }
  {
    const double a1 = h1, a2 = h2;
    h1 = 0.0;
    {
      chk(a1);
      const double f = div_rcp(a1, c.tp1 - c.tm1, c.rm12);
      h1 = h1 + f * e1;
      h2 = f * g1;
    }
    {
      chk(a2);
      const double f = div_rcp(a2, c.tp2 - c.t0, c.r02);
      h2 = h2 + f * e2;
      h3 = f * g0;
    }
  }
  {
    const double a1 = h1, a2 = h2, a3 = h3;
    h1 = 0.0;
    {
      chk(a1);
      const double f = div_rcp(a1, c.tp1 - c.tm2, c.rm23);
      h1 = h1 + f * e1;
      h2 = f * g2;
    }
    {
      chk(a2);
      const double f = div_rcp(a2, c.tp2 - c.tm1, c.rm13);
      h2 = h2 + f * e2;
      h3 = f * g1;
    }
    {
      chk(a3);
      const double f = div_rcp(a3, c.tp3 - c.t0, c.r03);
      h3 = h3 + f * e3;
      h4 = f * g0;
    }
  }
}

// fpgivs with the scaling-free divisions (giv_step<true>'s arithmetic) and its guard; rot = false leaves ww and returns cs = 1, sn = 0
__device__ __forceinline__ void givens_lane(double piv, double& ww, double& cs, double& sn, bool& bad) {
  const bool rot = piv != 0.0;
  const double w = ww;
  const double den = max_abs_nn(piv, w), num = min_abs_nn(piv, w);
  bad |= rot & !((den >= 0x1p-255) & (den <= 0x1p+255) & ((num == 0.0) | (num >= 0x1p-255)));
  double dd, rd;
  givens_dd_rd(den, num, dd, rd);
  const double c1 = div_rcp(w, dd, rd);
  const double s1 = div_rcp(piv, dd, rd);
  cs = rot ? c1 : 1.0;
  sn = rot ? s1 : 0.0;
  ww = rot ? dd : w;
}
// fprota(cs, sn, a, b): a = the data row's element, b = the band row's
__device__ __forceinline__ void rota_lane(double cs, double sn, double& a, double& b) {
  const double s1 = a, s2 = b;
  b = cs * s2 + sn * s1;
  a = cs * s1 - sn * s2;
}

// every difference t(a + j) - t(a), j = 1..3, of the knot vector is 0 or inside the exponent band of the scaling-free division,
// and no interior knot interval is empty (knot_reciprocals' checks, one lane's loop over its own knots)
__device__ __forceinline__ bool knots_safe_lane(const double* t, int n) {
  bool badk = false;
  for (int a = 1; a <= n; a++) {
#pragma unroll
    for (int j = 1; j <= 3; j++) {
      const int b = a + j <= n ? a + j : n;
      const double d = t[b] - t[a];
      const bool zero = d == 0.0;
      badk |= !(zero | ((d >= 0x1p-255) & (d <= 0x1p255)));
      if (j == 1 && a >= 4 && a <= n - 4) badk |= zero;
    }
  }
  return !badk;
}

// parcur / fppara for idim = 2, w = 1, iopt = 0, k = 3 on the polyline (U, X, Y)[i * STRIDE], i < m (STRIDE = 64: a tile of 64
// frames, point-major).  One lane = one fit; lanes with m = 0 skip.  Result: knots / coefficients in ws.t / ws.cx / ws.cy.
template <int NK, int STRIDE>
__device__ __forceinline__ SplineFit spline_fit_lane(LaneWS<NK>& ws, const double* U, const double* X, const double* Y, int m, double s) {
  constexpr int k = 3, k1 = 4, k2 = 5, nmin = 8;
  SplineFit R;
  R.k = k;
  R.n = 0;
  R.ier = 0;
  R.fp = 0.0;
  R.status = 0;
  int nest = m + 2 * k;
  if (nest > NK) nest = NK;
  if (m < k1 || nest < nmin) {
    R.status = 1;
    return R;
  }
  {  // parcur: u strictly increasing
    bool badl = false;
    double prev = U[0];
    for (int i = 1; i < m; i++) {
      const double cur = U[(size_t)i * STRIDE];
      badl |= !(prev < cur);
      prev = cur;
    }
    if (badl) {
      R.status = 1;
      return R;
    }
  }
  const double ub = U[0], ue = U[(size_t)(m - 1) * STRIDE];
  const double tol = 0.001;
  const int maxit = 20;
  const double one = 1.0, con1 = (double)0.1f, con9 = (double)0.9f, con4 = (double)0.04f, half = 0.5;
  const double acc = tol * s;
  const int nmax = m + k1;
  int n = nmin, ier = 0, nplus = 0, nrint = 0, nk1 = 0;
  double fp = 0, fpold = 0, fp0 = 0, fpms = 0;
  ws.nrd[1] = m - 2;
  bool bad = false;

  bool done = false, to_part2 = false, interp_knots = false;
  while (!done && !to_part2) {
    if (interp_knots) {
      interp_knots = false;
      const int mk1 = m - k1;
      if (mk1 != 0) {
        int i = k2, j = k / 2 + 2;  // k odd: t(i) = u(j)
        for (int l = 1; l <= mk1; l++) {
          ws.t[i] = U[(size_t)(j - 1) * STRIDE];
          i++;
          j++;
        }
      }
    }
    bool restart = false;
    for (int iter = 1; iter <= m && !restart; iter++) {
      if (n == nmin) ier = -2;
      nrint = n - nmin + 1;
      nk1 = n - k1;
      for (int j = 1; j <= k1; j++) {
        ws.t[j] = ub;
        ws.t[n + 1 - j] = ue;
      }
      if (!knots_safe_lane(ws.t, n)) bad = true;
      if (bad) break;
      for (int i = 1; i <= nk1; i++) {
        ws.a[i][0] = ws.a[i][1] = ws.a[i][2] = ws.a[i][3] = 0.0;
        ws.zx[i] = ws.zy[i] = 0.0;
      }
      // ---- observation pass: the row of every data point rotated into the band rows l-3 .. l of its knot interval l ----
      {
        // the Givens window: band rows l-3 .. l; a row's columns beyond interval l are still exact zeros and are not kept
        double w00 = 0, w01 = 0, w02 = 0, w03 = 0, w10 = 0, w11 = 0, w12 = 0, w20 = 0, w21 = 0, w30 = 0;
        double zx0 = 0, zx1 = 0, zx2 = 0, zx3 = 0, zy0 = 0, zy1 = 0, zy2 = 0, zy3 = 0;
        int l = k1;
        IntervalCtx ic;
        load_interval(ws.t, l, ic);
        fp = 0.0;
        bool ok = true;
        double un = U[0], xn = X[0], yn = Y[0];
        for (int i = 0; i < m; i++) {
          const double ui = un, xi0 = xn, yi0 = yn;
          {  // the next row's point is fetched while this one is rotated
            const int i2 = i + 1 < m ? i + 1 : m - 1;
            un = U[(size_t)i2 * STRIDE];
            xn = X[(size_t)i2 * STRIDE];
            yn = Y[(size_t)i2 * STRIDE];
          }
          while (!(ui < ic.tp1 || l == nk1)) {
            // the window moves on: band row l-3 is complete
            ws.a[l - 3][0] = w00;
            ws.a[l - 3][1] = w01;
            ws.a[l - 3][2] = w02;
            ws.a[l - 3][3] = w03;
            ws.zx[l - 3] = zx0;
            ws.zy[l - 3] = zy0;
            w00 = w10;
            w01 = w11;
            w02 = w12;
            w03 = 0.0;
            zx0 = zx1;
            zy0 = zy1;
            w10 = w20;
            w11 = w21;
            w12 = 0.0;
            zx1 = zx2;
            zy1 = zy2;
            w20 = w30;
            w21 = 0.0;
            zx2 = zx3;
            zy2 = zy3;
            w30 = 0.0;
            zx3 = 0.0;
            zy3 = 0.0;
            l++;
            load_interval(ws.t, l, ic);
          }
          double h1, h2, h3, h4;
          basis_lane<true>(ic, ui, h1, h2, h3, h4, ok);
          double xi = xi0, yi = yi0, cs, sn;
          givens_lane(h1, w00, cs, sn, bad);
          rota_lane(cs, sn, xi, zx0);
          rota_lane(cs, sn, yi, zy0);
          rota_lane(cs, sn, h2, w01);
          rota_lane(cs, sn, h3, w02);
          rota_lane(cs, sn, h4, w03);
          givens_lane(h2, w10, cs, sn, bad);
          rota_lane(cs, sn, xi, zx1);
          rota_lane(cs, sn, yi, zy1);
          rota_lane(cs, sn, h3, w11);
          rota_lane(cs, sn, h4, w12);
          givens_lane(h3, w20, cs, sn, bad);
          rota_lane(cs, sn, xi, zx2);
          rota_lane(cs, sn, yi, zy2);
          rota_lane(cs, sn, h4, w21);
          givens_lane(h4, w30, cs, sn, bad);
          rota_lane(cs, sn, xi, zx3);
          rota_lane(cs, sn, yi, zy3);
          fp = fp + xi * xi;
          fp = fp + yi * yi;
        }
        bad |= !ok;
        // the window's rows: l-3 .. l (l = nk1 once the last point, u = ue, has been seen)
        ws.a[l - 3][0] = w00;
        ws.a[l - 3][1] = w01;
        ws.a[l - 3][2] = w02;
        ws.a[l - 3][3] = w03;
        ws.zx[l - 3] = zx0;
        ws.zy[l - 3] = zy0;
        ws.a[l - 2][0] = w10;
        ws.a[l - 2][1] = w11;
        ws.a[l - 2][2] = w12;
        ws.a[l - 2][3] = 0.0;
        ws.zx[l - 2] = zx1;
        ws.zy[l - 2] = zy1;
        ws.a[l - 1][0] = w20;
        ws.a[l - 1][1] = w21;
        ws.a[l - 1][2] = 0.0;
        ws.a[l - 1][3] = 0.0;
        ws.zx[l - 1] = zx2;
        ws.zy[l - 1] = zy2;
        ws.a[l][0] = w30;
        ws.a[l][1] = 0.0;
        ws.a[l][2] = 0.0;
        ws.a[l][3] = 0.0;
        ws.zx[l] = zx3;
        ws.zy[l] = zy3;
      }
      if (bad) break;
      // fpback for both coordinates in one sweep over the band rows
      {
        double c1x = 0, c2x = 0, c3x = 0, c1y = 0, c2y = 0, c3y = 0;  // c(i+1), c(i+2), c(i+3)
        for (int i = nk1; i >= 1; i--) {
          const double b1 = ws.a[i][0], b2 = ws.a[i][1], b3 = ws.a[i][2], b4 = ws.a[i][3];
          const int i1 = (nk1 - i) < (k1 - 1) ? (nk1 - i) : (k1 - 1);
          double s1 = ws.zx[i], s2 = ws.zy[i];
          if (1 <= i1) {
            s1 = s1 - c1x * b2;
            s2 = s2 - c1y * b2;
          }
          if (2 <= i1) {
            s1 = s1 - c2x * b3;
            s2 = s2 - c2y * b3;
          }
          if (3 <= i1) {
            s1 = s1 - c3x * b4;
            s2 = s2 - c3y * b4;
          }
          const double vx = s1 / b1, vy = s2 / b1;
          ws.cx[i] = vx;
          ws.cy[i] = vy;
          c3x = c2x;
          c2x = c1x;
          c1x = vx;
          c3y = c2y;
          c2y = c1y;
          c1y = vy;
        }
      }
      if (ier == -2) fp0 = fp;
      ws.fpint[n] = fp0;
      ws.fpint[n - 1] = fpold;
      ws.nrd[n] = nplus;
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
        const double rn = nplus;
        if (fpold - fp > acc) npl1 = (int)(rn * fpms / (fpold - fp));
        int mx = npl1 > nplus / 2 ? npl1 : nplus / 2;
        mx = mx > 1 ? mx : 1;
        nplus = (nplus * 2 < mx) ? nplus * 2 : mx;
      } else {
        nplus = 1;
        ier = 0;
      }
      fpold = fp;
      // ---- residual sums per knot interval ----
      {
        double fpart = 0.0;
        int ii = 1;
        int l = k1;
        IntervalCtx ic;
        load_interval(ws.t, l, ic);
        double c0x = ws.cx[l - 3], c1x = ws.cx[l - 2], c2x = ws.cx[l - 1], c3x = ws.cx[l];
        double c0y = ws.cy[l - 3], c1y = ws.cy[l - 2], c2y = ws.cy[l - 1], c3y = ws.cy[l];
        double un = U[0], xn = X[0], yn = Y[0];
        for (int i = 0; i < m; i++) {
          const double ui = un, xi = xn, yi = yn;
          {
            const int i2 = i + 1 < m ? i + 1 : m - 1;
            un = U[(size_t)i2 * STRIDE];
            xn = X[(size_t)i2 * STRIDE];
            yn = Y[(size_t)i2 * STRIDE];
          }
          bool nw = false;
          while (!(ui < ic.tp1 || l == nk1)) {
            l++;
            nw = true;
            load_interval(ws.t, l, ic);
            c0x = ws.cx[l - 3];
            c1x = ws.cx[l - 2];
            c2x = ws.cx[l - 1];
            c3x = ws.cx[l];
            c0y = ws.cy[l - 3];
            c1y = ws.cy[l - 2];
            c2y = ws.cy[l - 1];
            c3y = ws.cy[l];
          }
          double h1, h2, h3, h4;
          bool unused = true;
          basis_lane<false>(ic, ui, h1, h2, h3, h4, unused);
          double term = 0.0;
          {
            double fac = 0.0;
            fac = fac + c0x * h1;
            fac = fac + c1x * h2;
            fac = fac + c2x * h3;
            fac = fac + c3x * h4;
            const double dv = 1.0 * (fac - xi);
            term = term + dv * dv;
          }
          {
            double fac = 0.0;
            fac = fac + c0y * h1;
            fac = fac + c1y * h2;
            fac = fac + c2y * h3;
            fac = fac + c3y * h4;
            const double dv = 1.0 * (fac - yi);
            term = term + dv * dv;
          }
          // fppara.f: fpart = fpart+term; on a new interval: store = term*half; fpint(i) = fpart-store; fpart = store
          fpart = fpart + term;
          if (nw) {
            const double store = term * half;
            ws.fpint[ii] = fpart - store;
            ii++;
            fpart = store;
          }
        }
        ws.fpint[nrint] = fpart;
      }
      // ---- add nplus knots (fpknot) ----
      for (int lq = 1; lq <= nplus; lq++) {
        const int kk = (n - nrint - 1) / 2;
        double fpmax = 0.;
        int jbegin = 1;
        int number = 0, maxpt = 0, maxbeg = 0;
        for (int j = 1; j <= nrint; j++) {
          const int jpoint = ws.nrd[j];
          const double fj = ws.fpint[j];
          if (!(fpmax >= fj || jpoint == 0)) {
            fpmax = fj;
            number = j;
            maxpt = jpoint;
            maxbeg = jbegin;
          }
          jbegin = jbegin + jpoint + 1;
        }
        const int ihalf = maxpt / 2 + 1;
        const int nrx = maxbeg + ihalf;
        const int next = number + 1;
        if (next <= nrint) {
          for (int j = next; j <= nrint; j++) {
            const int jj = next + nrint - j;
            ws.fpint[jj + 1] = ws.fpint[jj];
            ws.nrd[jj + 1] = ws.nrd[jj];
            const int jk = jj + kk;
            ws.t[jk + 1] = ws.t[jk];
          }
        }
        ws.nrd[number] = ihalf - 1;
        ws.nrd[next] = maxpt - ihalf;
        const double am = maxpt;
        double an = ws.nrd[number];
        ws.fpint[number] = fpmax * an / am;
        an = ws.nrd[next];
        ws.fpint[next] = fpmax * an / am;
        ws.t[next + kk] = U[(size_t)(nrx - 1) * STRIDE];
        n = n + 1;
        nrint = nrint + 1;
        if (n == nmax) {
          interp_knots = true;
          restart = true;
          break;
        }
        if (n == nest) {
          if (lq < nplus && nest == NK && m + 2 * k > NK) {
            R.status = ST_OVERFLOW_KNOTS;
            done = true;
            restart = true;
          }
          break;
        }
      }
    }
    if (bad) break;
    if (!restart && !done && !to_part2) to_part2 = true;
  }

  if (!bad && to_part2 && ier != -2) {
    // ---- part 2: smoothing spline, root of f(p) = s ----
    {  // fpdisc
      const int nrint2 = nk1 - k;
      const double an = nrint2;
      const double fac = an / (ws.t[nk1 + 1] - ws.t[k1]);
      for (int l = k2; l <= nk1; l++) {
        double h[2 * k + 3];
        const int lmk = l - k1;
#pragma unroll
        for (int j = 1; j <= k1; j++) {
          const int ik = j + k1, lj = l + j, lk = lj - k2;
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
          const int lk = lp + k1;
          ws.b[lmk][j - 1] = (ws.t[lk] - ws.t[lp]) / prod;
          lp = lp + 1;
        }
      }
    }
    double p1 = 0., f1 = fp0 - s, p3 = -one, f3 = fpms, p = 0.;
    for (int i = 1; i <= nk1; i++) p = p + ws.a[i][0];
    const double rn = nk1;
    p = rn / p;
    int ich1 = 0, ich3 = 0;
    const int n8 = n - nmin;
    for (int iter = 1; iter <= maxit; iter++) {
      const double pinv = one / p;
      for (int i = 1; i <= nk1; i++) {
        ws.cx[i] = ws.zx[i];
        ws.cy[i] = ws.zy[i];
        ws.g[i][0] = ws.a[i][0];
        ws.g[i][1] = ws.a[i][1];
        ws.g[i][2] = ws.a[i][2];
        ws.g[i][3] = ws.a[i][3];
        ws.g[i][4] = 0.;
      }
      for (int it = 1; it <= n8; it++) {
        double hh1 = ws.b[it][0] * pinv, hh2 = ws.b[it][1] * pinv, hh3 = ws.b[it][2] * pinv, hh4 = ws.b[it][3] * pinv, hh5 = ws.b[it][4] * pinv;
        double xi1 = 0., xi2 = 0.;
        for (int j = it; j <= nk1; j++) {
          const double piv = hh1;
          double cs, sn;
          double ww = ws.g[j][0];
          {
            int b2 = 0;
            fpgivs_guarded<true>(piv, ww, cs, sn, b2);
            bad |= b2 != 0;
          }
          ws.g[j][0] = ww;
          double c1 = ws.cx[j], c2 = ws.cy[j];
          fprota(cs, sn, xi1, c1);
          fprota(cs, sn, xi2, c2);
          ws.cx[j] = c1;
          ws.cy[j] = c2;
          if (j == nk1) break;
          int i2 = k1;
          if (j > n8) i2 = nk1 - j;
          // columns 2 .. i2 + 1 of band row j against h(2 .. i2 + 1); the row then moves one place to the left
          double g1 = ws.g[j][1], g2 = ws.g[j][2], g3 = ws.g[j][3], g4 = ws.g[j][4];
          if (1 <= i2) {
            fprota(cs, sn, hh2, g1);
            ws.g[j][1] = g1;
          }
          if (2 <= i2) {
            fprota(cs, sn, hh3, g2);
            ws.g[j][2] = g2;
          }
          if (3 <= i2) {
            fprota(cs, sn, hh4, g3);
            ws.g[j][3] = g3;
          }
          if (4 <= i2) {
            fprota(cs, sn, hh5, g4);
            ws.g[j][4] = g4;
          }
          // h(i) = h(i + 1) for i <= i2; h(i2 + 1) = 0
          hh1 = (1 <= i2) ? hh2 : hh1;
          hh2 = (2 <= i2) ? hh3 : hh2;
          hh3 = (3 <= i2) ? hh4 : hh3;
          hh4 = (4 <= i2) ? hh5 : hh4;
          if (i2 + 1 == 1) hh1 = 0.;
          if (i2 + 1 == 2) hh2 = 0.;
          if (i2 + 1 == 3) hh3 = 0.;
          if (i2 + 1 == 4) hh4 = 0.;
          if (i2 + 1 == 5) hh5 = 0.;
        }
      }
      // fpback, band width k2, both coordinates
      {
        double c1x = 0, c2x = 0, c3x = 0, c4x = 0, c1y = 0, c2y = 0, c3y = 0, c4y = 0;
        for (int i = nk1; i >= 1; i--) {
          const double b1 = ws.g[i][0], b2 = ws.g[i][1], b3 = ws.g[i][2], b4 = ws.g[i][3], b5 = ws.g[i][4];
          const int i1 = (nk1 - i) < (k2 - 1) ? (nk1 - i) : (k2 - 1);
          double s1 = ws.cx[i], s2 = ws.cy[i];
          if (1 <= i1) {
            s1 = s1 - c1x * b2;
            s2 = s2 - c1y * b2;
          }
          if (2 <= i1) {
            s1 = s1 - c2x * b3;
            s2 = s2 - c2y * b3;
          }
          if (3 <= i1) {
            s1 = s1 - c3x * b4;
            s2 = s2 - c3y * b4;
          }
          if (4 <= i1) {
            s1 = s1 - c4x * b5;
            s2 = s2 - c4y * b5;
          }
          const double vx = s1 / b1, vy = s2 / b1;
          ws.cx[i] = vx;
          ws.cy[i] = vy;
          c4x = c3x;
          c3x = c2x;
          c2x = c1x;
          c1x = vx;
          c4y = c3y;
          c3y = c2y;
          c2y = c1y;
          c1y = vy;
        }
      }
      // f(p): accumulation in data order
      fp = 0.;
      {
        int l = k1;
        IntervalCtx ic;
        load_interval(ws.t, l, ic);
        double c0x = ws.cx[l - 3], c1x = ws.cx[l - 2], c2x = ws.cx[l - 1], c3x = ws.cx[l];
        double c0y = ws.cy[l - 3], c1y = ws.cy[l - 2], c2y = ws.cy[l - 1], c3y = ws.cy[l];
        // four points per round: the next round's points are fetched while this round's terms are formed
        double un[4], xn[4], yn[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int i2 = q < m ? q : m - 1;
          un[q] = U[(size_t)i2 * STRIDE];
          xn[q] = X[(size_t)i2 * STRIDE];
          yn[q] = Y[(size_t)i2 * STRIDE];
        }
        for (int i0 = 0; i0 < m; i0 += 4) {
          double uc[4], xc[4], yc[4];
#pragma unroll
          for (int q = 0; q < 4; q++) {
            uc[q] = un[q];
            xc[q] = xn[q];
            yc[q] = yn[q];
          }
#pragma unroll
          for (int q = 0; q < 4; q++) {
            int i2 = i0 + 4 + q;
            i2 = i2 < m ? i2 : m - 1;
            un[q] = U[(size_t)i2 * STRIDE];
            xn[q] = X[(size_t)i2 * STRIDE];
            yn[q] = Y[(size_t)i2 * STRIDE];
          }
#pragma unroll
          for (int q = 0; q < 4; q++) {
            if (i0 + q < m) {
              const double ui = uc[q];
              while (!(ui < ic.tp1 || l == nk1)) {
                l++;
                load_interval(ws.t, l, ic);
                c0x = ws.cx[l - 3];
                c1x = ws.cx[l - 2];
                c2x = ws.cx[l - 1];
                c3x = ws.cx[l];
                c0y = ws.cy[l - 3];
                c1y = ws.cy[l - 2];
                c2y = ws.cy[l - 1];
                c3y = ws.cy[l];
              }
              double h1, h2, h3, h4;
              bool unused = true;
              basis_lane<false>(ic, ui, h1, h2, h3, h4, unused);
              double term = 0.0;
              {
                double fac = 0.0;
                fac = fac + c0x * h1;
                fac = fac + c1x * h2;
                fac = fac + c2x * h3;
                fac = fac + c3x * h4;
                const double dv = 1.0 * (fac - xc[q]);
                term = term + dv * dv;
              }
              {
                double fac = 0.0;
                fac = fac + c0y * h1;
                fac = fac + c1y * h2;
                fac = fac + c2y * h3;
                fac = fac + c3y * h4;
                const double dv = 1.0 * (fac - yc[q]);
                term = term + dv * dv;
              }
              fp = fp + term;
            }
          }
        }
      }
      fpms = fp - s;
      if (fabs(fpms) < acc) break;
      if (iter == maxit) {
        ier = 3;
        break;
      }
      const double p2 = p, f2 = fpms;
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
        double pn;
        if (p3 > 0.) {
          const double h1 = f1 * (f2 - f3);
          const double h2 = f2 * (f3 - f1);
          const double h3 = f3 * (f1 - f2);
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
  }
  if (bad) R.status = ST_RETRY;
  R.n = n;
  R.ier = ier;
  R.fp = fp;
  return R;
}

}  // namespace fsdp
