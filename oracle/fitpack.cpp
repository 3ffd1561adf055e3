// TEST INFRASTRUCTURE (oracle) — see fitpack.h for scope and provenance.
// Restates Dierckx's FITPACK routines parcur/fppara/fpknot/fpdisc/fprati/fpgivs/fprota/
// fpback/fpbspl/splev for idim=2, unit weights, iopt=0 — the only mode the reference
// reaches (utils/spline_fit.py:95-128).  Arrays are 1-based like the published algorithm.
#include "fitpack.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace fsdo {
namespace {

const int IDIM = 2;
const double TOL = 0.001;  // parcur.f: tol = 0.1e-02
const int MAXIT = 20;      // parcur.f: maxit = 20

// fpbspl: the (k+1) non-zero B-splines of degree k at t(l) <= x < t(l+1), de Boor-Cox,
// with SciPy's zero-weight rule for coincident knots.
void fpbspl(const double* t, int k, double x, int l, double* h /*1..6*/) {
  double hh[20];
  h[1] = 1.0;
  for (int j = 1; j <= k; j++) {
    for (int i = 1; i <= j; i++) hh[i] = h[i];
    h[1] = 0.0;
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

inline void fpgivs(double piv, double& ww, double& cs, double& sn) {
  double store = std::fabs(piv);
  double dd;
  if (store >= ww) {
    double r = ww / piv;
    dd = store * std::sqrt(1.0 + r * r);
  } else {
    double r = piv / ww;
    dd = ww * std::sqrt(1.0 + r * r);
  }
  cs = ww / dd;
  sn = piv / dd;
  ww = dd;
}

inline void fprota(double cs, double sn, double& a, double& b) {
  double stor1 = a, stor2 = b;
  b = cs * stor2 + sn * stor1;
  a = cs * stor1 - sn * stor2;
}

double fprati(double& p1, double& f1, double p2, double f2, double& p3, double& f3) {
  double p;
  if (p3 > 0.) {
    double h1 = f1 * (f2 - f3);
    double h2 = f2 * (f3 - f1);
    double h3 = f3 * (f1 - f2);
    p = -(p1 * p2 * h3 + p2 * p3 * h1 + p3 * p1 * h2) / (p1 * h1 + p2 * h2 + p3 * h3);
  } else {
    p = (p1 * (f1 - f3) * f2 - p2 * (f2 - f3) * f1) / ((f1 - f2) * f3);
  }
  if (f2 < 0.) {
    p3 = p2;
    f3 = f2;
  } else {
    p1 = p2;
    f1 = f2;
  }
  return p;
}

struct Band {
  int cols;
  std::vector<double> d;
  Band(int rows, int cols_) : cols(cols_), d((size_t)(rows + 1) * (cols_ + 1), 0.0) {}
  inline double& operator()(int i, int j) { return d[(size_t)i * (cols + 1) + j]; }
};

// fpback: back-substitution for the banded upper-triangular system a*c = z (bandwidth k)
void fpback(Band& a, const double* z, int n, int k, double* c) {
  int k1 = k - 1;
  c[n] = z[n] / a(n, 1);
  int i = n - 1;
  if (i == 0) return;
  for (int j = 2; j <= n; j++) {
    double store = z[i];
    int i1 = k1;
    if (j <= k1) i1 = j - 1;
    int m = i;
    for (int l = 1; l <= i1; l++) {
      m = m + 1;
      store = store - c[m] * a(i, l + 1);
    }
    c[i] = store / a(i, 1);
    i = i - 1;
  }
}

// fpdisc: discontinuity jumps of the k-th derivative of the B-splines at the interior knots
void fpdisc(const double* t, int n, int k2, Band& b) {
  double h[13];
  int k1 = k2 - 1;
  int k = k1 - 1;
  int nk1 = n - k1;
  int nrint = nk1 - k;
  double an = nrint;
  double fac = an / (t[nk1 + 1] - t[k1]);
  for (int l = k2; l <= nk1; l++) {
    int lmk = l - k1;
    for (int j = 1; j <= k1; j++) {
      int ik = j + k1;
      int lj = l + j;
      int lk = lj - k2;
      h[j] = t[l] - t[lk];
      h[ik] = t[l] - t[lj];
    }
    int lp = lmk;
    for (int j = 1; j <= k2; j++) {
      int jk = j;
      double prod = h[j];
      for (int i = 1; i <= k; i++) {
        jk = jk + 1;
        prod = prod * h[jk] * fac;
      }
      int lk = lp + k1;
      b(lmk, j) = (t[lk] - t[lp]) / prod;
      lp = lp + 1;
    }
  }
}

// fpknot: add one knot inside the interval with the largest residual sum
void fpknot(const double* x, double* t, int& n, double* fpint, int* nrdata, int& nrint, int istart) {
  int k = (n - nrint - 1) / 2;
  double fpmax = 0.;
  int jbegin = istart;
  int number = 0, maxpt = 0, maxbeg = 0;
  for (int j = 1; j <= nrint; j++) {
    int jpoint = nrdata[j];
    if (!(fpmax >= fpint[j] || jpoint == 0)) {
      fpmax = fpint[j];
      number = j;
      maxpt = jpoint;
      maxbeg = jbegin;
    }
    jbegin = jbegin + jpoint + 1;
  }
  int ihalf = maxpt / 2 + 1;
  int nrx = maxbeg + ihalf;
  int next = number + 1;
  if (next <= nrint) {
    for (int j = next; j <= nrint; j++) {
      int jj = next + nrint - j;
      fpint[jj + 1] = fpint[jj];
      nrdata[jj + 1] = nrdata[jj];
      int jk = jj + k;
      t[jk + 1] = t[jk];
    }
  }
  nrdata[number] = ihalf - 1;
  nrdata[next] = maxpt - ihalf;
  double am = maxpt;
  double an = nrdata[number];
  fpint[number] = fpmax * an / am;
  an = nrdata[next];
  fpint[next] = fpmax * an / am;
  int jk = next + k;
  t[jk] = x[nrx];
  n = n + 1;
  nrint = nrint + 1;
}

}  // namespace

bool parcur_fit(const double* u0, const double* x0, const double* y0, int m, int k, double s, Spline& out) {
  // --- parcur.f input checks (ier=10 -> scipy raises ValueError) ---
  if (k < 1 || k > 5) return false;
  const int k1 = k + 1, k2 = k1 + 1;
  const int nmin = 2 * k1;
  const int nest = m + 2 * k;  // scipy: nest = m + 2*k for task 0
  if (m < k1 || nest < nmin) return false;
  for (int i = 1; i < m; i++)
    if (!(u0[i - 1] < u0[i])) return false;  // u(i-1).ge.u(i) -> error
  if (s < 0.) return false;
  const double ub = u0[0], ue = u0[m - 1];

  // 1-based views
  std::vector<double> u(m + 1), xx((size_t)IDIM * m + 1);
  for (int i = 1; i <= m; i++) {
    u[i] = u0[i - 1];
    xx[(i - 1) * IDIM + 1] = x0[i - 1];
    xx[(i - 1) * IDIM + 2] = y0[i - 1];
  }
  const int nc = IDIM * nest;
  std::vector<double> t(nest + 2, 0.0), c(nc + 2, 0.0), z(nc + 2, 0.0), fpint(nest + 2, 0.0);
  std::vector<int> nrdata(nest + 2, 0);
  Band a(nest, k1), b(nest, k2), g(nest, k2), q(m, k1);
  double h[8], xi[3];

  // --- fppara ---
  // SciPy's fppara.f sets con1/con9/con4 from single-precision literals (0.1e0, 0.9e0, 0.4e-01)
  // stored into real*8 variables; with these values the restatement is bit-identical to
  // scipy.interpolate.splprep (tests/test_oracle_fitpack.py), with 0.1/0.9/0.04 it is not.
  const double one = 1.0, con1 = (double)0.1f, con9 = (double)0.9f, con4 = (double)0.04f, half = 0.5;
  int n = 0, ier = 0, nplus = 0, nrint = 0, nk1 = 0;
  double fp = 0, fpold = 0, fp0 = 0, fpms = 0, acc = 0;
  int nmax = 0;

  acc = TOL * s;
  nmax = m + k1;
  bool goto_interp_knots = false;
  if (!(s > 0.)) {
    // s == 0: interpolating curve (not reached by the reference: s is 0.2 or 0.01)
    n = nmax;
    if (nmax > nest) {
      out.ier = 1;
      return true;
    }
    goto_interp_knots = true;
  } else {
    n = nmin;
    fpold = 0.;
    nplus = 0;
    nrdata[1] = m - 2;
  }

  bool done = false;       // label 440
  bool to_part2 = false;   // label 250
  while (!done && !to_part2) {
    if (goto_interp_knots) {
      // label 10: knots for interpolation
      goto_interp_knots = false;
      int mk1 = m - k1;
      if (mk1 != 0) {
        int k3 = k / 2;
        int i = k2;
        int j = k3 + 2;
        if (k3 * 2 == k) {
          for (int l = 1; l <= mk1; l++) {
            t[i] = (u[j] + u[j - 1]) * half;
            i++;
            j++;
          }
        } else {
          for (int l = 1; l <= mk1; l++) {
            t[i] = u[j];
            i++;
            j++;
          }
        }
      }
    }
    // label 60: main loop for the different sets of knots
    bool restart = false;
    for (int iter = 1; iter <= m && !restart; iter++) {
      if (n == nmin) ier = -2;
      nrint = n - nmin + 1;
      nk1 = n - k1;
      {
        int i = n;
        for (int j = 1; j <= k1; j++) {
          t[j] = ub;
          t[i] = ue;
          i--;
        }
      }
      fp = 0.;
      for (int i = 1; i <= nc; i++) z[i] = 0.;
      for (int i = 1; i <= nk1; i++)
        for (int j = 1; j <= k1; j++) a(i, j) = 0.;
      int l = k1;
      int jj = 0;
      for (int it = 1; it <= m; it++) {
        double ui = u[it];
        double wi = 1.0;
        for (int j = 1; j <= IDIM; j++) {
          jj++;
          xi[j] = xx[jj] * wi;
        }
        while (!(ui < t[l + 1] || l == nk1)) l++;
        fpbspl(t.data(), k, ui, l, h);
        for (int i = 1; i <= k1; i++) {
          q(it, i) = h[i];
          h[i] = h[i] * wi;
        }
        int j = l - k1;
        for (int i = 1; i <= k1; i++) {
          j++;
          double piv = h[i];
          if (piv == 0.) continue;
          double cs, sn;
          fpgivs(piv, a(j, 1), cs, sn);
          int j1 = j;
          for (int j2 = 1; j2 <= IDIM; j2++) {
            fprota(cs, sn, xi[j2], z[j1]);
            j1 += n;
          }
          if (i == k1) break;
          int i2 = 1;
          int i3 = i + 1;
          for (int i1 = i3; i1 <= k1; i1++) {
            i2++;
            fprota(cs, sn, h[i1], a(j, i2));
          }
        }
        for (int j2 = 1; j2 <= IDIM; j2++) fp = fp + xi[j2] * xi[j2];
      }
      if (ier == -2) fp0 = fp;
      fpint[n] = fp0;
      fpint[n - 1] = fpold;
      nrdata[n] = nplus;
      {
        int j1 = 1;
        for (int j2 = 1; j2 <= IDIM; j2++) {
          fpback(a, &z[j1 - 1], nk1, k1, &c[j1 - 1]);
          j1 += n;
        }
      }
      fpms = fp - s;
      if (std::fabs(fpms) < acc) {
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
        done = true;
        break;
      }
      if (ier == 0) {
        int npl1 = nplus * 2;
        double rn = nplus;
        if (fpold - fp > acc) npl1 = (int)(rn * fpms / (fpold - fp));
        nplus = std::min(nplus * 2, std::max(std::max(npl1, nplus / 2), 1));
      } else {
        nplus = 1;
        ier = 0;
      }
      fpold = fp;
      // residual sums per knot interval
      double fpart = 0.;
      int i = 1;
      l = k2;
      int nw = 0;
      jj = 0;
      for (int it = 1; it <= m; it++) {
        if (!(u[it] < t[l] || l > nk1)) {
          nw = 1;
          l++;
        }
        double term = 0.;
        int l0 = l - k2;
        for (int j2 = 1; j2 <= IDIM; j2++) {
          double fac = 0.;
          int j1 = l0;
          for (int j = 1; j <= k1; j++) {
            j1++;
            fac = fac + c[j1] * q(it, j);
          }
          jj++;
          double d = 1.0 * (fac - xx[jj]);
          term = term + d * d;
          l0 += n;
        }
        // fppara.f: fpart = fpart+term; if(new.eq.0) go to 180; store = term*half; fpint(i) = fpart-store; ... —
        // (fpart + term) - term/2, not fpart + term/2: the two round differently, and the knot selection compares
        // these sums (pinned by tests/golden/params_path.npz frame 119 and splines.npz)
        fpart = fpart + term;
        if (nw != 0) {
          double store = term * half;
          fpint[i] = fpart - store;
          i++;
          fpart = store;
          nw = 0;
        }
      }
      fpint[nrint] = fpart;
      for (int lq = 1; lq <= nplus; lq++) {
        fpknot(u.data(), t.data(), n, fpint.data(), nrdata.data(), nrint, 1);
        if (n == nmax) {
          goto_interp_knots = true;
          restart = true;
          break;
        }
        if (n == nest) break;
      }
      // restart the computations with the new set of knots (next iter)
    }
    if (!restart && !done && !to_part2) {
      // fell out of "do 200 iter=1,m" without a decision: FITPACK continues at label 250
      to_part2 = true;
    }
  }

  if (to_part2 && ier != -2) {
    // --- part 2: smoothing spline sp(u), root of f(p) = s ---
    fpdisc(t.data(), n, k2, b);
    double p1 = 0., f1 = fp0 - s, p3 = -one, f3 = fpms, p = 0.;
    for (int i = 1; i <= nk1; i++) p = p + a(i, 1);
    double rn = nk1;
    p = rn / p;
    int ich1 = 0, ich3 = 0;
    int n8 = n - nmin;
    bool finished = false;
    for (int iter = 1; iter <= MAXIT; iter++) {
      double pinv = one / p;
      for (int i = 1; i <= nc; i++) c[i] = z[i];
      for (int i = 1; i <= nk1; i++) {
        g(i, k2) = 0.;
        for (int j = 1; j <= k1; j++) g(i, j) = a(i, j);
      }
      for (int it = 1; it <= n8; it++) {
        for (int i = 1; i <= k2; i++) h[i] = b(it, i) * pinv;
        for (int j = 1; j <= IDIM; j++) xi[j] = 0.;
        for (int j = it; j <= nk1; j++) {
          double piv = h[1];
          double cs, sn;
          fpgivs(piv, g(j, 1), cs, sn);
          int j1 = j;
          for (int j2 = 1; j2 <= IDIM; j2++) {
            fprota(cs, sn, xi[j2], c[j1]);
            j1 += n;
          }
          if (j == nk1) break;
          int i2 = k1;
          if (j > n8) i2 = nk1 - j;
          for (int i = 1; i <= i2; i++) {
            int i1 = i + 1;
            fprota(cs, sn, h[i1], g(j, i1));
            h[i] = h[i1];
          }
          h[i2 + 1] = 0.;
        }
      }
      {
        int j1 = 1;
        for (int j2 = 1; j2 <= IDIM; j2++) {
          fpback(g, &c[j1 - 1], nk1, k2, &c[j1 - 1]);
          j1 += n;
        }
      }
      fp = 0.;
      int l = k2;
      int jj = 0;
      for (int it = 1; it <= m; it++) {
        if (!(u[it] < t[l] || l > nk1)) l++;
        int l0 = l - k2;
        double term = 0.;
        for (int j2 = 1; j2 <= IDIM; j2++) {
          double fac = 0.;
          int j1 = l0;
          for (int j = 1; j <= k1; j++) {
            j1++;
            fac = fac + c[j1] * q(it, j);
          }
          jj++;
          double d = fac - xx[jj];
          term = term + d * d;
          l0 += n;
        }
        fp = fp + term * (1.0 * 1.0);
      }
      fpms = fp - s;
      if (std::fabs(fpms) < acc) {
        finished = true;
        break;
      }
      if (iter == MAXIT) {
        ier = 3;
        finished = true;
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
          finished = true;
          break;
        }
        p = fprati(p1, f1, p2, f2, p3, f3);
      }
    }
    (void)finished;
  }

  out.k = k;
  out.n = n;
  out.ier = ier;
  out.fp = fp;
  out.t.assign(t.begin() + 1, t.begin() + 1 + n);
  out.cx.assign(c.begin() + 1, c.begin() + 1 + n);
  out.cy.assign(c.begin() + 1 + n, c.begin() + 1 + 2 * n);
  return true;
}

void splev_points(const Spline& sp, const double* xev, long m, double* out_x, double* out_y) {
  const int k = sp.k, n = sp.n;
  const int k1 = k + 1, k2 = k1 + 1, nk1 = n - k1;
  std::vector<double> t(n + 2);
  for (int i = 1; i <= n; i++) t[i] = sp.t[i - 1];
  double h[8];
  int l = k1, l1 = l + 1;
  for (long i = 0; i < m; i++) {
    double arg = xev[i];
    // ext=0: extrapolate from the end intervals
    while (!(arg >= t[l] || l1 == k2)) {
      l1 = l;
      l = l - 1;
    }
    while (!(arg < t[l1] || l == nk1)) {
      l = l1;
      l1 = l + 1;
    }
    fpbspl(t.data(), k, arg, l, h);
    double sx = 0., sy = 0.;
    int ll = l - k1;
    for (int j = 1; j <= k1; j++) {
      ll++;
      sx = sx + sp.cx[ll - 1] * h[j];
      sy = sy + sp.cy[ll - 1] * h[j];
    }
    out_x[i] = sx;
    out_y[i] = sy;
  }
}

}  // namespace fsdo

// ---- C entry points used by the tests to pin the restatement against SciPy ----
extern "C" {

// returns 0 ok, 1 if scipy would raise ValueError.  t_out/cx_out/cy_out need m+2k+2 slots.
int fsdo_splprep(const double* u, const double* x, const double* y, int m, int k, double s, double* t_out,
                 double* cx_out, double* cy_out, int* n_out, int* ier_out, double* fp_out) {
  fsdo::Spline sp;
  if (!fsdo::parcur_fit(u, x, y, m, k, s, sp)) return 1;
  *n_out = sp.n;
  *ier_out = sp.ier;
  *fp_out = sp.fp;
  for (int i = 0; i < sp.n; i++) {
    t_out[i] = sp.t[i];
    cx_out[i] = sp.cx[i];
    cy_out[i] = sp.cy[i];
  }
  return 0;
}

void fsdo_splev(const double* t, const double* cx, const double* cy, int n, int k, const double* u_eval, long n_eval,
                double* out_x, double* out_y) {
  fsdo::Spline sp;
  sp.k = k;
  sp.n = n;
  sp.t.assign(t, t + n);
  sp.cx.assign(cx, cx + n);
  sp.cy.assign(cy, cy + n);
  fsdo::splev_points(sp, u_eval, n_eval, out_x, out_y);
}
}
