// TEST INFRASTRUCTURE (oracle) — see np_compat.h header.
//
// Restatement of the smoothing-spline arithmetic the reference reaches through SciPy:
//   utils/spline_fit.py:117  splprep(trace.T, s=..., k=..., u=u_fit, per=False)
//   utils/spline_fit.py:61   splev(u_eval, tck, der=0)
// SciPy is a third-party dependency (unpinned in the reference's pyproject.toml:9 /
// requirements.txt:4; the build container holds scipy 1.15.3 as compiled FITPACK only).
// The algorithm restated here is P. Dierckx's published FITPACK: parcur -> fppara
// (knot placement fpknot, Givens QR fpgivs/fprota, back-substitution fpback, jump matrix
// fpdisc, rational root step fprati), B-spline basis fpbspl, evaluation splev.
// Pinned by tests against scipy.interpolate.splprep/splev outputs (tests/golden).
#pragma once
#include <vector>

namespace fsdo {

struct Spline {
  int k = 3;               // degree
  int n = 0;               // number of knots
  int ier = 0;             // FITPACK ier (0, -1, -2 ok; 1,2,3 warnings scipy still returns)
  double fp = 0;           // weighted sum of squared residuals
  std::vector<double> t;   // knots (n)
  std::vector<double> cx;  // coefficients dim 0 (n-k-1 meaningful)
  std::vector<double> cy;  // coefficients dim 1
};

// parcur with iopt=0, ipar=1, idim=2, w=1, ub=u[0], ue=u[m-1], nest=m+2k (what
// scipy.interpolate.splprep passes for task=0, _fitpack_impl.py:160-168).
// Returns false where scipy raises ValueError (ier=10: u not strictly increasing, ...).
bool parcur_fit(const double* u, const double* x, const double* y, int m, int k, double s, Spline& out);

// splev(der=0, ext=0) for both coordinates at one parameter value (sequential-search
// state `l` is carried by the caller exactly like FITPACK's loop over x(i)).
void splev_points(const Spline& sp, const double* u_eval, long n_eval, double* out_x, double* out_y);

}  // namespace fsdo
