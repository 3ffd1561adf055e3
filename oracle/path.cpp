// TEST INFRASTRUCTURE (oracle) — calculate_path step, scalar restatement with NumPy semantics.
// Reference: calculate_path/core_calculate_path.py, calculate_path/path_parameterization.py,
// calculate_path/path_calculator_helpers.py, utils/spline_fit.py, utils/math_utils.py:579-646.
// Semantics for independent frames: a fresh PathPlanner per frame, i.e. previous_paths[-1] is
// the constant initial path (core_calculate_path.py:103-107).
#include <cmath>
#include <vector>
#include <algorithm>
#include <mutex>
#include <cstdio>
#include <cstdlib>

#include "oracle_internal.h"
#include "det_math.h"

namespace fsdo {

// 0: libm (what NumPy calls; pins the oracle to the reference's golden vectors)
// 1: deterministic correctly-rounded sin/cos/atan2 (det_math.h) — what the HIP kernels use, for exact GPU parity
int g_math_mode = 0;
double m_atan2(double y, double x) { return g_math_mode ? detm::det_atan2(y, x) : std::atan2(y, x); }
double m_cos(double a) { return g_math_mode ? detm::det_cos(a) : std::cos(a); }
double m_sin(double a) { return g_math_mode ? detm::det_sin(a) : std::sin(a); }

// config.py:48 / :55-59, overridable (fsdo_set_params)
#define SMOOTHING (g_prm.smoothing)
#define PREDICT_EVERY (g_prm.predict_every)
#define MAX_DIST_VALID_PATH (g_prm.maximal_distance_for_valid_path)
#define MPC_PATH_LENGTH (g_prm.mpc_path_length)
OParams g_prm;
#define HORIZON (g_prm.horizon)  // mpc_prediction_horizon (<= FSDO_PATH_POINTS, the stride of the path arrays)

// utils/math_utils.py:579-646 circle_fit (hyper fit); returns (cx, cy, r)
void circle_fit(const Pts& p, double& ocx, double& ocy, double& orad) {
  const int n = (int)p.size();
  std::vector<double> X(n), Y(n), tmp(n);
  for (int i = 0; i < n; i++) {
    X[i] = p[i].x;
    Y[i] = p[i].y;
  }
  double xm = np_sum(X) / n, ym = np_sum(Y) / n;
  std::vector<double> Xi(n), Yi(n), Zi(n);
  for (int i = 0; i < n; i++) {
    Xi[i] = X[i] - xm;
    Yi[i] = Y[i] - ym;
    Zi[i] = Xi[i] * Xi[i] + Yi[i] * Yi[i];
  }
  auto msum = [&](const std::vector<double>& a, const std::vector<double>& b) {
    for (int i = 0; i < n; i++) tmp[i] = a[i] * b[i];
    return np_sum(tmp) / n;
  };
  double Mxy = msum(Xi, Yi), Mxx = msum(Xi, Xi), Myy = msum(Yi, Yi);
  double Mxz = msum(Xi, Zi), Myz = msum(Yi, Zi), Mzz = msum(Zi, Zi);
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
    if (x_new == x || !std::isfinite(x_new)) break;
    double y_new = A0 + x_new * (A1 + x_new * (A2 + 4.0 * x_new * x_new));
    if (std::fabs(y_new) >= std::fabs(y)) break;
    x = x_new;
    y = y_new;
  }
  double det = x * x - x * Mz + Cov_xy;
  double Xc = (Mxz * (Myy - x) - Myz * Mxy) / det / 2.0;
  double Yc = (Myz * (Mxx - x) - Mxz * Mxy) / det / 2.0;
  ocx = Xc + xm;
  ocy = Yc + ym;
  // `X_center**2` of a NumPy scalar is C pow(x, 2.0), and glibc's pow is not correctly rounded for every argument (nor the
  // same in its FMA and non-FMA variants): on ~1 % of frames one window's radius — hence one curvature sample — differs from
  // the product x * x in its last bit.  Host-libm mode follows the reference on this machine; det mode (what the kernels
  // compute) takes the correctly rounded product.
  volatile double two = 2.0;  // (a literal exponent would be folded into x * x by the compiler: this must be libm's pow)
  orad = g_math_mode ? std::sqrt(std::fabs(Xc * Xc + Yc * Yc + Mz)) : std::sqrt(std::fabs(std::pow(Xc, two) + std::pow(Yc, two) + Mz));
}

// utils/spline_fit.py:95-128 fit + :46-63 predict(der=0, max_u).  Throws PyValueError where
// splprep raises.  A trace with < 2 points yields the NullSplineEvaluator (empty prediction).
struct Fitted {
  bool null = false;
  Spline sp;
  double max_u = 0;
  double predict_every = 0;
};

// optional capture of every fit of a frame, in call order (fsdo_plan_frame_capture: per-stage intermediates for the tests)
thread_local std::vector<Spline>* g_fit_capture = nullptr;

static Fitted spline_fit(const Pts& trace, double smoothing, double predict_every, int max_deg) {
  Fitted f;
  f.predict_every = predict_every;
  const int m = (int)trace.size();
  if (m < 2) {
    f.null = true;
    return f;
  }
  int k = std::min(std::max(m - 1, 1), max_deg);  // np.clip(len(trace) - 1, 1, max_deg)
  std::vector<double> u(m), x(m), y(m);
  double acc = 0.0;  // np.cumsum: sequential
  u[0] = 0.0;
  for (int i = 0; i < m; i++) {
    x[i] = trace[i].x;
    y[i] = trace[i].y;
    if (i > 0) {
      double dx = trace[i].x - trace[i - 1].x, dy = trace[i].y - trace[i - 1].y;
      acc += std::sqrt(dx * dx + dy * dy);
      u[i] = acc;
    }
  }
  if (!parcur_fit(u.data(), x.data(), y.data(), m, k, smoothing, f.sp)) throw PyValueError{1};
  if (g_fit_capture) g_fit_capture->push_back(f.sp);
  f.max_u = u[m - 1];
  return f;
}

static Pts spline_predict(const Fitted& f, double max_u, std::vector<double>* u_out = nullptr) {
  Pts out;
  if (f.null) return out;
  long n = arange_len(max_u, f.predict_every);
  std::vector<double> ue(n), ox(n), oy(n);
  for (long i = 0; i < n; i++) ue[i] = (double)i * f.predict_every;
  splev_points(f.sp, ue.data(), n, ox.data(), oy.data());
  out.resize(n);
  for (long i = 0; i < n; i++) out[i] = Vec2{ox[i], oy[i]};
  if (u_out) *u_out = ue;
  return out;
}

// calculate_path/path_parameterization.py:49-93 calculate_path_curvature (open path)
static std::vector<double> path_curvature(const Pts& path, int window_size) {
  const int L = (int)path.size();
  const int half = window_size / 2;
  std::vector<double> curv(L, 0.0);
  for (int i = 0; i < L; i++) {
    std::vector<int> win(window_size);
    for (int q = 0; q < window_size; q++) win[q] = (((q - half + i) % L) + L) % L;
    int cut = -1;
    for (int q = 0; q + 1 < window_size; q++)
      if (win[q + 1] - win[q] != 1) {
        cut = q + 1;
        break;
      }
    if (cut >= 0) {
      if (i < window_size)
        win.erase(win.begin(), win.begin() + cut);
      else
        win.resize(cut);
    }
    Pts pts(win.size());
    for (size_t q = 0; q < win.size(); q++) pts[q] = path[win[q]];
    double cx, cy, r;
    circle_fit(pts, cx, cy, r);
    r = py_min(py_max(r, 1.0), 3000.0);
    double c = 1 / r;
    int np_ = (int)pts.size();
    int i0 = 0, i1 = (int)(np_ / 2), i2 = np_ - 1;
    double hm[3][3] = {{1.0, pts[i0].x, pts[i0].y}, {1.0, pts[i1].x, pts[i1].y}, {1.0, pts[i2].x, pts[i2].y}};
    double sg = det3_lu(hm);
    curv[i] = c * np_sign(sg);
  }
  return curv;
}

// scipy.ndimage.uniform_filter1d(x, size, mode="nearest"), origin 0: window [i - size/2, i - size/2 + size)
// accumulated as a running sum like the library does.
static std::vector<double> uniform_filter_nearest(const std::vector<double>& in, int size) {
  const int L = (int)in.size();
  std::vector<double> out(L);
  if (L == 0) return out;
  int s1 = size / 2, s2 = size - s1 - 1;
  auto at = [&](int j) { return in[std::min(std::max(j, 0), L - 1)]; };
  double tmp = 0.0;
  for (int j = -s1; j <= s2; j++) tmp += at(j);
  out[0] = tmp / size;
  for (int i = 1; i < L; i++) {
    tmp += at(i + s2) - at(i - 1 - s1);
    out[i] = tmp / size;
  }
  return out;
}

// calculate_path/path_parameterization.py:297-328 parameterize_path(path_is_closed=False)
static void parameterize_path(const Pts& path, double out[][4]) {
  // _refit_spline :125-161
  const int n = (int)path.size();
  std::vector<double> d(std::max(n - 1, 0));
  for (int i = 0; i + 1 < n; i++) {
    double dx = path[i + 1].x - path[i].x, dy = path[i + 1].y - path[i].y;
    d[i] = std::sqrt(dx * dx + dy * dy);
  }
  double path_length = np_sum(d);
  long n10 = std::min<long>(10, (long)d.size());
  double mean_pd = np_sum(d.data(), n10) / (double)n10;  // mean of empty -> nan
  double predict_every = path_length / HORIZON / 3;
  int skip;
  {
    double q = predict_every / mean_pd;
    if (std::isnan(q))
      skip = 1;  // int(nan) -> ValueError -> skip_factor = 1
    else if (std::isinf(q))
      throw RefUndefined{FSDO_REF_UNDEFINED_PATH};  // OverflowError propagates
    else
      skip = std::max((int)q, 1);
  }
  Pts skipped;
  for (int i = 0; i < n; i += skip) skipped.push_back(path[i]);
  Fitted fit = spline_fit(skipped, 0.01, predict_every, 3);  // path_parameterization.py:155-157: max_deg = 3 here, whatever the configuration
  // _calculate_path_curvature :163-193
  std::vector<double> ue;
  Pts pts = spline_predict(fit, fit.max_u, &ue);
  int L = (int)pts.size();
  if (fit.null || L == 0) throw RefUndefined{FSDO_REF_UNDEFINED_PATH};
  int window = std::min(L / 5, 30);
  if (window % 2 == 0) window += 1;
  std::vector<double> curv = path_curvature(pts, window);
  int fsize = std::max(2, window / 2);
  std::vector<double> filt = uniform_filter_nearest(curv, fsize);
  // _sample_path_parameters_for_prediction_horizon :252-295; np.linspace(0, L-1, 40, dtype=int)
  long idx[FSDO_PATH_POINTS];
  double step = ((double)(L - 1) - 0.0) / (double)(HORIZON - 1);
  for (int i = 0; i < HORIZON; i++) {
    double v = (double)i * step + 0.0;
    if (i == HORIZON - 1) v = (double)(L - 1);
    idx[i] = HORIZON == 1 ? 0 : (long)std::floor(v);  // np.linspace(0, L - 1, 1) = [0.]
  }
  for (int i = 1; i < HORIZON; i++)
    if (idx[i] == idx[i - 1]) throw PyValueError{2};  // "Indices of resampled path appear twice"
  for (int i = 0; i < HORIZON; i++) {
    out[i][0] = ue[idx[i]];
    out[i][1] = pts[idx[i]].x;
    out[i][2] = pts[idx[i]].y;
    out[i][3] = filt[idx[i]];
  }
  for (int i = HORIZON; i < FSDO_PATH_POINTS; i++) out[i][0] = out[i][1] = out[i][2] = out[i][3] = NAN;  // rows the reference's path does not have
}

// core_calculate_path.py:430-457 connect_path_to_car
static Pts connect_path_to_car(const Pts& path, Vec2 pos, Vec2 dir) {
  double d = norm2(pos.x - path[0].x, pos.y - path[0].y);
  double cx = path[0].x - pos.x, cy = path[0].y - pos.y;
  double ang = vec_angle_between(cx, cy, dir.x, dir.y);
  if (d < 0.5 || ang > PI / 2) return path;
  double nrm = norm2(cx, cy);
  Vec2 np_{pos.x + (cx / nrm) * 0.2, pos.y + (cy / nrm) * 0.2};
  Pts out;
  out.push_back(np_);
  out.insert(out.end(), path.begin(), path.end());
  return out;
}

// core_calculate_path.py:261-334 extend_path
static Pts extend_path(const Pts& path, Vec2 pos, Vec2 dir, int* flags) {
  const int n = (int)path.size();
  std::vector<char> front(n, 0);
  for (int i = 0; i < n; i++) front[i] = blas_dot2(path[i].x - pos.x, dir.x, path[i].y - pos.y, dir.y) > 0;
  for (int i = 0; i < n; i++)
    if (front[i]) {
      for (int j = i; j < n; j++) front[j] = 1;
      break;
    }
  for (int i = std::max(0, n - 20); i < n; i++) front[i] = 1;
  Pts infront;
  for (int i = 0; i < n; i++)
    if (front[i]) infront.push_back(path[i]);
  if (infront.empty()) return path;
  if (infront.size() < 2) throw RefUndefined{FSDO_REF_UNDEFINED_PATH};  // cumsum([])[-1] IndexError
  double plen = 0.0;
  for (size_t i = 0; i + 1 < infront.size(); i++)
    plen += norm2_axis(infront[i + 1].x - infront[i].x, infront[i + 1].y - infront[i].y);
  if (plen > MPC_PATH_LENGTH) return path;
  Pts rel(infront.end() - std::min<size_t>(20, infront.size()), infront.end());
  double cx, cy, radius;
  circle_fit(rel, cx, cy, radius);
  double r_use = py_min(py_max(radius, 10), 100);
  Pts newp;
  if (r_use < 80) {
    *flags |= 16;  // arc extension (sin/cos/atan2 feed the spline input)
    int nr = (int)rel.size();
    int i0 = 0, i1 = (int)(nr / 2), i2 = nr - 1;
    Vec2 t0{rel[i0].x - cx, rel[i0].y - cy}, t1{rel[i1].x - cx, rel[i1].y - cy}, t2{rel[i2].x - cx, rel[i2].y - cy};
    double hm[3][3] = {{1.0, t0.x, t0.y}, {1.0, t1.x, t1.y}, {1.0, t2.x, t2.y}};
    double sg = np_sign(det3_lu(hm));
    double start = m_atan2(t0.y, t0.x);
    double end = start + sg * PI;
    // np.linspace(start, end) -> 50 points
    const int NP = 50;
    double step = (end - start) / (double)(NP - 1);
    Pts raw(NP);
    for (int i = 0; i < NP; i++) {
      double a = (double)i * step + start;
      if (i == NP - 1) a = end;
      raw[i] = Vec2{m_cos(a) * r_use, m_sin(a) * r_use};
    }
    Vec2 last = path[n - 1];
    for (int i = 0; i < NP; i++) newp.push_back(Vec2{raw[i].x - raw[0].x + last.x, raw[i].y - raw[0].y + last.y});
  } else {
    *flags |= 32;  // straight extension
    Vec2 sl = path[n - 2], last = path[n - 1];
    double dx = last.x - sl.x, dy = last.y - sl.y;
    double nrm = norm2(dx, dy);
    dx /= nrm;
    dy /= nrm;
    for (int i = 0; i < 30; i++) newp.push_back(Vec2{last.x + dx * (double)i, last.y + dy * (double)i});
  }
  Pts out = path;
  out.insert(out.end(), newp.begin() + 1, newp.end());
  return out;
}

// core_calculate_path.py:459-465 remove_path_behind_car
static Pts remove_path_behind_car(const Pts& path, Vec2 pos) {
  int best = 0;
  double bd = 0;
  for (size_t i = 0; i < path.size(); i++) {
    double d = norm2_axis(pos.x - path[i].x, pos.y - path[i].y);
    if (i == 0 || d < bd) {
      bd = d;
      best = (int)i;
    }
  }
  return Pts(path.begin() + best, path.end());
}

struct FourColPath {};  // marker: remove_path_not_in_prediction_horizon returned the (40,4) previous path

// core_calculate_path.py:380-417 do_all_mpc_parameter_calculations
void do_all_mpc(const Pts& path_update, Vec2 pos, Vec2 dir, double out[][4], int* flags) {
  if (path_update.empty()) throw RefUndefined{FSDO_REF_UNDEFINED_PATH};
  Pts p1 = connect_path_to_car(path_update, pos, dir);
  Pts p2 = extend_path(p1, pos, dir, flags);
  Pts p3 = remove_path_behind_car(p2, pos);
  // refit_path_for_mpc_with_safety_factor :239-259
  Fitted fit = spline_fit(p3, SMOOTHING, PREDICT_EVERY, g_prm.max_deg);
  Pts p4 = spline_predict(fit, MPC_PATH_LENGTH * 1.5);
  // remove_path_not_in_prediction_horizon :467-499
  int nseg = (int)p4.size() - 1;
  if (nseg <= 1) {
    // returns previous_paths[-1], a (40,4) array, into a (n,2) context: parameterize_path then
    // fits 4-D points and np.linalg.det of a 3x5 matrix raises LinAlgError (a ValueError subclass)
    throw PyValueError{3};
  }
  std::vector<char> over(nseg);
  double cum = 0.0;
  for (int i = 0; i < nseg; i++) {
    cum += norm2_axis(p4[i + 1].x - p4[i].x, p4[i + 1].y - p4[i].y);
    over[i] = cum > MPC_PATH_LENGTH;
  }
  int first = 0;
  bool found = false;
  for (int i = 0; i < nseg; i++)
    if (over[i]) {
      first = i;
      found = true;
      break;
    }
  if (!found) first = nseg;
  Pts p5(p4.begin(), p4.begin() + first);
  parameterize_path(p5, out);
}

// calculate_path/path_calculator_helpers.py:26-68 + core_calculate_path.py:103-121: constant
// initial previous path = parameterize_path(fit(almost_straight_path).predict())
static double g_default[FSDO_PATH_POINTS][4];
static std::once_flag g_default_once;

static void build_default() {
  const int NP = 40;
  double max_angle = PI / 50;
  double step = (std::fabs(max_angle) - 0.0) / (double)(NP - 1);
  Rot rot(-(PI / 2));
  Pts chord(NP);
  for (int i = 0; i < NP; i++) {
    double a = (double)i * step + 0.0;
    if (i == NP - 1) a = std::fabs(max_angle);
    double px = (std::cos(a) - 1.0) * 1000.0, py = (std::sin(a) - 0.0) * 1000.0;
    Vec2 q = rot.apply(px, py);
    q.y *= np_sign(max_angle);
    chord[i] = q;
  }
  Fitted fit = spline_fit(chord, SMOOTHING, PREDICT_EVERY, g_prm.max_deg);
  Pts initial = spline_predict(fit, fit.max_u);
  parameterize_path(initial, g_default);
}

const double (*default_previous_path())[4] {
  std::call_once(g_default_once, build_default);
  return g_default;
}
// after fsdo_set_params: the constant initial previous path depends on smoothing / predict_every
void rebuild_default_previous_path() {
  std::call_once(g_default_once, build_default);
  build_default();
}

// core_calculate_path.py:514-575 run_path_calculation (global_path is None)
// global_path (core_calculate_path.py:514-529, set through PathPlanner.set_global_path / the relocalizers): when given, the
// basis of the path is the part of it within 30 m of the car, rolled so that it starts a third of the table before the
// closest point; sorting / matching results are ignored.
void calculate_path(const Pts& left_v, const Pts& right_v, const std::vector<int>& l2r, const std::vector<int>& r2l,
                    Vec2 pos, Vec2 dir, PathOut& out, const double (*prev_in)[4], const Pts* global_path) {
  const double(*prev)[4] = prev_in ? prev_in : default_previous_path();
  Pts prev_xy(HORIZON);
  for (int i = 0; i < HORIZON; i++) prev_xy[i] = Vec2{prev[i][1], prev[i][2]};
  out.fallback = 0;
  Pts center;
  if (global_path) {
    const Pts& gp = *global_path;
    const long n = (long)gp.size();
    if (n == 0) throw RefUndefined{FSDO_REF_UNDEFINED_PATH};  // argmin of an empty array
    std::vector<double> dist(n);
    long imin = 0;
    for (long i = 0; i < n; i++) {
      dist[i] = norm2_axis(pos.x - gp[i].x, pos.y - gp[i].y);  // np.linalg.norm(position - path, axis=1)
      if (dist[i] < dist[imin]) imin = i;                      // first smallest
    }
    const long roll = -imin + n / 3;
    for (long k = 0; k < n; k++) {
      long src = ((k - roll) % n + n) % n;  // np.roll: out[k] = in[(k - roll) mod n]
      if (dist[src] < 30) center.push_back(gp[src]);
    }
  } else if (left_v.size() < 3 && right_v.size() < 3) {
    center = prev_xy;
    out.fallback |= 1;
  } else {
    // select_side_to_use :151-183: max over [LEFT, RIGHT] of (n_matches, sum of indices); LEFT wins ties
    auto score = [](const std::vector<int>& m, long& n, long& s) {
      n = 0;
      s = 0;
      for (int v : m)
        if (v != -1) {
          n++;
          s += v;
        }
    };
    long nl, sl, nr, sr;
    score(l2r, nl, sl);
    score(r2l, nr, sr);
    bool use_left = !((nr > nl) || (nr == nl && sr > sl));
    const Pts& side = use_left ? left_v : right_v;
    const std::vector<int>& m = use_left ? l2r : r2l;
    const Pts& other = use_left ? right_v : left_v;
    // other_side_cones[matches] — NumPy fancy indexing, -1 wraps; empty other with any index raises
    if (!m.empty() && other.empty()) throw RefUndefined{FSDO_REF_UNDEFINED_MATCH_IDX};
    for (size_t i = 0; i < side.size(); i++) {
      int mi = m[i];
      int j = (mi < 0) ? (int)other.size() + mi : mi;
      if (j < 0 || j >= (int)other.size()) throw RefUndefined{FSDO_REF_UNDEFINED_MATCH_IDX};
      if (mi != -1) center.push_back(Vec2{(side[i].x + other[j].x) / 2, (side[i].y + other[j].y) / 2});
    }
    if (center.size() < 2) {
      center = prev_xy;
      out.fallback |= 1;
    }
  }
  // fit_matches_as_spline :207-223
  Pts path_update;
  try {
    Fitted f = spline_fit(center, SMOOTHING, PREDICT_EVERY, g_prm.max_deg);
    path_update = spline_predict(f, f.max_u);
  } catch (PyValueError&) {
    out.fallback |= 2;
    Fitted f = spline_fit(prev_xy, SMOOTHING, PREDICT_EVERY, g_prm.max_deg);
    path_update = spline_predict(f, f.max_u);
  }
  finish_path(path_update, prev_xy, pos, dir, out);
}

// core_calculate_path.py:555-575: overwrite-if-too-far, MPC step with its ValueError retry
void finish_path(Pts path_update, const Pts& prev_xy, Vec2 pos, Vec2 dir, PathOut& out) {
  // overwrite_path_if_it_is_too_far_away :225-237
  if (path_update.empty()) throw RefUndefined{FSDO_REF_UNDEFINED_PATH};  // min() of empty
  {
    double md = 0;
    for (size_t i = 0; i < path_update.size(); i++) {
      double d = norm2_axis(pos.x - path_update[i].x, pos.y - path_update[i].y);
      if (i == 0 || d < md) md = d;
    }
    if (md > MAX_DIST_VALID_PATH) {
      path_update = prev_xy;
      out.fallback |= 4;
    }
  }
  try {
    do_all_mpc(path_update, pos, dir, out.p, &out.fallback);
  } catch (PyValueError&) {
    out.fallback |= 8;
    try {
      do_all_mpc(prev_xy, pos, dir, out.p, &out.fallback);
    } catch (PyValueError&) {
      throw RefUndefined{FSDO_REF_UNDEFINED_PATH};
    }
  }
}

// path_calculator_helpers.py:56-68 calculate_almost_straight_path (40 chord points)
Pts almost_straight_path() {
  const int NP = 40;
  double max_angle = PI / 50;
  double step = (std::fabs(max_angle) - 0.0) / (double)(NP - 1);
  Rot rot(-(PI / 2));
  Pts chord(NP);
  for (int i = 0; i < NP; i++) {
    double a = (double)i * step + 0.0;
    if (i == NP - 1) a = std::fabs(max_angle);
    double px = (std::cos(a) - 1.0) * 1000.0, py = (std::sin(a) - 0.0) * 1000.0;
    Vec2 q = rot.apply(px, py);
    q.y *= np_sign(max_angle);
    chord[i] = q;
  }
  return chord;
}

}  // namespace fsdo
