// TEST INFRASTRUCTURE (oracle) — C entry points.  Orchestration restates
// full_pipeline/full_pipeline.py:84-207 (mission without relocalizer: sorting -> matching -> path).
#include <cstring>
#include <thread>
#include <atomic>
#include <cmath>
#include <malloc.h>

#include "oracle_internal.h"

using namespace fsdo;
namespace fsdo {
extern int g_math_mode;
}

static Frame make_frame(const double* xyt, int n, const double* pose) {
  Frame f;
  f.n = n;
  f.x.resize(n);
  f.x.clear();
  f.y.clear();
  f.type.clear();
  f.orig.clear();
  for (int i = 0; i < n; i++) {
    const int t = (int)xyt[3 * i + 2];
    // core_cone_sorting.py:113-115: use_unknown_cones = False empties the UNKNOWN list before the cones are flattened
    if (!g_prm.use_unknown_cones && t == 0) continue;
    f.x.push_back(xyt[3 * i]);
    f.y.push_back(xyt[3 * i + 1]);
    f.type.push_back(t);
    f.orig.push_back(i);
  }
  f.n = (int)f.x.size();
  f.px = pose[0];
  f.py = pose[1];
  f.dx = pose[2];
  f.dy = pose[3];
  return f;
}

static void clear_result(fsdo_frame_result* o) {
  std::memset(o, 0, sizeof(*o));
  for (int i = 0; i < FSDO_MAX_LEN; i++) o->left_idx[i] = o->right_idx[i] = -1;
  for (int i = 0; i < FSDO_MAX_MATCH; i++) o->l2r[i] = o->r2l[i] = -1;
  o->first_k_left[0] = o->first_k_left[1] = o->first_k_right[0] = o->first_k_right[1] = -1;
  o->best_cost_left = o->best_cost_right = NAN;
  for (int i = 0; i < FSDO_PATH_POINTS; i++)
    for (int j = 0; j < 4; j++) o->path[i][j] = NAN;
}

static void fill_sort(const Frame& f, const std::vector<int>& l, const std::vector<int>& r, const SideResult& L, const SideResult& R,
                      fsdo_frame_result* o) {
  o->n_left = (int)l.size();
  o->n_right = (int)r.size();
  // indices are reported in the caller's array (f.orig is the identity unless UNKNOWN cones were dropped)
  for (size_t i = 0; i < l.size(); i++) o->left_idx[i] = f.orig[l[i]];
  for (size_t i = 0; i < r.size(); i++) o->right_idx[i] = f.orig[r[i]];
  o->n_configs_left = L.has ? (int)L.configs.size() : 0;
  o->n_configs_right = R.has ? (int)R.configs.size() : 0;
  for (int i = 0; i < 2; i++) {
    o->first_k_left[i] = L.first_k[i] >= 0 ? f.orig[L.first_k[i]] : -1;
    o->first_k_right[i] = R.first_k[i] >= 0 ? f.orig[R.first_k[i]] : -1;
  }
  if (L.has) o->best_cost_left = L.costs[0];
  if (R.has) o->best_cost_right = R.costs[0];
}

static void fill_match(const Pts& lv, const Pts& rv, const std::vector<int>& l2r, const std::vector<int>& r2l,
                       fsdo_frame_result* o) {
  if (lv.size() > FSDO_MAX_MATCH || rv.size() > FSDO_MAX_MATCH) throw RefUndefined{FSDO_REF_UNDEFINED_OTHER};
  o->n_left_v = (int)lv.size();
  o->n_right_v = (int)rv.size();
  for (size_t i = 0; i < lv.size(); i++) {
    o->left_v[i][0] = lv[i].x;
    o->left_v[i][1] = lv[i].y;
    o->l2r[i] = l2r[i];
  }
  for (size_t i = 0; i < rv.size(); i++) {
    o->right_v[i][0] = rv[i].x;
    o->right_v[i][1] = rv[i].y;
    o->r2l[i] = r2l[i];
  }
}

extern "C" {

// np_compat.h det3_lu for the tests: xy6 = x0,y0,x1,y1,x2,y2 -> determinant of [[1,x0,y0],[1,x1,y1],[1,x2,y2]] (sign-exact)
double fsdo_det3(const double* xy6) {
  const double h[3][3] = {{1.0, xy6[0], xy6[1]}, {1.0, xy6[2], xy6[3]}, {1.0, xy6[4], xy6[5]}};
  return det3_lu(h);
}

// decision margins of the sorting stage (oracle_internal.h MarginRec): enable / reset, read (9 classes x 6 values:
// min non-zero margin, decisions, below 1e-6, below 1e-9, below 1e-12, exactly zero)
void fsdo_margins_enable(int on) {
  g_margins = MarginRec();
  g_margins.on = on != 0;
  for (int i = 0; i < MG_CLASSES; i++) {
    g_margins.min_margin[i] = 1e300;
    g_margins.n[i] = g_margins.below_1e6[i] = g_margins.below_1e9[i] = g_margins.below_1e12[i] = g_margins.zero[i] = 0;
  }
}
void fsdo_margins_get(double* out54) {
  for (int i = 0; i < MG_CLASSES; i++) {
    out54[6 * i + 0] = g_margins.min_margin[i];
    out54[6 * i + 1] = (double)g_margins.n[i];
    out54[6 * i + 2] = (double)g_margins.below_1e6[i];
    out54[6 * i + 3] = (double)g_margins.below_1e9[i];
    out54[6 * i + 4] = (double)g_margins.below_1e12[i];
    out54[6 * i + 5] = (double)g_margins.zero[i];
  }
}

void fsdo_sort_frame(const double* xyt, int n, const double* pose, fsdo_frame_result* o) {
  clear_result(o);
  try {
    Frame f = make_frame(xyt, n, pose);
    std::vector<int> l, r;
    SideResult L, R;
    sort_frame(f, l, r, &L, &R);
    fill_sort(f, l, r, L, R, o);
  } catch (RefUndefined& e) {
    o->status = e.code;
  }
}

void fsdo_match(const double* left, int nl, const double* right, int nr, const double* pose, fsdo_frame_result* o) {
  clear_result(o);
  try {
    Pts L(nl), R(nr), lv, rv;
    for (int i = 0; i < nl; i++) L[i] = Vec2{left[2 * i], left[2 * i + 1]};
    for (int i = 0; i < nr; i++) R[i] = Vec2{right[2 * i], right[2 * i + 1]};
    std::vector<int> l2r, r2l;
    match_cones(L, R, Vec2{pose[0], pose[1]}, lv, rv, l2r, r2l);
    fill_match(lv, rv, l2r, r2l, o);
  } catch (RefUndefined& e) {
    o->status = e.code;
  }
}

void fsdo_path(const double* left_v, int nl, const double* right_v, int nr, const int32_t* l2r, const int32_t* r2l,
               const double* pose, fsdo_frame_result* o) {
  clear_result(o);
  try {
    Pts L(nl), R(nr);
    for (int i = 0; i < nl; i++) L[i] = Vec2{left_v[2 * i], left_v[2 * i + 1]};
    for (int i = 0; i < nr; i++) R[i] = Vec2{right_v[2 * i], right_v[2 * i + 1]};
    std::vector<int> a(l2r, l2r + nl), b(r2l, r2l + nr);
    PathOut po;
    calculate_path(L, R, a, b, Vec2{pose[0], pose[1]}, Vec2{pose[2], pose[3]}, po);
    std::memcpy(o->path, po.p, sizeof(po.p));
    o->path_fallback = po.fallback;
  } catch (RefUndefined& e) {
    o->status = e.code;
  } catch (PyValueError&) {
    o->status = FSDO_REF_UNDEFINED_PATH;
  }
}

void fsdo_plan_frame_global(const double* xyt, int n, const double* pose, const double* prev40x4, const double* gpath_xy,
                            int n_gpath, fsdo_frame_result* o);

void fsdo_plan_frame(const double* xyt, int n, const double* pose, fsdo_frame_result* o) {
  fsdo_plan_frame_global(xyt, n, pose, nullptr, nullptr, 0, o);
}

// sequential-replay form: prev40x4 = the previous output of this planner (CalculatePath.previous_paths[-1]) or NULL
void fsdo_plan_frame_prev(const double* xyt, int n, const double* pose, const double* prev40x4, fsdo_frame_result* o) {
  fsdo_plan_frame_global(xyt, n, pose, prev40x4, nullptr, 0, o);
}

// the same with PathPlanner.global_path set (full_pipeline.py:81-82,181-183): gpath_xy (n_gpath,2) or NULL
void fsdo_plan_frame_global(const double* xyt, int n, const double* pose, const double* prev40x4, const double* gpath_xy,
                            int n_gpath, fsdo_frame_result* o) {
  clear_result(o);
  Pts gp;
  if (gpath_xy)
    for (int i = 0; i < n_gpath; i++) gp.push_back(Vec2{gpath_xy[2 * i], gpath_xy[2 * i + 1]});
  try {
    Frame f = make_frame(xyt, n, pose);
    std::vector<int> l, r;
    SideResult L, R;
    sort_frame(f, l, r, &L, &R);
    fill_sort(f, l, r, L, R, o);
    Pts sl(l.size()), sr(r.size()), lv, rv;
    for (size_t i = 0; i < l.size(); i++) sl[i] = Vec2{f.x[l[i]], f.y[l[i]]};
    for (size_t i = 0; i < r.size(); i++) sr[i] = Vec2{f.x[r[i]], f.y[r[i]]};
    std::vector<int> l2r, r2l;
    match_cones(sl, sr, Vec2{f.px, f.py}, lv, rv, l2r, r2l);
    fill_match(lv, rv, l2r, r2l, o);
    PathOut po;
    calculate_path(lv, rv, l2r, r2l, Vec2{f.px, f.py}, Vec2{f.dx, f.dy}, po, (const double(*)[4])prev40x4, gpath_xy ? &gp : nullptr);
    std::memcpy(o->path, po.p, sizeof(po.p));
    o->path_fallback = po.fallback;
  } catch (RefUndefined& e) {
    o->status = e.code;
  } catch (PyValueError&) {
    o->status = FSDO_REF_UNDEFINED_PATH;
  }
}

void fsdo_plan_batch(int n_frames, const int32_t* off, const double* xyt, const double* poses, fsdo_frame_result* out,
                     int n_threads) {
  default_previous_path();
  static bool tuned = false;
  if (!tuned) {  // keep per-frame temporaries off mmap/munmap (many host threads)
    mallopt(M_MMAP_THRESHOLD, 64 << 20);
    mallopt(M_TRIM_THRESHOLD, 256 << 20);
    tuned = true;
  }
  if (n_threads <= 1) {
    for (int i = 0; i < n_frames; i++) fsdo_plan_frame(xyt + 3 * (size_t)off[i], off[i + 1] - off[i], poses + 4 * (size_t)i, &out[i]);
    return;
  }
  std::atomic<int> next(0);
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; t++)
    th.emplace_back([&]() {
      for (;;) {
        int i = next.fetch_add(1);
        if (i >= n_frames) break;
        fsdo_plan_frame(xyt + 3 * (size_t)off[i], off[i + 1] - off[i], poses + 4 * (size_t)i, &out[i]);
      }
    });
  for (auto& t : th) t.join();
}

void fsdo_default_path(double* out) { std::memcpy(out, default_previous_path(), sizeof(double) * FSDO_PATH_POINTS * 4); }

int fsdo_side_configs(const double* xyt, int n, const double* pose, int cone_type, int32_t* configs_out, double* costs_out,
                      int max_configs, int32_t* first_k_out) {
  try {
    Frame f = make_frame(xyt, n, pose);
    SideResult s = configs_for_one_side(f, cone_type);
    first_k_out[0] = s.first_k[0];
    first_k_out[1] = s.first_k[1];
    if (!s.has) return 0;
    int C = (int)s.configs.size();
    for (int i = 0; i < C && i < max_configs; i++) {
      for (int l = 0; l < FSDO_MAX_LEN; l++) configs_out[i * FSDO_MAX_LEN + l] = s.configs[i].v[l];
      costs_out[i] = s.costs[i];
    }
    return C;
  } catch (RefUndefined& e) {
    return -e.code;
  }
}

int fsdo_result_size(void) { return (int)sizeof(fsdo_frame_result); }

// plan one frame and hand back the smoothing splines it fitted, in call order (the reference's splprep calls inside
// calculate_path_in_global_frame): up to max_fits records of FSDO_FIT_STRIDE doubles [k, n, t[0..n), cx[0..n), cy[0..n)]
// (n <= FSDO_FIT_KNOTS, longer knot vectors are truncated and flagged by n > FSDO_FIT_KNOTS).  Returns the number of fits.
int fsdo_plan_frame_capture(const double* cones_xyt, int n, const double* pose, fsdo_frame_result* out, double* fits, int max_fits) {
  std::vector<fsdo::Spline> cap;
  fsdo::g_fit_capture = &cap;
  fsdo_plan_frame(cones_xyt, n, pose, out);
  fsdo::g_fit_capture = nullptr;
  for (int i = 0; i < (int)cap.size() && i < max_fits; i++) {
    double* r = fits + (size_t)i * FSDO_FIT_STRIDE;
    const fsdo::Spline& s = cap[i];
    r[0] = s.k;
    r[1] = s.n;
    const int nn = s.n < FSDO_FIT_KNOTS ? s.n : FSDO_FIT_KNOTS;
    for (int j = 0; j < nn; j++) {
      r[2 + j] = s.t[j];
      r[2 + FSDO_FIT_KNOTS + j] = s.cx[j];
      r[2 + 2 * FSDO_FIT_KNOTS + j] = s.cy[j];
    }
  }
  return (int)cap.size();
}
void fsdo_set_math_mode(int mode) { fsdo::g_math_mode = mode; }
// 17 values in the order of OParams (ints as doubles); NULL restores the reference's defaults.  Not thread-safe: call
// between batches.
void fsdo_set_params(const double* v) {
  fsdo::OParams p;
  if (v) {
    p.max_n_neighbors = (int)v[0];
    p.max_length = (int)v[1];
    p.max_dist = v[2];
    p.max_dist_to_first = v[3];
    p.threshold_directional_angle = v[4];
    p.threshold_absolute_angle = v[5];
    p.min_track_width = v[6];
    p.max_search_range = v[7];
    p.max_search_angle = v[8];
    p.smoothing = v[9];
    p.predict_every = v[10];
    p.maximal_distance_for_valid_path = v[11];
    p.mpc_path_length = v[12];
    p.max_deg = (int)v[13];
    p.horizon = (int)v[14];
    p.matches_should_be_monotonic = (int)v[15];
    p.use_unknown_cones = (int)v[16];
  }
  fsdo::g_prm = p;
  fsdo::rebuild_default_previous_path();
}
int fsdo_get_math_mode(void) { return fsdo::g_math_mode; }
}
