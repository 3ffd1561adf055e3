// TEST INFRASTRUCTURE (oracle) — sorting_cones step, scalar restatement with NumPy semantics.
// Each function names the reference function (file:line under /root/reference/fsd_path_planning)
// whose behaviour it restates.
#include <cmath>
#include <vector>
#include <algorithm>

#include "fsd_oracle.h"
#include "np_compat.h"
#include "oracle_internal.h"

namespace fsdo {

MarginRec g_margins;


static const int T_UNKNOWN = 0, T_RIGHT = 1, T_LEFT = 2;

static inline int invert_cone_type(int t) {  // utils/cone_types.py:22-34
  if (t == T_LEFT) return T_RIGHT;
  if (t == T_RIGHT) return T_LEFT;
  return t;
}

// utils/math_utils.py:493-530 points_inside_ellipse for one point
static inline bool point_inside_ellipse(double px, double py, double cx, double cy, double dirx, double diry,
                                        double major, double minor) {
  double ang = std::atan2(diry, dirx);
  Rot r(-ang);
  Vec2 q = r.apply(px - cx, py - cy);
  double crit = (q.x * q.x) / (major * major) + (q.y * q.y) / (minor * minor);
  return crit < 1;
}

// ---- S4: start cones -------------------------------------------------------------------
// trace_sorter/core_trace_sorter.py:379-407 mask_cone_can_be_first_in_config
static void mask_first(const Frame& f, int cone_type, std::vector<double>& dist, std::vector<char>& valid) {
  const int n = f.n;
  dist.assign(n, 0.0);
  valid.assign(n, 0);
  double car_ang = std::atan2(f.dy, f.dx);
  Rot r(-car_ang);
  const double max_dist_to_first = g_prm.max_dist_to_first;
  for (int i = 0; i < n; i++) {
    Vec2 rel = r.apply(f.x[i] - f.px, f.y[i] - f.py);
    double ang = std::atan2(rel.y, rel.x);
    dist[i] = std::sqrt(rel.x * rel.x + rel.y * rel.y);
    bool in_ell = point_inside_ellipse(f.x[i], f.y[i], f.px, f.py, f.dx, f.dy, max_dist_to_first * 1.5,
                                       max_dist_to_first / 1.5);
    double sgn = np_sign(ang);
    double want = (cone_type == T_LEFT) ? 1.0 : -1.0;
    bool side = (sgn == want);
    bool a_max = std::fabs(ang) < PI - PI / 5;
    bool a_min = std::fabs(ang) > PI / 10;
    if (in_ell && f.type[i] == T_UNKNOWN) {  // (only there does the bearing decide anything)
      margin(MG_START_BEARING, ang, 0.0);
      margin(MG_START_BEARING, std::fabs(ang), PI - PI / 5);
      margin(MG_START_BEARING, std::fabs(ang), PI / 10);
    }
    bool right_color = (f.type[i] == cone_type);
    bool mask_side = (side && a_max && a_min) || right_color;
    bool not_opp = (f.type[i] != invert_cone_type(cone_type));
    valid[i] = in_ell && mask_side && not_opp;
  }
}

// trace_sorter/core_trace_sorter.py:344-377 select_starting_cone
static int select_starting_cone(const Frame& f, int cone_type, const std::vector<char>* skip) {
  std::vector<double> dist;
  std::vector<char> valid;
  mask_first(f, cone_type, dist, valid);
  if (skip)
    for (int i = 0; i < f.n; i++)
      if ((*skip)[i]) valid[i] = 0;
  int best = -1;
  double bd = INFINITY;
  for (int i = 0; i < f.n; i++) {
    if (!valid[i]) continue;
    if (best < 0 || dist[i] < bd) {  // argsort -> first smallest
      bd = dist[i];
      best = i;
    }
  }
  if (best < 0) return -1;
  if (dist[best] > g_prm.max_dist_to_first) return -1;
  return best;
}

// trace_sorter/core_trace_sorter.py:409-465 select_first_k_starting_cones
static std::vector<int> select_first_k(const Frame& f, int cone_type) {
  int index_1 = select_starting_cone(f, cone_type, nullptr);
  if (index_1 < 0) return {};
  std::vector<char> skip(f.n, 0);
  for (int i = 0; i < f.n; i++) {
    double a = vec_angle_between(f.x[i] - f.px, f.y[i] - f.py, f.dx, f.dy);
    if (std::fabs(a) < PI / 2) skip[i] = 1;
  }
  skip[index_1] = 1;
  int index_2 = select_starting_cone(f, cone_type, &skip);
  if (index_2 < 0) return {index_1};
  double d1x = f.x[index_1] - f.x[index_2], d1y = f.y[index_1] - f.y[index_2];
  double d2x = f.x[index_2] - f.x[index_1], d2y = f.y[index_2] - f.y[index_1];
  double angle_1 = vec_angle_between(d1x, d1y, f.dx, f.dy);
  double angle_2 = vec_angle_between(d2x, d2y, f.dx, f.dy);
  if (angle_1 > angle_2) std::swap(index_1, index_2);
  double dist = norm2(d1x, d1y);
  const double max_dist = g_prm.max_dist;
  if (dist > max_dist * 1.1 || dist < 1.4) return {index_1};
  return {index_2, index_1};
}

// ---- S5: adjacency -----------------------------------------------------------------------
// trace_sorter/adjacency_matrix.py:60-128 create_adjacency_matrix (+ common.py:36-67 BFS)
static void create_adjacency(const Frame& f, int n_neighbors, int start_idx, int cone_type,
                             std::vector<std::vector<int>>& nbrs, int& n_reachable) {
  const int n = f.n;
  const double max_dist = g_prm.max_dist;
  const int other = invert_cone_type(cone_type);
  // thread-local scratch: an N x N double matrix is 128 KB at N = 128 — above glibc's mmap threshold, so a fresh
  // std::vector per call would mmap/munmap on every frame and serialise many host threads in the kernel
  static thread_local std::vector<double> D;
  if (D.size() < (size_t)n * n) D.resize((size_t)n * n);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) {
      double d = cdist_sq(f.x[i], f.y[i], f.x[j], f.y[j]);
      if (i == j) d = INFINITY;
      if (f.type[i] == other || f.type[j] == other) d = INFINITY;
      D[(size_t)i * n + j] = d;
    }
  static thread_local std::vector<char> adj;
  adj.assign((size_t)n * n, 0);
  std::vector<int> order(n);
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < n; j++) order[j] = j;
    const double* row = &D[(size_t)i * n];
    // (np.argsort: NaN distances — a cone with a coordinate that is not finite — go last, behind the infinities)
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return row[a] < row[b] || (row[b] != row[b] && row[a] == row[a]); });
    for (int q = 0; q < n_neighbors && q < n; q++) {
      int j = order[q];
      if (!(row[j] > max_dist * max_dist)) adj[(size_t)i * n + j] = 1;
    }
  }
  nbrs.assign(n, {});
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++)
      if (adj[(size_t)i * n + j] && adj[(size_t)j * n + i]) nbrs[i].push_back(j);  // ascending j
  // BFS order length
  std::vector<char> vis(n, 0);
  std::vector<int> queue;
  queue.push_back(start_idx);
  vis[start_idx] = 1;
  for (size_t qp = 0; qp < queue.size(); qp++) {
    int node = queue[qp];
    for (int j : nbrs[node])
      if (!vis[j]) {
        vis[j] = 1;
        queue.push_back(j);
      }
  }
  n_reachable = (int)queue.size();
}

// ---- line segment intersection: trace_sorter/line_segment_intersection.py:18-200 ---------
static bool parallel_case(Vec2 a0, Vec2 a1, Vec2 b0, Vec2 b1, double eps) {
  double dx = a1.x - a0.x, dy = a1.y - a0.y;
  bool maybe;
  double slope;
  if (dx < eps) {
    maybe = std::fabs(a0.x - b0.x) < eps;
    slope = INFINITY;
  } else {
    slope = dy / dx;
    double ia = a0.y - slope * a0.x;
    double ib = b0.y - slope * b0.x;
    maybe = std::fabs(ia - ib) < eps;
  }
  if (!maybe) return false;
  bool use_y = slope > 1;
  auto ax = [&](Vec2 p) { return use_y ? p.y : p.x; };
  double left_end, right_start;
  if (ax(a0) < ax(b0)) {
    left_end = ax(a1);
    right_start = std::min(ax(b0), ax(b1));
  } else {
    left_end = ax(b1);
    right_start = std::min(ax(a0), ax(a1));
  }
  return left_end >= right_start;
}

static inline void cross3(const double a[3], const double b[3], double o[3]) {
  // numpy.cross for 3-vectors: cp0 = a1*b2 - a2*b1, cp1 = a2*b0 - a0*b2, cp2 = a0*b1 - a1*b0
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

bool segments_intersect(Vec2 a0, Vec2 a1, Vec2 b0, Vec2 b1) {
  const double eps = 1e-6;
  double h0[3] = {a0.x, a0.y, 1.0}, h1[3] = {a1.x, a1.y, 1.0}, h2[3] = {b0.x, b0.y, 1.0}, h3[3] = {b1.x, b1.y, 1.0};
  double la[3], lb[3], it[3];
  cross3(h0, h1, la);
  cross3(h2, h3, lb);
  cross3(la, lb, it);
  if (std::fabs(it[2]) < eps) return parallel_case(a0, a1, b0, b1, eps);
  double ix = it[0] / it[2], iy = it[1] / it[2];
  double al = std::min(a0.x, a1.x), ar = std::max(a0.x, a1.x);
  double bl = std::min(b0.x, b1.x), br = std::max(b0.x, b1.x);
  double ab = std::min(a0.y, a1.y), at = std::max(a0.y, a1.y);
  double bb = std::min(b0.y, b1.y), bt = std::max(b0.y, b1.y);
  return (al - eps <= ix && ix <= ar + eps) && (bl - eps <= ix && ix <= br + eps) && (ab - eps <= iy && iy <= at + eps) &&
         (bb - eps <= iy && iy <= bt + eps);
}

// ---- S9: trace_sorter/end_configurations.py:108-223 ----------------------------------------
static void neighbor_mask(const Frame& f, int cone_type, const std::vector<int>& attempt, int pos,
                          const std::vector<int>& neighbors, std::vector<char>& can) {
  const double thr_dir = g_prm.threshold_directional_angle, thr_abs = g_prm.threshold_absolute_angle, car_size = 2.1;
  const int m = (int)neighbors.size();
  can.assign(m, 0);
  double nrm = norm2(f.dx, f.dy);
  double dnx = f.dx / nrm, dny = f.dy / nrm;
  for (int i = 0; i < m; i++) {
    bool in_attempt = false;
    for (int q = 0; q <= pos; q++)
      if (attempt[q] == neighbors[i]) in_attempt = true;
    can[i] = !in_attempt;
  }
  if (pos >= 1) {
    // calculate_mask_within_ellipse :281-300
    int last = attempt[pos], sl = attempt[pos - 1];
    double mdx = f.x[last] - f.x[sl], mdy = f.y[last] - f.y[sl];
    for (int i = 0; i < m; i++) {
      int c = neighbors[i];
      bool in = point_inside_ellipse(f.x[c], f.y[c], f.x[last], f.y[last], mdx, mdy, 6, 3);
      can[i] = can[i] && in;
    }
  }
  if (pos == 0) {
    // mask_second_in_attempt_is_on_right_vehicle_side :260-278
    double a_car = std::atan2(dny, dnx);
    for (int i = 0; i < m; i++) {
      int c = neighbors[i];
      double a_n = std::atan2(f.y[c] - f.py, f.x[c] - f.px);
      double diff = angle_difference(a_n, a_car);
      double want = (cone_type == T_LEFT) ? 1.0 : -1.0;
      bool ok = (np_sign(diff) == want) || (std::fabs(diff) < deg2rad(5));
      if (can[i]) {
        margin(MG_SECOND_SIDE, diff, 0.0);
        margin(MG_SECOND_SIDE, std::fabs(diff), deg2rad(5));
      }
      can[i] = can[i] && ok;
    }
  }
  for (int i = 0; i < m; i++) {
    if (!can[i]) continue;
    int cand = neighbors[i];
    int last = attempt[pos];
    // check_if_neighbor_lies_between_last_in_attempt_and_candidate :226-257
    for (int nb : neighbors) {
      if (nb == neighbors[i]) continue;
      double lx = f.x[last] - f.x[nb], ly = f.y[last] - f.y[nb];
      double cx = f.x[cand] - f.x[nb], cy = f.y[cand] - f.y[nb];
      double dc = norm2(cx, cy), dl = norm2(lx, ly);
      if (dc < 6.0 && dl < 6.0) margin(MG_ACOS_THRESHOLDS, std::cos(vec_angle_between(lx, ly, cx, cy)), std::cos(deg2rad(150)));
      if (dc < 6.0 && dl < 6.0 && vec_angle_between(lx, ly, cx, cy) > deg2rad(150)) {
        can[i] = 0;
        break;
      }
    }
    double cpx = f.x[cand], cpy = f.y[cand];
    if (can[i] && pos >= 1) {
      int sl = attempt[pos - 1];
      double s2lx = f.x[last] - f.x[sl], s2ly = f.y[last] - f.y[sl];
      double l2cx = cpx - f.x[last], l2cy = cpy - f.y[last];
      double angle_1 = std::atan2(s2ly, s2lx);
      double angle_2 = std::atan2(l2cy, l2cx);
      double difference = angle_difference(angle_2, angle_1);
      double len = norm2(l2cx, l2cy);
      margin(MG_ABS_ANGLE, std::fabs(difference), thr_abs);
      if (!(std::fabs(difference) > thr_abs) && !(len < 4.0)) margin(MG_DIR_ANGLE, difference, cone_type == T_LEFT ? thr_dir : -thr_dir);
      if (std::fabs(difference) > thr_abs)
        can[i] = 0;
      else if (cone_type == T_LEFT)
        can[i] = (difference < thr_dir) || (len < 4.0);
      else
        can[i] = (difference > -thr_dir) || (len < 4.0);
      if (pos >= 2) {
        int tl = attempt[pos - 2];
        double t2sx = f.x[sl] - f.x[tl], t2sy = f.y[sl] - f.y[tl];
        double angle_3 = std::atan2(t2sy, t2sx);
        double difference_2 = angle_difference(angle_1, angle_3);
        if (can[i]) {
          margin(MG_SIGN_FLIP, difference, 0.0);
          margin(MG_SIGN_FLIP, difference_2, 0.0);
          if (np_sign(difference) != np_sign(difference_2)) margin(MG_SIGN_FLIP, std::fabs(difference - difference_2), 1.3);
        }
        if (np_sign(difference) != np_sign(difference_2) && std::fabs(difference - difference_2) > 1.3) can[i] = 0;
      }
    }
    if (can[i] && pos == 1) {
      int st = attempt[0];
      double off = vec_angle_between(f.dx, f.dy, cpx - f.x[st], cpy - f.y[st]);
      margin(MG_ACOS_THRESHOLDS, std::cos(off), 0.0);
      can[i] = can[i] && (off < PI / 2);
    }
    if (can[i] && pos >= 0) {
      Vec2 car_start{f.px - dnx * car_size / 2, f.py - dny * car_size / 2};
      Vec2 car_end{f.px + dnx * car_size, f.py + dny * car_size};
      bool hit = segments_intersect(Vec2{f.x[last], f.y[last]}, Vec2{cpx, cpy}, car_start, car_end);
      can[i] = can[i] && !hit;
    }
  }
}

// ---- S8 + S10: trace_sorter/end_configurations.py:320-431, 434-520 ----------------------
// returns configs (rows of 12, -1 padded) lexicographically sorted & de-prefixed; empty => NoPathError
static std::vector<Config> find_all_end_configurations(const Frame& f, int cone_type, int start_idx,
                                                       const std::vector<std::vector<int>>& nbrs, int target_length,
                                                       const std::vector<int>& first_k) {
  std::vector<std::vector<int>> ends;
  std::vector<int> attempt(target_length, -1);
  struct Item {
    int node, pos;
  };
  std::vector<Item> stack;
  if (!first_k.empty()) {
    int pos = (int)first_k.size() - 1;
    for (int q = 0; q < pos; q++) {
      if (q >= target_length) throw RefUndefined{FSDO_REF_UNDEFINED_DFS_OOB};
      attempt[q] = first_k[q];
    }
    stack.push_back({first_k.back(), pos});
  } else {
    stack.push_back({start_idx, 0});
  }
  std::vector<char> can;
  while (!stack.empty()) {
    Item it = stack.back();
    stack.pop_back();
    if (it.pos >= target_length) throw RefUndefined{FSDO_REF_UNDEFINED_DFS_OOB};  // numpy IndexError
    attempt[it.pos] = it.node;
    for (int q = it.pos + 1; q < target_length; q++) attempt[q] = -1;
    const std::vector<int>& neighbors = nbrs[it.node];
    neighbor_mask(f, cone_type, attempt, it.pos, neighbors, can);
    bool any = false;
    for (char c : can) any = any || c;
    bool has_valid = (it.pos < target_length - 1) && any;
    if (has_valid) {
      for (size_t i = 0; i < can.size(); i++)
        if (can[i]) stack.push_back({neighbors[i], it.pos + 1});
    } else {
      ends.push_back(attempt);
    }
  }
  // rows with > 2 nodes (:420-421), then the post-filters of find_all_end_configurations
  const int L = target_length;
  auto count = [&](const std::vector<int>& c) {
    int k = 0;
    for (int v : c) k += (v != -1);
    return k;
  };
  std::vector<std::vector<int>> cur;
  for (auto& c : ends)
    if (count(c) > 2) cur.push_back(c);
  if (!first_k.empty() && !cur.empty()) {  // :484-486
    std::vector<std::vector<int>> keep;
    for (auto& c : cur) {
      bool ok = true;
      for (size_t q = 0; q < first_k.size(); q++)
        if ((int)q >= L || c[q] != first_k[q]) ok = false;
      if (ok) keep.push_back(c);
    }
    cur.swap(keep);
  }
  {  // :488-489
    std::vector<std::vector<int>> keep;
    for (auto& c : cur)
      if (count(c) >= 3) keep.push_back(c);
    cur.swap(keep);
  }
  for (auto& c : cur) {  // :491-500 drop last cone if not of the side's colour
    int am = 0;
    bool found = false;
    for (int q = 0; q < L; q++)
      if (c[q] == -1) {
        am = q;
        found = true;
        break;
      }
    if (!found) am = 0;
    int last_idx = ((am - 1) % L + L) % L;
    int last_cone = c[last_idx];
    if (f.type[last_cone] != cone_type) c[last_idx] = -1;
  }
  {  // :503-504
    std::vector<std::vector<int>> keep;
    for (auto& c : cur)
      if (count(c) >= 3) keep.push_back(c);
    cur.swap(keep);
  }
  // np.unique(axis=0): lexicographic sort + dedupe (:507)
  std::sort(cur.begin(), cur.end());
  cur.erase(std::unique(cur.begin(), cur.end()), cur.end());
  // remove rows that are a prefix of another (:509-515)
  std::vector<Config> out;
  for (size_t j = 0; j < cur.size(); j++) {
    int cnt = 0;
    for (size_t i = 0; i < cur.size(); i++) {
      bool all = true;
      for (int l = 0; l < L; l++)
        if (!(cur[i][l] == cur[j][l] || cur[j][l] == -1)) {
          all = false;
          break;
        }
      cnt += all;
    }
    if (!(cnt > 1)) {
      Config c;
      c.L = L;
      for (int l = 0; l < FSDO_MAX_LEN; l++) c.v[l] = (l < L) ? cur[j][l] : -1;
      out.push_back(c);
    }
  }
  return out;
}

// ---- S12: trace_sorter/nearby_cone_search.py:213-297 -------------------------------------
static std::vector<int> sorted_set_diff(const std::vector<int>& a, const std::vector<int>& b) {
  // nearby_cone_search.py:88-94: mask[np.searchsorted(a, b)] = False  (quirk: b need not be in a)
  std::vector<char> mask(a.size(), 1);
  for (int v : b) {
    size_t pos = std::lower_bound(a.begin(), a.end(), v) - a.begin();
    if (pos >= a.size()) throw RefUndefined{FSDO_REF_UNDEFINED_SET_DIFF};
    mask[pos] = 0;
  }
  std::vector<int> out;
  for (size_t i = 0; i < a.size(); i++)
    if (mask[i]) out.push_back(a[i]);
  return out;
}

// cone_matching/match_directions.py:7-20 calculate_search_direction_for_one
Vec2 search_direction(double x0, double y0, double x1, double y1, int cone_type) {
  double tx = x1 - x0, ty = y1 - y0;
  double ang = (cone_type == T_RIGHT) ? PI / 2 : -PI / 2;
  Rot r(ang);
  Vec2 d = r.apply(tx, ty);
  double nrm = norm2(d.x, d.y);
  return Vec2{d.x / nrm, d.y / nrm};
}

static void cones_on_each_side(const Frame& f, const std::vector<Config>& configs, int cone_type, std::vector<long>& good,
                               std::vector<long>& bad) {
  const int n = f.n;
  const double search_distance = 6.0, search_angle = PI / 1.5;
  std::vector<int> all;
  for (auto& c : configs)
    for (int l = 0; l < c.L; l++)
      if (c.v[l] != -1) all.push_back(c.v[l]);
  std::sort(all.begin(), all.end());
  all.erase(std::unique(all.begin(), all.end()), all.end());
  auto within = [&](int i, int j) {
    double d = (i == j) ? 1e7 : cdist_sq(f.x[i], f.y[i], f.x[j], f.y[j]);
    return d < search_distance * search_distance;
  };
  // find_nearby_cones_for_idxs :97-103
  std::vector<int> near_all;
  for (int j = 0; j < n; j++) {
    bool any = false;
    for (int i : all)
      if (within(i, j)) {
        any = true;
        break;
      }
    if (any) near_all.push_back(j);
  }
  std::vector<int> close = sorted_set_diff(near_all, all);
  good.assign(configs.size(), 0);
  bad.assign(configs.size(), 0);
  for (size_t ci = 0; ci < configs.size(); ci++) {
    std::vector<int> c;
    for (int l = 0; l < configs[ci].L; l++)
      if (configs[ci].v[l] != -1) c.push_back(configs[ci].v[l]);
    std::vector<int> extra = sorted_set_diff(all, c);
    std::vector<int> other = close;
    other.insert(other.end(), extra.begin(), extra.end());
    const int len = (int)c.size();
    for (int j = 0; j < len; j++) {
      int a, b;
      if (j == 0) {
        a = c[0];
        b = c[1];
      } else if (j == len - 1) {
        a = c[j - 1];
        b = c[j];
      } else {
        a = c[j - 1];
        b = c[j + 1];
      }
      Vec2 dir = search_direction(f.x[a], f.y[a], f.x[b], f.y[b], cone_type);
      int cj = c[j];
      for (int idx : other) {
        if (!within(cj, idx)) continue;
        double vx = f.x[idx] - f.x[cj], vy = f.y[idx] - f.y[cj];
        bool g = vec_angle_between(vx, vy, dir.x, dir.y) < search_angle / 2;
        bool bd = vec_angle_between(vx, vy, -dir.x, -dir.y) < search_angle / 2;
        margin(MG_ACOS_THRESHOLDS, std::cos(vec_angle_between(vx, vy, dir.x, dir.y)), std::cos(search_angle / 2));
        margin(MG_ACOS_THRESHOLDS, std::cos(vec_angle_between(vx, vy, -dir.x, -dir.y)), std::cos(search_angle / 2));
        good[ci] += g;
        bad[ci] += bd;
      }
    }
  }
}

// ---- S11: trace_sorter/cost_function.py:213-304 -----------------------------------------
static std::vector<double> cost_configurations(const Frame& f, const std::vector<Config>& configs, int cone_type) {
  const size_t C = configs.size();
  std::vector<double> costs(C, 0.0);
  if (C == 0) return costs;
  const int n = f.n;
  std::vector<long> good, bad;
  cones_on_each_side(f, configs, cone_type, good, bad);
  long mval = 0;
  for (size_t i = 0; i < C; i++) {
    long d = good[i] - bad[i];
    if (i == 0 || d < mval) mval = d;
  }
  const double fac_raw[7] = {1000.0, 200.0, 5000.0, 1000.0, 0.0, 1000.0, 1000.0};
  double fsum = np_sum(fac_raw, 7);
  double factors[7];
  for (int i = 0; i < 7; i++) factors[i] = fac_raw[i] / fsum;

  for (size_t ci = 0; ci < C; ci++) {
    const Config& cf = configs[ci];
    const int L = cf.L;
    auto P = [&](int l) -> Vec2 {  // points[configurations] with NumPy -1 wrap-around
      int idx = cf.v[l];
      if (idx < 0) idx = n + idx;
      return Vec2{f.x[idx], f.y[idx]};
    };
    int len = 0;
    for (int l = 0; l < L; l++) len += (cf.v[l] != -1);
    // angle cost :41-79 (calc_angle_to_next :23-38)
    double angle_cost;
    {
      std::vector<Vec2> to_next(L - 1);
      for (int l = 0; l < L - 1; l++) {
        Vec2 a = P(l), b = P(l + 1);
        to_next[l] = Vec2{a.x - b.x, a.y - b.y};
        if (cf.v[l + 1] == -1) to_next[l] = Vec2{100, 100};
      }
      int na = L - 2;
      std::vector<double> filtered(std::max(na, 0)), part(std::max(na, 0));
      long under = 0;
      for (int a = 0; a < na; a++) {
        Vec2 v1 = to_next[a + 1];
        Vec2 v2{-to_next[a].x, -to_next[a].y};
        double ang = vec_angle_between(v1.x, v1.y, v2.x, v2.y);
        bool is_part = cf.v[a + 2] != -1;
        double as_cost = (PI - ang) / PI;
        filtered[a] = as_cost * (is_part ? 1.0 : 0.0);
        part[a] = is_part ? 1.0 : 0.0;
        if (ang < deg2rad(40) && is_part) under++;
      }
      double s = np_sum(filtered);
      double cnt = np_sum(part);
      angle_cost = s / cnt * (double)(under + 1);
    }
    // residual distance cost: cone_distance_cost.py:15-32
    double dist_cost;
    {
      std::vector<double> resid(L - 1);
      for (int l = 0; l < L - 1; l++) {
        Vec2 a = P(l), b = P(l + 1);
        double dx = b.x - a.x, dy = b.y - a.y;
        double d = std::sqrt(dx * dx + dy * dy);
        d = d * ((cf.v[l + 1] != -1) ? 1.0 : 0.0);
        resid[l] = std::max(0.0, d - 3.0);
      }
      dist_cost = np_sum(resid);
    }
    double ncones_cost = 1.0 / (double)len;  // :82-96
    // initial direction cost :99-104
    double init_cost;
    {
      Vec2 a = P(0), b = P(1);
      init_cost = vec_angle_between(b.x - a.x, b.y - a.y, f.dx, f.dy);
    }
    double change_cost = 0.0;  // weight 0 (:283); finite for finite inputs
    double either_cost;
    {
      long d = good[ci] - bad[ci];
      d += std::labs(mval) + 1;
      either_cost = 1.0 / (double)d;
    }
    // wrong direction cost :149-188
    double wrong_cost = 0.0;
    if (len != 3) {
      std::vector<double> ang(len - 1);
      for (int l = 0; l < len - 1; l++) {
        Vec2 a = P(l), b = P(l + 1);
        ang[l] = std::atan2(b.y - a.y, b.x - a.x);
      }
      double unwanted = (cone_type == T_LEFT) ? 1.0 : -1.0;
      std::vector<double> sel;
      for (int l = 0; l + 1 < len - 1; l++) {
        double diff = angle_difference(ang[l], ang[l + 1]);
        margin(MG_WRONG_DIRECTION, diff, 0.0);
        if (np_sign(diff) == unwanted) margin(MG_WRONG_DIRECTION, std::fabs(diff), deg2rad(40));
        if (np_sign(diff) == unwanted && std::fabs(diff) > deg2rad(40)) sel.push_back(diff);
      }
      wrong_cost = std::fabs(np_sum(sel));
    }
    double cols[7] = {angle_cost * factors[0],  dist_cost * factors[1],   ncones_cost * factors[2], init_cost * factors[3],
                      change_cost * factors[4], either_cost * factors[5], wrong_cost * factors[6]};
    costs[ci] = np_sum(cols, 7);
  }
  return costs;
}

// ---- S3/S6: one side ---------------------------------------------------------------------
SideResult configs_for_one_side(const Frame& f, int cone_type) {
  SideResult res;
  res.has = false;
  res.first_k[0] = res.first_k[1] = -1;
  if (f.n < 3) return res;
  std::vector<int> first_k = select_first_k(f, cone_type);
  if (first_k.empty()) return res;
  for (size_t i = 0; i < first_k.size(); i++) res.first_k[i] = first_k[i];
  int start_idx = first_k[0];
  std::vector<int> must = (first_k.size() > 1) ? first_k : std::vector<int>{};
  int n_neighbors = std::min(g_prm.max_n_neighbors, f.n - 1);
  std::vector<std::vector<int>> nbrs;
  int n_reach = 0;
  create_adjacency(f, n_neighbors, start_idx, cone_type, nbrs, n_reach);
  int target_length = std::min(n_reach, g_prm.max_length);
  std::vector<Config> configs = find_all_end_configurations(f, cone_type, start_idx, nbrs, target_length, must);
  if (configs.empty()) return res;  // NoPathError
  std::vector<double> costs = cost_configurations(f, configs, cone_type);
  std::vector<int> order = argsort(costs);
  if (order.size() > 1) {
    const double c0 = costs[order[0]], c1 = costs[order[1]];
    margin(MG_COST_ARGMIN, (c1 - c0) / std::fmax(std::fabs(c0), 1e-300), 0.0);  // relative gap to the runner-up
  }
  for (int i : order) {
    res.configs.push_back(configs[i]);
    res.costs.push_back(costs[i]);
  }
  res.has = true;
  return res;
}

// ---- S14: trace_sorter/combine_traces.py ---------------------------------------------------
static double angle_change_at(const Frame& f, const std::vector<int>& cfg, int pos) {  // :260-275
  int p = cfg[pos - 1], c = cfg[pos], nx = cfg[pos + 1];
  double a_next = std::atan2(f.y[nx] - f.y[c], f.x[nx] - f.x[c]);
  double a_prev = std::atan2(f.y[p] - f.y[c], f.x[p] - f.x[c]);
  margin(MG_COMBINE, angle_difference(a_next, a_prev), 0.0);
  return angle_difference(a_next, a_prev);
}

static void handle_same_cone(const Frame& f, std::vector<int>& left, std::vector<int>& right) {  // :115-257
  int li = -1, ri = -1;
  for (size_t i = 0; i < left.size() && li < 0; i++)
    for (int v : right)
      if (v == left[i]) {
        li = (int)i;
        break;
      }
  if (li < 0) return;
  for (size_t i = 0; i < right.size() && ri < 0; i++)
    for (int v : left)
      if (v == right[i]) {
        ri = (int)i;
        break;
      }
  const int nl = (int)left.size(), nr = (int)right.size();
  int ls = -1, rs = -1;
  bool have = false;
  if (li > 0 && ri > 0) {
    int pl = left[li - 1], pr = right[ri - 1], ic = left[li];
    double dl = norm2(f.x[ic] - f.x[pl], f.y[ic] - f.y[pl]);
    double dr = norm2(f.x[ic] - f.x[pr], f.y[ic] - f.y[pr]);
    bool lv = dl < 3.0, rv = dr < 3.0;
    if ((lv || rv) && !(lv && rv)) {
      have = true;
      if (lv) {
        ls = nl;
        rs = ri;
      } else {
        ls = li;
        rs = nr;
      }
    }
  }
  if (!have && left[li] == right[ri] && (li >= 1 && li < nl - 1) && (ri >= 1 && ri < nr - 1)) {
    double al = angle_change_at(f, left, li);
    double ar = angle_change_at(f, right, ri);
    double sl = np_sign(al), sr = np_sign(ar);
    double absdiff = std::fabs(std::fabs(al) - std::fabs(ar));
    int ndiff = std::abs(nl - nr);
    if (sl == sr) {
      if (sl == 1) {
        ls = nl;
        rs = ri;
      } else {
        ls = li;
        rs = nr;
      }
    } else if (ndiff > 2) {
      if (nl > nr) {
        ls = nl;
        rs = ri;
      } else {
        ls = li;
        rs = nr;
      }
    } else if (absdiff > deg2rad(5)) {
      if (std::fabs(al) > std::fabs(ar)) {
        ls = nl;
        rs = ri;
      } else {
        ls = li;
        rs = nr;
      }
    } else {
      ls = li;
      rs = ri;
    }
  } else if (!have) {
    bool le = (li == nl - 1), re = (ri == nr - 1);
    if (le && re) {
      ls = nl - 1;
      rs = nr - 1;
    } else if (le) {
      rs = nr;
      ls = li;
    } else if (re) {
      ls = nl;
      rs = ri;
    } else {
      ls = li;
      rs = ri;
    }
  }
  left.resize(ls);
  right.resize(rs);
}

// trace_sorter/core_trace_sorter.py:148-216 sort_left_right (index part)
void sort_frame(const Frame& f, std::vector<int>& left, std::vector<int>& right, SideResult* lres_out,
                SideResult* rres_out) {
  SideResult L = configs_for_one_side(f, T_LEFT);
  SideResult R = configs_for_one_side(f, T_RIGHT);
  left.clear();
  right.clear();
  auto strip = [](const Config& c) {
    std::vector<int> o;
    for (int l = 0; l < c.L; l++)
      if (c.v[l] != -1) o.push_back(c.v[l]);
    return o;
  };
  if (L.has) left = strip(L.configs[0]);
  if (R.has) right = strip(R.configs[0]);
  if (L.has && R.has) handle_same_cone(f, left, right);
  if (lres_out) *lres_out = L;
  if (rres_out) *rres_out = R;
}

}  // namespace fsdo
