// TEST INFRASTRUCTURE (oracle) — skidpad mission (BASELINE config 5), scalar restatement with NumPy semantics.
// Reference: full_pipeline/full_pipeline.py:118-194 (relocalizer branch), relocalization/relocalization_base_class.py:50-95,
// relocalization/skidpad/skidpad_relocalizer.py:31-243, relocalization/relocalization_information.py:12-35,
// calculate_path/skidpad_calculate_path.py:49-71, calculate_path/core_calculate_path.py:127-134,514-575.
// Stateful per planner instance: relocalization transform (+ the pose latched at the first attempt), index_along_path,
// previous path.  The known skidpad path table (skidpad_path_data.py, 5786 x 2) and the noise table
// RandomState(42).randn(1140,3,2)*1e-3 (circle_fit_powerset re-seeds on every call, :38) are DATA handed in by the caller.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "oracle_internal.h"

namespace fsdo {

struct SkidpadPlanner {
  Pts table;            // BASE_SKIDPAD_PATH (full)
  Pts global_path;      // table[::2]
  std::vector<double> noise;  // (n_subsets, 3, 2) * 1e-3 already applied? -> raw randn values, scaled here
  Vec2 ref_right{0, 0}, ref_left{0, 0};  // calculate_reference_centers_for_skidpad_path :172-183
  bool has_original = false;
  Vec2 orig_pos{0, 0}, orig_dir{1, 0};
  bool relocalized = false;
  Vec2 translation{0, 0}, right_calc{0, 0};
  double rotation = 0.0;
  int index_along_path = 0;
  double prev[FSDO_PATH_POINTS][4];
};

static Vec2 rot_point(double theta_cos, double theta_sin, double x, double y) {
  return Vec2{blas_dot2(x, theta_cos, y, -theta_sin), blas_dot2(x, theta_sin, y, theta_cos)};
}
// rotate() of a single point: np.dot(1-D, 2 x 2) is gemv (np_compat.h blas_dot2_single_row)
static Vec2 rot_single_point(double theta_cos, double theta_sin, double x, double y) {
  return Vec2{blas_dot2_single_row(x, theta_cos, y, -theta_sin), blas_dot2_single_row(x, theta_sin, y, theta_cos)};
}

// skidpad_relocalizer.py:198-240 do_relocalization_once; true on success
static bool relocalize(SkidpadPlanner& P, const double* xyt, int n, Vec2 pos) {
  // 20 closest cones (np.argsort of np.linalg.norm(axis=1))
  std::vector<double> dist(n);
  for (int i = 0; i < n; i++) dist[i] = norm2_axis(xyt[3 * i] - pos.x, xyt[3 * i + 1] - pos.y);
  std::vector<int> order = argsort(dist);
  int m = std::min(n, 20);
  Pts pts(m);
  for (int i = 0; i < m; i++) pts[i] = Vec2{xyt[3 * order[i]], xyt[3 * order[i] + 1]};
  // circle_fit_powerset :31-64 (only 3-subsets are ever enumerated: the loop variable shadows `idxs`)
  std::vector<Vec2> centers;
  if (m >= 3) {
    size_t s = 0;
    for (int a = 0; a < m; a++)
      for (int b = a + 1; b < m; b++)
        for (int c = b + 1; c < m; c++, s++) {
          int id[3] = {a, b, c};
          // distance to the closest other point of the set: sqrt of the expansion-form d^2 (diag inf), min over axis 0, mean
          double mind[3];
          for (int col = 0; col < 3; col++) {
            double best = 0.0;
            bool first = true;
            for (int row = 0; row < 3; row++) {
              // (d2 ** 0.5: NumPy evaluates a scalar exponent 0.5 as sqrt); diagonal is inf
              double d = (row == col) ? INFINITY : std::sqrt(cdist_sq(pts[id[row]].x, pts[id[row]].y, pts[id[col]].x, pts[id[col]].y));
              if (first) {
                best = d;
                first = false;
              } else if (std::isnan(d) || std::isnan(best)) {
                best = NAN;  // np.min propagates NaN
              } else if (d < best) {
                best = d;
              }
            }
            mind[col] = best;
          }
          double mean_distance = np_sum(mind, 3) / 3.0;
          if ((s + 1) * 6 > P.noise.size()) throw RefUndefined{FSDO_REF_UNDEFINED_OTHER};
          Pts q(3);
          for (int r = 0; r < 3; r++)
            q[r] = Vec2{pts[id[r]].x + P.noise[s * 6 + 2 * r] * 1e-3, pts[id[r]].y + P.noise[s * 6 + 2 * r + 1] * 1e-3};
          double cx, cy, rad;
          circle_fit(q, cx, cy, rad);
          double res[3];
          for (int r = 0; r < 3; r++) res[r] = std::fabs(norm2_axis(cx - q[r].x, cy - q[r].y) - rad);
          double residual = np_sum(res, 3) / 3.0;
          if (std::fabs(rad - 7.625) < 1.0 && std::fabs(mean_distance - 2.4) < 1.5 && residual < 0.4) centers.push_back(Vec2{cx, cy});
        }
  }
  if (centers.size() < 3) return false;
  // calculate_circle_centers :67-98 — DBSCAN(eps=3, min_samples=1) = connected components of the <= 3 m graph,
  // labels numbered by first occurrence
  const int C = (int)centers.size();
  std::vector<int> label(C, -1);
  int n_labels = 0;
  for (int i = 0; i < C; i++) {
    if (label[i] >= 0) continue;
    label[i] = n_labels;
    std::vector<int> stack{i};
    while (!stack.empty()) {
      int u = stack.back();
      stack.pop_back();
      for (int v = 0; v < C; v++) {
        if (label[v] >= 0) continue;
        double dx = centers[u].x - centers[v].x, dy = centers[u].y - centers[v].y;
        if (std::sqrt(dx * dx + dy * dy) <= 3.0) {
          label[v] = n_labels;
          stack.push_back(v);
        }
      }
    }
    n_labels++;
  }
  if (!(n_labels > 1)) return false;  // AssertionError
  auto median_of = [&](int lab) {
    std::vector<double> xs, ys;
    for (int i = 0; i < C; i++)
      if (label[i] == lab) {
        xs.push_back(centers[i].x);
        ys.push_back(centers[i].y);
      }
    auto med = [](std::vector<double>& v) {
      std::sort(v.begin(), v.end());
      size_t k = v.size();
      if (k % 2 == 1) return v[k / 2];
      double two[2] = {v[k / 2 - 1], v[k / 2]};
      return np_sum(two, 2) / 2.0;  // np.median -> mean of the two middle values
    };
    return Vec2{med(xs), med(ys)};
  };
  double best_distance = 1000;
  Vec2 bc0{0, 0}, bc1{0, 0};
  for (int l1 = 0; l1 < n_labels; l1++)
    for (int l2 = l1 + 1; l2 < n_labels; l2++) {
      Vec2 c0 = median_of(l1), c1 = median_of(l2);
      double d = std::fabs(18.25 - norm2(c0.x - c1.x, c0.y - c1.y));
      if (d < best_distance) {
        best_distance = d;
        bc0 = c0;
        bc1 = c1;
      }
    }
  if (best_distance > 0.5) return false;  // ValueError
  // calculate_transformation :101-169 (uses the pose latched at the FIRST attempt)
  double yaw0 = m_atan2(P.orig_dir.y, P.orig_dir.x);
  double c = m_cos(-yaw0), s = m_sin(-yaw0);
  Vec2 cc[2] = {bc0, bc1};
  bool is_right[2];
  for (int i = 0; i < 2; i++) {
    Vec2 v = rot_point(c, s, cc[i].x - P.orig_pos.x, cc[i].y - P.orig_pos.y);
    is_right[i] = v.y < 0.0;
  }
  int ir = -1, il = -1;
  for (int i = 0; i < 2; i++) {
    if (is_right[i] && ir < 0) ir = i;
    if (!is_right[i] && il < 0) il = i;
  }
  if (ir < 0 || il < 0) return false;  // IndexError
  Vec2 right_calc = cc[ir], left_calc = cc[il];
  P.translation = Vec2{P.ref_right.x - right_calc.x, P.ref_right.y - right_calc.y};
  double reference_angle = m_atan2(P.ref_left.y - P.ref_right.y, P.ref_left.x - P.ref_right.x);
  double calculated_angle = m_atan2(left_calc.y - right_calc.y, left_calc.x - right_calc.x);
  P.rotation = reference_angle - calculated_angle;
  P.right_calc = right_calc;
  P.relocalized = true;
  return true;
}

static void skidpad_step(SkidpadPlanner& P, const double* xyt, int n, const double* pose, fsdo_frame_result* o, double* info) {
  Vec2 pos{pose[0], pose[1]}, dir{pose[2], pose[3]};
  // Relocalizer.attempt_relocalization_calculation (relocalization_base_class.py:50-75)
  if (!P.relocalized) {
    if (!P.has_original) {
      P.has_original = true;
      P.orig_pos = pos;
      P.orig_dir = dir;
    }
    if (n == 0) {
      // np.row_stack of five empty (0,2) arrays works; argsort of empty -> no subsets -> < 3 circles
    }
    relocalize(P, xyt, n, pos);
  }
  Pts path_update;
  if (P.relocalized) {
    // full_pipeline.py:126-134: pose -> known map frame
    double yaw = m_atan2(dir.y, dir.x);
    double c = m_cos(P.rotation), s = m_sin(P.rotation);
    Vec2 t{pos.x + P.translation.x - P.ref_right.x, pos.y + P.translation.y - P.ref_right.y};
    Vec2 r = rot_single_point(c, s, t.x, t.y);  // (position_2d is one point: skidpad_relocalizer.py:140-147)
    pos = Vec2{r.x + P.ref_right.x, r.y + P.ref_right.y};
    yaw = yaw + P.rotation;
    dir = Vec2{m_cos(yaw), m_sin(yaw)};
    // SkidpadCalculatePath.fit_matches_as_spline (skidpad_calculate_path.py:49-71)
    const Pts& g = P.global_path;
    double d9[9];
    for (int i = 0; i < 9; i++) d9[i] = norm2_axis(g[i + 1].x - g[i].x, g[i + 1].y - g[i].y);
    double mean_distance = np_sum(d9, 9) / 9.0;
    int max_allowed_change = (int)(20 / mean_distance);
    int min_index = std::max(P.index_along_path - max_allowed_change, 0);
    int max_index = std::min(P.index_along_path + max_allowed_change, (int)g.size());
    int best = 0;
    double bd = 0;
    for (int i = min_index; i < max_index; i++) {
      double d = norm2_axis(pos.x - g[i].x, pos.y - g[i].y);
      if (i == min_index || d < bd) {
        bd = d;
        best = i;
      }
    }
    if (max_index <= min_index) throw RefUndefined{FSDO_REF_UNDEFINED_PATH};  // argmin of empty
    P.index_along_path = best;
    int final_index = std::min(best + (int)(25 / mean_distance), (int)g.size());
    path_update.assign(g.begin() + best, g.begin() + final_index);
  } else {
    // calculate_trivial_path (core_calculate_path.py:127-134): chord[1:] rotated by the car yaw + position
    Pts chord = almost_straight_path();
    double yaw = m_atan2(dir.y, dir.x);
    double c = m_cos(yaw), s = m_sin(yaw);
    for (int i = 1; i < (int)chord.size(); i++) {
      Vec2 r = rot_point(c, s, chord[i].x, chord[i].y);
      path_update.push_back(Vec2{r.x + pos.x, r.y + pos.y});
    }
  }
  Pts prev_xy(g_prm.horizon);
  for (int i = 0; i < g_prm.horizon; i++) prev_xy[i] = Vec2{P.prev[i][1], P.prev[i][2]};
  PathOut po;
  po.fallback = 0;
  finish_path(path_update, prev_xy, pos, dir, po);
  std::memcpy(P.prev, po.p, sizeof(po.p));  // previous_paths[-1] for the next frame (map frame)
  std::memcpy(o->path, po.p, sizeof(po.p));
  o->path_fallback = po.fallback;
  if (P.relocalized) {
    // full_pipeline.py:178-194: path xy back to the original frame
    double c = m_cos(-P.rotation), s = m_sin(-P.rotation);
    for (int i = 0; i < FSDO_PATH_POINTS; i++) {
      Vec2 t{o->path[i][1] - P.translation.x - P.right_calc.x, o->path[i][2] - P.translation.y - P.right_calc.y};
      Vec2 r = rot_point(c, s, t.x, t.y);
      o->path[i][1] = r.x + P.right_calc.x;
      o->path[i][2] = r.y + P.right_calc.y;
    }
  }
  if (info) {
    // RelocalizationInformation.from_transform_function (relocalization_information.py:12-35)
    info[0] = P.relocalized ? 1.0 : 0.0;
    info[1] = info[2] = info[3] = NAN;
    if (P.relocalized) {
      double c = m_cos(P.rotation), s = m_sin(P.rotation);
      auto tf = [&](double x, double y) {
        Vec2 t{x + P.translation.x - P.ref_right.x, y + P.translation.y - P.ref_right.y};
        Vec2 r = rot_single_point(c, s, t.x, t.y);  // (relocalization_information.py:21-27 transforms single points)
        return Vec2{r.x + P.ref_right.x, r.y + P.ref_right.y};
      };
      Vec2 o0 = tf(0.0, 0.0), o1 = tf(1.0, 0.0);
      info[1] = o0.x;
      info[2] = o0.y;
      info[3] = std::atan2(o1.y - o0.y, o1.x - o0.x);
    }
    info[4] = (double)P.index_along_path;
  }
}

}  // namespace fsdo

using namespace fsdo;

extern "C" {

void* fsdo_skidpad_create(const double* table_xy, int n_table, const double* noise_randn, int n_noise) {
  SkidpadPlanner* P = new SkidpadPlanner();
  P->table.resize(n_table);
  for (int i = 0; i < n_table; i++) P->table[i] = Vec2{table_xy[2 * i], table_xy[2 * i + 1]};
  for (int i = 0; i < n_table; i += 2) P->global_path.push_back(P->table[i]);  // BASE_SKIDPAD_PATH[::2]
  P->noise.assign(noise_randn, noise_randn + n_noise);
  // calculate_reference_centers_for_skidpad_path :172-183
  Pts neg, posy;
  for (auto& p : P->table) {
    if (p.y < -2) neg.push_back(p);
    if (p.y > 2) posy.push_back(p);
  }
  double cx, cy, r;
  circle_fit(neg, cx, cy, r);
  P->ref_right = Vec2{cx, cy};
  circle_fit(posy, cx, cy, r);
  P->ref_left = Vec2{cx, cy};
  std::memcpy(P->prev, default_previous_path(), sizeof(P->prev));
  return P;
}

void fsdo_skidpad_destroy(void* h) { delete (SkidpadPlanner*)h; }

// one frame of one planner instance; info: [relocalized, translation x, y, rotation, index_along_path]
void fsdo_skidpad_step(void* h, const double* cones_xyt, int n, const double* pose, fsdo_frame_result* out, double* info) {
  SkidpadPlanner* P = (SkidpadPlanner*)h;
  std::memset(out, 0, sizeof(*out));
  for (int i = 0; i < FSDO_MAX_LEN; i++) out->left_idx[i] = out->right_idx[i] = -1;
  for (int i = 0; i < FSDO_MAX_MATCH; i++) out->l2r[i] = out->r2l[i] = -1;
  try {
    skidpad_step(*P, cones_xyt, n, pose, out, info);
  } catch (RefUndefined& e) {
    out->status = e.code;
  } catch (PyValueError&) {
    out->status = FSDO_REF_UNDEFINED_PATH;
  }
}

// SkidpadCalculatePath.index_along_path as it stands (a step that raised reports nothing, but may have moved it)
int fsdo_skidpad_index(void* h) { return ((SkidpadPlanner*)h)->index_along_path; }

// the relocalization transform as the planner holds it: translation x, y, rotation, calculated right centre x, y
void fsdo_skidpad_transform(void* h, double* out5) {
  SkidpadPlanner* P = (SkidpadPlanner*)h;
  out5[0] = P->translation.x;
  out5[1] = P->translation.y;
  out5[2] = P->rotation;
  out5[3] = P->right_calc.x;
  out5[4] = P->right_calc.y;
}

void fsdo_skidpad_reference_centers(void* h, double* out4) {
  SkidpadPlanner* P = (SkidpadPlanner*)h;
  out4[0] = P->ref_right.x;
  out4[1] = P->ref_right.y;
  out4[2] = P->ref_left.x;
  out4[3] = P->ref_left.y;
}
}
