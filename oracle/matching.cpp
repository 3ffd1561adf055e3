// TEST INFRASTRUCTURE (oracle) — cone_matching step, scalar restatement with NumPy semantics.
// Reference: cone_matching/functional_cone_matching.py, cone_matching/match_directions.py,
// cone_matching/core_cone_matching.py:87-124 (major = 5*1.5 = 7.5, minor = 3, angle 50 deg,
// matches_should_be_monotonic False via config.py:148-163).
#include <cmath>
#include <vector>
#include <algorithm>

#include "oracle_internal.h"

namespace fsdo {

static const int T_RIGHT = 1, T_LEFT = 2;
// core_cone_matching.py:101-102: major_radius = max_search_range * 1.5, minor_radius = min_track_width
#define MIN_TRACK_WIDTH (g_prm.min_track_width)
#define MAJOR_RADIUS (g_prm.max_search_range * 1.5)
#define MINOR_RADIUS (g_prm.min_track_width)

// match_directions.py:23-44 calculate_match_search_direction
static Pts match_search_directions(const Pts& c, int cone_type) {
  const int n = (int)c.size();
  Pts out(n);
  out[0] = search_direction(c[0].x, c[0].y, c[1].x, c[1].y, cone_type);
  out[n - 1] = search_direction(c[n - 2].x, c[n - 2].y, c[n - 1].x, c[n - 1].y, cone_type);
  for (int i = 1; i < n - 1; i++) out[i] = search_direction(c[i - 1].x, c[i - 1].y, c[i + 1].x, c[i + 1].y, cone_type);
  return out;
}

// functional_cone_matching.py:73-144 reduced to what select_best_match_candidate consumes
// (:174 only uses mask.any(axis=1); the "two nearest" pruning :132-142 keeps any() unchanged).
static std::vector<char> has_potential_match(const Pts& start, const Pts& dirs, const Pts& other, const Pts& other_dirs) {
  const int M = (int)start.size(), N = (int)other.size();
  std::vector<char> any(M, 0);
  if (M == 0 || N == 0) return any;
  const double max_search_angle = g_prm.max_search_angle;
  const double r0 = MAJOR_RADIUS * MAJOR_RADIUS, r1 = MINOR_RADIUS * MINOR_RADIUS;
  for (int i = 0; i < M; i++) {
    double ang = std::atan2(dirs[i].y, dirs[i].x);
    Rot rot(-ang);
    for (int j = 0; j < N; j++) {
      Vec2 q = rot.apply(other[j].x - start[i].x, other[j].y - start[i].y);
      double s = (q.x * q.x) / r0 + (q.y * q.y) / r1;
      bool ok = s < 1;
      double a = std::atan2(q.y, q.x);
      if (std::fabs(a / 2) > max_search_angle) ok = false;
      if (!other_dirs.empty()) {  // empty (N==1) boolean mask indexes nothing in NumPy
        double dd = vec_angle_between(dirs[i].x, dirs[i].y, other_dirs[j].x, other_dirs[j].y);
        if (dd < PI / 2) ok = false;
      }
      if (ok) any[i] = 1;
    }
  }
  return any;
}

// functional_cone_matching.py:340-384 calculate_matches_for_side (+ :147-175)
static void matches_for_side(const Pts& cones, int cone_type, const Pts& other, std::vector<int>& matches, Pts& dirs) {
  const int M = (int)cones.size(), N = (int)other.size();
  matches.assign(M, -1);
  dirs.clear();
  if (M > 1) {
    dirs = match_search_directions(cones, cone_type);
    Pts other_dirs;
    if (N > 1) other_dirs = match_search_directions(other, cone_type == T_RIGHT ? T_LEFT : T_RIGHT);
    std::vector<char> any = has_potential_match(cones, dirs, other, other_dirs);
    if (N == 0) return;
    for (int i = 0; i < M; i++) {
      int best = 0;
      double bd = 0;
      for (int j = 0; j < N; j++) {
        double d = cdist_sq(cones[i].x, cones[i].y, other[j].x, other[j].y);
        if (j == 0 || d < bd) {
          bd = d;
          best = j;
        }
      }
      matches[i] = best;
    }
    if (g_prm.matches_should_be_monotonic) {  // functional_cone_matching.py:164-171
      int current_max = matches[0];
      for (int i = 1; i < M; i++) {
        current_max = std::max(current_max, matches[i]);
        matches[i] = (matches[i] != current_max) ? -1 : current_max;
      }
    }
    for (int i = 0; i < M; i++)
      if (!any[i]) matches[i] = -1;  // :174
  }
}

// functional_cone_matching.py:195-261 insert_virtual_cones_to_existing
static Pts insert_virtual(const Pts& other, const Pts& virt, Vec2 car) {
  Pts existing, to_insert;
  if (other.size() > virt.size()) {
    existing = other;
    to_insert = virt;
  } else {
    existing = virt;
    to_insert = other;
  }
  std::vector<double> mind(to_insert.size());
  for (size_t i = 0; i < to_insert.size(); i++) {
    double b = 0;
    for (size_t j = 0; j < existing.size(); j++) {
      double d = cdist_sq(to_insert[i].x, to_insert[i].y, existing[j].x, existing[j].y);
      if (j == 0 || d < b) b = d;
    }
    mind[i] = b;
  }
  std::vector<int> order = argsort(mind);
  for (int oi : order) {
    Vec2 cone = to_insert[oi];
    std::vector<double> dist(existing.size());
    for (size_t j = 0; j < existing.size(); j++) {
      double dx = existing[j].x - cone.x, dy = existing[j].y - cone.y;
      dist[j] = std::sqrt(dx * dx + dy * dy);
    }
    std::vector<int> srt = argsort(dist);
    int index_to_insert;
    if (srt.size() == 1) {
      // calculate_insert_index_for_one_cone :264-282
      double d_other = norm2(cone.x - car.x, cone.y - car.y);
      double d_exist = norm2(existing[0].x - car.x, existing[0].y - car.y);
      index_to_insert = (d_other < d_exist) ? 0 : 1;
    } else {
      int closest = srt[0], second = srt[1];
      if (std::abs(closest - second) != 1) continue;
      double ax = existing[closest].x - cone.x, ay = existing[closest].y - cone.y;
      double bx = existing[second].x - cone.x, by = existing[second].y - cone.y;
      bool between = vec_angle_between(ax, ay, bx, by) > PI / 2;
      if (between)
        index_to_insert = std::min(closest, second) + 1;
      else
        index_to_insert = (closest < second) ? closest : closest + 1;
    }
    existing.insert(existing.begin() + index_to_insert, cone);
  }
  // trace_angles_between (utils/math_utils.py:237-252) < 85 deg -> drop interior cones
  const int n = (int)existing.size();
  if (n >= 3) {
    std::vector<char> low(n, 0);
    bool anylow = false;
    for (int i = 1; i < n - 1; i++) {
      double nx = existing[i + 1].x - existing[i].x, ny = existing[i + 1].y - existing[i].y;
      double px = -(existing[i].x - existing[i - 1].x), py = -(existing[i].y - existing[i - 1].y);
      double a = vec_angle_between(nx, ny, px, py);
      if (a < deg2rad(85)) {
        low[i] = 1;
        anylow = true;
      }
    }
    if (anylow) {
      Pts kept;
      for (int i = 0; i < n; i++)
        if (!low[i]) kept.push_back(existing[i]);
      existing.swap(kept);
    }
  }
  return existing;
}

// functional_cone_matching.py:387-440 calculate_cones_for_other_side
static Pts cones_for_other_side(const Pts& cones, int cone_type, const Pts& other, Vec2 car) {
  std::vector<int> matches;
  Pts dirs;
  matches_for_side(cones, cone_type, other, matches, dirs);
  Pts virt;
  for (size_t i = 0; i < cones.size(); i++)
    if (matches[i] == -1) virt.push_back(Vec2{cones[i].x + dirs[i].x * MIN_TRACK_WIDTH, cones[i].y + dirs[i].y * MIN_TRACK_WIDTH});
  Pts combined;
  if (other.empty())
    combined = virt;
  else if (virt.empty())
    combined = other;
  else
    combined = insert_virtual(other, virt, car);
  if (combined.size() < 2) combined = other;
  return combined;
}

// functional_cone_matching.py:479-588 calculate_virtual_cones_for_both_sides
void match_cones(const Pts& left_in, const Pts& right_in, Vec2 car, Pts& left_v, Pts& right_v, std::vector<int>& l2r,
                 std::vector<int>& r2l) {
  Pts left = left_in, right = right_in;
  left_v.clear();
  right_v.clear();
  l2r.clear();
  r2l.clear();
  if (left.size() < 2 && right.size() < 2) return;
  size_t min_len = std::min(left.size(), right.size());
  size_t max_len = std::max(left.size(), right.size());
  bool discard = (min_len == 0) || (((double)max_len / (double)min_len) > 2);
  if (discard) {
    if (left.size() < right.size())
      left.clear();
    else
      right.clear();
  }
  right_v = (left.size() >= 2) ? cones_for_other_side(left, T_LEFT, right, car) : right;
  left_v = (right.size() >= 2) ? cones_for_other_side(right, T_RIGHT, left, car) : left;
  Pts d;
  matches_for_side(left_v, T_LEFT, right_v, l2r, d);
  matches_for_side(right_v, T_RIGHT, left_v, r2l, d);
}

}  // namespace fsdo
