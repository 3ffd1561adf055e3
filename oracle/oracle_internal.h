// TEST INFRASTRUCTURE (oracle) — internal declarations shared by the oracle's translation units.
#pragma once
#include <cmath>
#include <vector>

#include "fsd_oracle.h"
#include "np_compat.h"
#include "fitpack.h"

namespace fsdo {

// The reference's configuration constants (fsd_path_planning/config.py:33-41,48,55-59,124-129); defaults = the factories'
// values.  One process-wide set (fsdo_set_params), read-only while frames are planned.
struct OParams {
  int max_n_neighbors = 5, max_length = 12;
  double max_dist = 6.5, max_dist_to_first = 6.0, threshold_directional_angle = 40 * (PI / 180.0),
         threshold_absolute_angle = 65 * (PI / 180.0);
  double min_track_width = 3.0, max_search_range = 5.0, max_search_angle = 50 * (PI / 180.0);
  double smoothing = 0.2, predict_every = 0.1, maximal_distance_for_valid_path = 5.0, mpc_path_length = 20.0;
  // config.py:48 max_deg, :58 mpc_prediction_horizon, :124-146 matches_should_be_monotonic (the pipeline's choice: False,
  // full_pipeline.py:65), :40 use_unknown_cones
  int max_deg = 3, horizon = 40, matches_should_be_monotonic = 0, use_unknown_cones = 1;
};
extern OParams g_prm;
struct Spline;
extern thread_local std::vector<Spline>* g_fit_capture;
void rebuild_default_previous_path();

struct Frame {
  std::vector<int> orig;  // index of every cone in the caller's array (cones of type UNKNOWN are dropped when use_unknown_cones is off)
  int n = 0;
  std::vector<double> x, y;
  std::vector<int> type;
  double px = 0, py = 0, dx = 1, dy = 0;
};

struct Config {
  int L = 12;            // target_length of the search that produced it
  int v[FSDO_MAX_LEN];   // -1 padded
};

struct SideResult {
  bool has = false;
  std::vector<Config> configs;  // sorted by cost
  std::vector<double> costs;
  int first_k[2] = {-1, -1};
};

typedef std::vector<Vec2> Pts;

// Decision margins of the sorting stage (diagnostics, tools/sort_margins.py): every discrete decision that hangs on a libm
// value — atan2 / acos compared with a threshold, or the arg-min over configuration costs that hold such values — records
// how far the value was from the threshold / the runner-up.  Off unless fsdo_margins_enable(1); single-threaded use.
enum MarginClass {
  MG_START_BEARING = 0,   // core_trace_sorter.py:379-407: bearing sign, pi/10, pi - pi/5
  MG_SECOND_SIDE = 1,     // end_configurations.py:260-278: sign of the bearing difference, 5 deg
  MG_ABS_ANGLE = 2,       // :172-205 |difference| vs threshold_absolute_angle
  MG_DIR_ANGLE = 3,       // :172-205 difference vs +-threshold_directional_angle
  MG_SIGN_FLIP = 4,       // :196-205 sign(difference) vs sign(difference_2), |difference - difference_2| vs 1.3
  MG_ACOS_THRESHOLDS = 5, // vec_angle_between vs 150 deg / 90 deg / search_angle / 2 / 40 deg (in the cosine: |cos - cos thr|)
  MG_WRONG_DIRECTION = 6, // cost_function.py:149-188 sign and 40 deg of the turn angles
  MG_COST_ARGMIN = 7,     // relative gap between the best and the second-best configuration cost
  MG_COMBINE = 8,         // combine_traces.py:150-257 angle signs / 5 deg
  MG_CLASSES = 9
};
struct MarginRec {
  bool on = false;
  double min_margin[MG_CLASSES];
  long long n[MG_CLASSES], below_1e6[MG_CLASSES], below_1e9[MG_CLASSES], below_1e12[MG_CLASSES], zero[MG_CLASSES];
};
extern MarginRec g_margins;
static inline void margin(int cls, double value, double thr) {
  if (!g_margins.on) return;
  const double m = std::fabs(value - thr);
  if (!(m == m)) return;
  MarginRec& r = g_margins;
  r.n[cls]++;
  if (m == 0.0) {  // value == threshold exactly: equal operands (straight / lattice tracks), the same bits under any libm
    r.zero[cls]++;
    return;
  }
  if (m < r.min_margin[cls]) r.min_margin[cls] = m;
  if (m < 1e-6) r.below_1e6[cls]++;
  if (m < 1e-9) r.below_1e9[cls]++;
  if (m < 1e-12) r.below_1e12[cls]++;
}

// sorting.cpp
SideResult configs_for_one_side(const Frame& f, int cone_type);
void sort_frame(const Frame& f, std::vector<int>& left, std::vector<int>& right, SideResult* l = nullptr,
                SideResult* r = nullptr);
Vec2 search_direction(double x0, double y0, double x1, double y1, int cone_type);
bool segments_intersect(Vec2 a0, Vec2 a1, Vec2 b0, Vec2 b1);

// matching.cpp
void match_cones(const Pts& left, const Pts& right, Vec2 car_pos, Pts& left_v, Pts& right_v, std::vector<int>& l2r,
                 std::vector<int>& r2l);

// path.cpp
struct PathOut {
  double p[FSDO_PATH_POINTS][4];
  int fallback = 0;
};
const double (*default_previous_path())[4];
void calculate_path(const Pts& left_v, const Pts& right_v, const std::vector<int>& l2r, const std::vector<int>& r2l,
                    Vec2 pos, Vec2 dir, PathOut& out, const double (*prev)[4] = nullptr, const Pts* global_path = nullptr);

void finish_path(Pts path_update, const Pts& prev_xy, Vec2 pos, Vec2 dir, PathOut& out);
Pts almost_straight_path();
void circle_fit(const Pts& p, double& ocx, double& ocy, double& orad);
void do_all_mpc(const Pts& path_update, Vec2 pos, Vec2 dir, double out[][4], int* flags);
double m_atan2(double y, double x);
double m_cos(double a);
double m_sin(double a);

}  // namespace fsdo
