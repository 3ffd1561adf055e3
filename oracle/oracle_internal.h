// TEST INFRASTRUCTURE (oracle) — internal declarations shared by the oracle's translation units.
#pragma once
#include <vector>

#include "fsd_oracle.h"
#include "np_compat.h"
#include "fitpack.h"

namespace fsdo {

struct Frame {
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
