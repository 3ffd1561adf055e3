// TEST INFRASTRUCTURE — CPU oracle for PathPlanner.calculate_path_in_global_frame
// (reference full_pipeline/full_pipeline.py:84-207, trackdrive/autocross branch).
// A scalar C++ restatement with NumPy semantics; pinned against golden vectors captured
// from the reference import (tests/golden/).  Only tests/, bench.py's cpu_baseline leg and
// __graft_entry__.smoke() may load this; the product never links or calls it.
#pragma once
#include <cstdint>

#ifdef __cplusplus
extern "C" {
#endif

// Record shapes: the standard ones (the reference's default structural parameters) or, with -DFSDO_WIDE_SHAPES
// (liboracle_wide.so), the shapes of the library's wide build (include/fsdp.h FSDP_WIDE_SHAPES).  The algorithm itself takes
// any max_n_neighbors / max_length / horizon; only the result record is shaped.
#ifdef FSDO_WIDE_SHAPES
enum {
  FSDO_MAX_LEN = 16,       // config.py:36 max_length
  FSDO_MAX_MATCH = 32,     // cones-with-virtual per side: base (<=16) + inserted (<=16)
  FSDO_PATH_POINTS = 64,   // config.py:58 mpc_prediction_horizon
};
#else
enum {
  FSDO_MAX_LEN = 12,       // config.py:36 max_length
  FSDO_MAX_MATCH = 24,     // cones-with-virtual per side: base (<=12) + inserted (<=12)
  FSDO_PATH_POINTS = 40,   // config.py:58 mpc_prediction_horizon
};
#endif

// Per-frame status.  0 = the reference returns normally and so does the oracle.
// >= 100: the reference raises out of calculate_path_in_global_frame (or would corrupt
// memory under numba) on this input — "REF_UNDEFINED", excluded from parity.
enum {
  FSDO_OK = 0,
  FSDO_REF_UNDEFINED_SET_DIFF = 101,   // nearby_cone_search.py:88-94 searchsorted index == len
  FSDO_REF_UNDEFINED_DFS_OOB = 102,    // end_configurations.py:369 current_attempt[pos] with pos >= target_length
  FSDO_REF_UNDEFINED_PATH = 103,       // path stage raises after its own fallback
  FSDO_REF_UNDEFINED_MATCH_IDX = 104,  // matching / centre points index an empty array
  FSDO_REF_UNDEFINED_OTHER = 109,
};

typedef struct {
  int32_t status;
  // sorting (S2): trace_sorter/core_trace_sorter.py:197-214 left_config/right_config
  int32_t n_left, n_right;
  int32_t left_idx[FSDO_MAX_LEN];
  int32_t right_idx[FSDO_MAX_LEN];
  // matching (M1)
  int32_t n_left_v, n_right_v;
  double left_v[FSDO_MAX_MATCH][2];
  double right_v[FSDO_MAX_MATCH][2];
  int32_t l2r[FSDO_MAX_MATCH];
  int32_t r2l[FSDO_MAX_MATCH];
  // path (P1): (40,4) [u, x, y, curvature]
  double path[FSDO_PATH_POINTS][4];
  // diagnostics
  int32_t n_configs_left, n_configs_right;   // C after post-filters (0 = side has no result)
  int32_t first_k_left[2], first_k_right[2]; // -1 padded
  double best_cost_left, best_cost_right;
  int32_t path_fallback;                      // bit0 centre<2 -> previous path, bit1 spline ValueError, bit2 path too far, bit3 mpc ValueError retry, bit4 arc extension, bit5 straight extension
} fsdo_frame_result;

// cones_xyt: (n,3) row-major [x,y,type]; pose: [px,py,dx,dy].
void fsdo_plan_frame(const double* cones_xyt, int n, const double* pose, fsdo_frame_result* out);
// the same with a previous path (40,4) (or NULL) and PathPlanner.global_path (n_gpath,2) (or NULL)
void fsdo_plan_frame_global(const double* cones_xyt, int n, const double* pose, const double* prev40x4, const double* gpath_xy,
                            int n_gpath, fsdo_frame_result* out);

// batch over CSR offsets; n_threads <= 1 runs serially, otherwise std::thread workers.
void fsdo_plan_batch(int n_frames, const int32_t* cone_offsets, const double* cones_xyt, const double* poses,
                     fsdo_frame_result* out, int n_threads);

// stage-level entry points (used by stage parity tests)
void fsdo_sort_frame(const double* cones_xyt, int n, const double* pose, fsdo_frame_result* out);
void fsdo_match(const double* left, int nl, const double* right, int nr, const double* pose, fsdo_frame_result* out);
// path from matched cones; returns status in out->status
void fsdo_path(const double* left_v, int nl, const double* right_v, int nr, const int32_t* l2r, const int32_t* r2l,
               const double* pose, fsdo_frame_result* out);
// the constant initial previous path (P11), (40,4)
void fsdo_default_path(double* out40x4);

// per-side config dump for stage tests: configs (C,12) sorted by cost, costs (C,)
int fsdo_side_configs(const double* cones_xyt, int n, const double* pose, int cone_type, int32_t* configs_out,
                      double* costs_out, int max_configs, int32_t* first_k_out /*2*/);

// math mode for the arc extension: 0 = libm (default; reference pinning), 1 = det_math.h (exact parity with the HIP kernels)
// fsdo_plan_frame + the smoothing splines the frame fitted (utils/spline_fit.py:117 splprep calls), in call order
enum { FSDO_FIT_KNOTS = 48, FSDO_FIT_STRIDE = 2 + 3 * 48 };
int fsdo_plan_frame_capture(const double* cones_xyt, int n, const double* pose, fsdo_frame_result* out, double* fits, int max_fits);
// configuration constants (config.py): 13 doubles [max_n_neighbors, max_length, max_dist, max_dist_to_first,
// threshold_directional_angle, threshold_absolute_angle, min_track_width, max_search_range, max_search_angle, smoothing,
// predict_every, maximal_distance_for_valid_path, mpc_path_length]; NULL = defaults.  Process-wide, set between batches.
void fsdo_set_params(const double* v13);
void fsdo_set_math_mode(int mode);
int fsdo_get_math_mode(void);

#ifdef __cplusplus
}
#endif
