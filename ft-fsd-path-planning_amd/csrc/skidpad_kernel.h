// Skidpad mission (BASELINE config 5) as wave-per-planner-instance HIP kernels (gfx950).
//
// Replaces, per planner instance and per frame, the relocalizer branch of
// PathPlanner.calculate_path_in_global_frame (reference full_pipeline/full_pipeline.py:118-194):
//   skid_reloc_kernel : Relocalizer.attempt_relocalization_calculation (relocalization_base_class.py:50-75) ->
//                       SkidpadRelocalizer.do_relocalization_once (skidpad_relocalizer.py:198-240):
//                       20 nearest cones (wave arg-min), all C(20,3) = 1140 circle fits one per lane,
//                       DBSCAN(eps=3, min_samples=1) as min-label propagation over the <= 3 m graph,
//                       per-cluster medians by rank counting, best centre pair, rigid transform.
//   skid_path_kernel  : pose into the map frame, SkidpadCalculatePath.fit_matches_as_spline
//                       (skidpad_calculate_path.py:49-71: stateful window arg-min on the known path) or the trivial
//                       path before relocalization (core_calculate_path.py:127-134), then the common MPC step
//                       (finish_path) with the instance's previous path, path back to the original frame
//                       (full_pipeline.py:178-194), state update.
// State lives in HBM per instance (SkidState); the known path table, the fixed-seed noise table and the
// reference centres are constant inputs uploaded once (fsdp_skidpad_set_tables).
#pragma once
#include "path_kernel.h"

namespace fsdp {

constexpr int SKID_MAX_CLUSTERS = 64;
constexpr int SKID_NEAR = 20;
constexpr int ST_OVERFLOW_CLUSTERS = 205;
constexpr int ST_SYNC_LOST = 206;  // internal: a step's wavefront never saw its predecessor's state published

constexpr int SKID_HIST = 32;  // > SKID_GROUP_MAX

struct SkidState {
  int32_t has_original, relocalized, index_along_path;
  int32_t reloc_step;             // the step (since the reset) whose relocalization attempt latched the transform
  int32_t index_hist[SKID_HIST];  // index_along_path after step g at [g % SKID_HIST]
  double orig[4];          // pose latched at the first relocalization attempt
  double translation[2];
  double right_calc[2];
  double rotation;
  double prev[PATH_POINTS][4];  // previous_paths[-1] (map frame once relocalized)
};

struct SkidTables {
  const double* path;    // known global path = BASE_SKIDPAD_PATH[::2], (n_path, 2)
  int n_path;
  const double* noise;   // RandomState(42).randn(n_noise / 6, 3, 2)
  int n_noise;
  double ref_right[2], ref_left[2];  // calculate_reference_centers_for_skidpad_path
  double mean_distance;              // mean of the first 9 segment lengths of the known path (NumPy mean)
  const Params* prm;                 // the context's configuration constants
};

struct SkidInfo {  // what RelocalizationInformation / the planner state expose (relocalization_information.py:12-35)
  int32_t relocalized, index_along_path;
  double translation[2];
  double rotation;
};

struct SkidShared {
  double nx[SKID_NEAR], ny[SKID_NEAR];
  double medx[SKID_MAX_CLUSTERS], medy[SKID_MAX_CLUSTERS];
  int32_t root[SKID_MAX_CLUSTERS];
  int32_t n_roots;
};

// calculate_reference_centers_for_skidpad_path (skidpad_relocalizer.py:172-183): hyper circle fits
// (utils/math_utils.py:579-646) of the table points with y < -2 (right circle) and y > 2 (left circle).  One-off per
// context, one wavefront: lanes compact the points and form the element-wise products, the np.sum reductions run in
// NumPy's pairwise order (uniform across the wave).  scratch: 3 * n_table doubles.  out4[0..5) = right xy, left xy, and
// the mean spacing of the known path.
__global__ void __launch_bounds__(64) skid_centers_kernel(const double* __restrict__ table, int n_table,
                                                          double* __restrict__ scratch, double* __restrict__ out4) {
  const int lane = lane_id();
  double* X = scratch;
  double* Y = scratch + n_table;
  double* P = scratch + 2 * (size_t)n_table;
  for (int side = 0; side < 2; side++) {
    int n = 0;
    for (int base = 0; base < n_table; base += WAVE) {
      const int i = base + lane;
      double x = 0, y = 0;
      bool keep = false;
      if (i < n_table) {
        x = table[2 * i];
        y = table[2 * i + 1];
        keep = side == 0 ? (y < -2) : (y > 2);
      }
      const unsigned long long m = __ballot(keep);
      if (keep) {
        const int p = n + __popcll(m & ((1ull << lane) - 1ull));
        X[p] = x;
        Y[p] = y;
      }
      n += __popcll(m);
    }
    __syncthreads();
    const double dn = (double)n;
    const double xm = np_sum_long(X, n) / dn, ym = np_sum_long(Y, n) / dn;
    // Mxy, Mxx, Myy, Mxz, Myz, Mzz: each product array is materialised, then np.sum'ed
    double M[6];
    for (int q = 0; q < 6; q++) {
      __syncthreads();
      for (int i = lane; i < n; i += WAVE) {
        const double a = X[i] - xm, b = Y[i] - ym;
        const double z = a * a + b * b;
        P[i] = q == 0 ? a * b : q == 1 ? a * a : q == 2 ? b * b : q == 3 ? a * z : q == 4 ? b * z : z * z;
      }
      __syncthreads();
      M[q] = np_sum_long(P, n) / dn;
    }
    const double Mxy = M[0], Mxx = M[1], Myy = M[2], Mxz = M[3], Myz = M[4], Mzz = M[5];
    const double Mz = Mxx + Myy;
    const double Cov_xy = Mxx * Myy - Mxy * Mxy;
    const double Var_z = Mzz - Mz * Mz;
    const double A2 = 4 * Cov_xy - 3 * Mz * Mz - Mzz;
    const double A1 = Var_z * Mz + 4.0 * Cov_xy * Mz - Mxz * Mxz - Myz * Myz;
    const double A0 = Mxz * (Mxz * Myy - Myz * Mxy) + Myz * (Myz * Mxx - Mxz * Mxy) - Var_z * Cov_xy;
    const double A22 = A2 + A2;
    double y = A0, x = 0.0;
    for (int it = 0; it < 99; it++) {
      const double Dy = A1 + x * (A22 + 16.0 * x * x);
      const double x_new = x - y / Dy;
      if (x_new == x || !isfinite(x_new)) break;
      const double y_new = A0 + x_new * (A1 + x_new * (A2 + 4.0 * x_new * x_new));
      if (fabs(y_new) >= fabs(y)) break;
      x = x_new;
      y = y_new;
    }
    const double det = x * x - x * Mz + Cov_xy;
    const double Xc = (Mxz * (Myy - x) - Myz * Mxy) / det / 2.0;
    const double Yc = (Myz * (Mxx - x) - Mxz * Mxy) / det / 2.0;
    if (lane == 0) {
      out4[2 * side] = Xc + xm;
      out4[2 * side + 1] = Yc + ym;
    }
    __syncthreads();
  }
  // skidpad_calculate_path.py:58: np.mean(np.linalg.norm(np.diff(path[:10], axis=-2), axis=-1)) of the known path
  // (= table[::2]); nine segment lengths, np.mean = pairwise sum / 9
  if (lane == 0) {
    double seg[9];
    for (int i = 0; i < 9; i++) {
      const double dx = table[2 * (2 * (i + 1))] - table[2 * (2 * i)], dy = table[2 * (2 * (i + 1)) + 1] - table[2 * (2 * i) + 1];
      seg[i] = sqrt(dx * dx + dy * dy);
    }
    out4[4] = np_sum_small(seg, 9) / 9.0;
  }
}

// unrank the s-th 3-subset of {0..m-1} in itertools.combinations order
__device__ inline void unrank3(int s, int m, int& a, int& b, int& c) {
  a = 0;
  for (;;) {
    int rest = m - 1 - a;
    int cnt = rest * (rest - 1) / 2;
    if (s < cnt) break;
    s -= cnt;
    a++;
  }
  b = a + 1;
  for (;;) {
    int cnt = m - 1 - b;
    if (s < cnt) break;
    s -= cnt;
    b++;
  }
  c = b + 1 + s;
}

__global__ void __launch_bounds__(64) skid_reloc_kernel(int n_inst, const int32_t* __restrict__ cone_offsets,
                                                        const double* __restrict__ cones_xyt, const double* __restrict__ poses,
                                                        SkidState* __restrict__ states, SkidTables T, double* __restrict__ arena,
                                                        int32_t* __restrict__ status_out, int step_no) {
  __shared__ SkidShared S;
  const int inst = blockIdx.x;
  if (inst >= n_inst) return;
  const int lane = lane_id();
  SkidState* st = &states[inst];
  if (lane == 0) status_out[inst] = ST_OK;
  const int already = st->relocalized;
  if (already) return;
  const double px = poses[4 * inst + 0], py = poses[4 * inst + 1], dx = poses[4 * inst + 2], dy = poses[4 * inst + 3];
  // latch the pose of the first attempt (relocalization_base_class.py:66-68)
  const int had_original = st->has_original;
  const double opx = had_original ? st->orig[0] : px, opy = had_original ? st->orig[1] : py;
  const double odx = had_original ? st->orig[2] : dx, ody = had_original ? st->orig[3] : dy;
  __syncthreads();
  if (!had_original && lane == 0) {
    st->has_original = 1;
    st->orig[0] = px;
    st->orig[1] = py;
    st->orig[2] = dx;
    st->orig[3] = dy;
  }
  const int off = cone_offsets[inst];
  const int n = cone_offsets[inst + 1] - off;
  const double* cones = cones_xyt + 3 * (size_t)off;
  const Arena A = frame_arena(arena, inst, T.prm);
  // ---- 20 closest cones, in argsort order (stable) ----
  const int m = n < SKID_NEAR ? n : SKID_NEAR;
  {
    // distances into the arena (n may exceed 64); repeated wave arg-min with the taken ones masked by +inf
    for (int i = lane; i < n; i += WAVE) A.u[i] = norm_axis(cones[3 * i] - px, cones[3 * i + 1] - py);
    __syncthreads();
    for (int k = 0; k < m; k++) {
      double bv = 0.0;
      int bi = -1;
      for (int i = lane; i < n; i += WAVE) {
        double d = A.u[i];
        if (d >= 0 && (bi < 0 || d < bv)) {  // taken entries are marked negative
          bv = d;
          bi = i;
        }
      }
      wave_argmin(bv, bi);
      __syncthreads();
      if (lane == 0) {
        S.nx[k] = cones[3 * bi];
        S.ny[k] = cones[3 * bi + 1];
        A.u[bi] = -1.0;
      }
      __syncthreads();
    }
  }
  // ---- circle_fit_powerset: every 3-subset, one per lane, accepted centres compacted in subset order ----
  const int n_sub = (m >= 3) ? m * (m - 1) * (m - 2) / 6 : 0;
  int n_centers = 0;
  bool noise_short = false;
  for (int base = 0; base < n_sub; base += WAVE) {
    int s = base + lane;
    bool acc = false;
    double cx = 0, cy = 0;
    if (s < n_sub) {
      if ((s + 1) * 6 > T.n_noise) {
        noise_short = true;
      } else {
        int ia, ib, ic;
        unrank3(s, m, ia, ib, ic);
        int id[3] = {ia, ib, ic};
        double qx[3], qy[3];
        double mind[3];
#pragma unroll
        for (int col = 0; col < 3; col++) {
          double best = 0.0;
          bool first = true;
#pragma unroll
          for (int row = 0; row < 3; row++) {
            double d = (row == col) ? INFINITY : sqrt(cdist_sq(S.nx[id[row]], S.ny[id[row]], S.nx[id[col]], S.ny[id[col]]));
            if (first) {
              best = d;
              first = false;
            } else if (isnan(d) || isnan(best)) {
              best = NAN;
            } else if (d < best) {
              best = d;
            }
          }
          mind[col] = best;
        }
        double mean_distance = (0.0 + ((mind[0] + mind[1]) + mind[2])) / 3.0;
#pragma unroll
        for (int r = 0; r < 3; r++) {
          qx[r] = S.nx[id[r]] + T.noise[(size_t)s * 6 + 2 * r] * 1e-3;
          qy[r] = S.ny[id[r]] + T.noise[(size_t)s * 6 + 2 * r + 1] * 1e-3;
        }
        double rad;
        circle_fit(qx, qy, 0, 3, cx, cy, rad);
        double res[3];
#pragma unroll
        for (int r = 0; r < 3; r++) res[r] = fabs(norm_axis(cx - qx[r], cy - qy[r]) - rad);
        double residual = (0.0 + ((res[0] + res[1]) + res[2])) / 3.0;
        acc = fabs(rad - 7.625) < 1.0 && fabs(mean_distance - 2.4) < 1.5 && residual < 0.4;
      }
    }
    unsigned long long am = __ballot(acc);
    if (acc) {
      int p = n_centers + __popcll(am & ((1ull << lane) - 1ull));
      A.x[p] = cx;
      A.y[p] = cy;
    }
    n_centers += __popcll(am);
  }
  if (__ballot(noise_short) != 0ull) {
    if (lane == 0) status_out[inst] = ST_REF_UNDEFINED_PATH;
    return;
  }
  __syncthreads();
  if (n_centers < 3) return;
  // ---- DBSCAN(eps = 3, min_samples = 1): connected components by min-label propagation; label = smallest member index ----
  const int C = n_centers;
  int32_t* label = (int32_t*)A.u;  // the distance scratch is dead
  for (int i = lane; i < C; i += WAVE) label[i] = i;
  __syncthreads();
  for (int iter = 0; iter < C; iter++) {
    bool changed = false;
    for (int base = 0; base < C; base += WAVE) {
      int i = base + lane;
      int nl = 0;
      if (i < C) {
        nl = label[i];
        double xi = A.x[i], yi = A.y[i];
        for (int j = 0; j < C; j++) {
          double ddx = xi - A.x[j], ddy = yi - A.y[j];
          if (sqrt(ddx * ddx + ddy * ddy) <= 3.0) {
            int lj = label[j];
            nl = lj < nl ? lj : nl;
          }
        }
      }
      __syncthreads();
      if (i < C && nl != label[i]) {
        label[i] = nl;
        changed = true;
      }
      __syncthreads();
    }
    if (__ballot(changed) == 0ull) break;
  }
  // roots in ascending order = np.unique(labels)
  int n_roots = 0;
  bool too_many = false;
  for (int base = 0; base < C; base += WAVE) {
    int i = base + lane;
    bool is_root = i < C && label[i] == i;
    unsigned long long rm = __ballot(is_root);
    if (is_root) {
      int p = n_roots + __popcll(rm & ((1ull << lane) - 1ull));
      if (p < SKID_MAX_CLUSTERS) S.root[p] = i;
    }
    n_roots += __popcll(rm);
  }
  if (n_roots > SKID_MAX_CLUSTERS) too_many = true;
  __syncthreads();
  if (too_many) {
    if (lane == 0) status_out[inst] = ST_OVERFLOW_CLUSTERS;
    return;
  }
  if (!(n_roots > 1)) return;  // AssertionError in the reference -> attempt fails
  // per-cluster medians (np.median per axis): rank counting, lanes = members
  for (int r = 0; r < n_roots; r++) {
    const int root = S.root[r];
    int k = 0;
    for (int base = 0; base < C; base += WAVE) {
      int i = base + lane;
      k += __popcll(__ballot(i < C && label[i] == root));
    }
    const int lo_rank = (k % 2 == 1) ? k / 2 : k / 2 - 1, hi_rank = k / 2;
    double lox = 0, hix = 0, loy = 0, hiy = 0;
    for (int base = 0; base < C; base += WAVE) {
      int i = base + lane;
      bool mem = i < C && label[i] == root;
      int rx = 0, ry = 0;
      double xi = 0, yi = 0;
      if (mem) {
        xi = A.x[i];
        yi = A.y[i];
        for (int j = 0; j < C; j++) {
          if (label[j] != root) continue;
          double xj = A.x[j], yj = A.y[j];
          rx += (xj < xi) || (xj == xi && j < i);
          ry += (yj < yi) || (yj == yi && j < i);
        }
      }
      // exactly one member holds each rank: broadcast through ballots
      unsigned long long b;
      b = __ballot(mem && rx == lo_rank);
      if (b) lox = __shfl(xi, __ffsll(b) - 1, WAVE);
      b = __ballot(mem && rx == hi_rank);
      if (b) hix = __shfl(xi, __ffsll(b) - 1, WAVE);
      b = __ballot(mem && ry == lo_rank);
      if (b) loy = __shfl(yi, __ffsll(b) - 1, WAVE);
      b = __ballot(mem && ry == hi_rank);
      if (b) hiy = __shfl(yi, __ffsll(b) - 1, WAVE);
    }
    __syncthreads();
    if (lane == 0) {
      S.medx[r] = (k % 2 == 1) ? hix : (0.0 + (lox + hix)) / 2.0;
      S.medy[r] = (k % 2 == 1) ? hiy : (0.0 + (loy + hiy)) / 2.0;
    }
    __syncthreads();
  }
  // best centre pair: |18.25 - distance| minimal, first pair (label order) on ties
  double best_distance = 1000.0;
  int best_pair = -1;
  {
    const int n_pairs = n_roots * (n_roots - 1) / 2;
    double bv = 0.0;
    int bi = -1;
    for (int pidx = lane; pidx < n_pairs; pidx += WAVE) {
      // unrank pair in combinations order
      int a = 0, s = pidx;
      while (s >= n_roots - 1 - a) {
        s -= n_roots - 1 - a;
        a++;
      }
      int b = a + 1 + s;
      double d = fabs(18.25 - norm_blas(S.medx[a] - S.medx[b], S.medy[a] - S.medy[b]));
      if (bi < 0 || d < bv) {
        bv = d;
        bi = pidx;
      }
    }
    wave_argmin(bv, bi);
    if (bi >= 0 && bv < best_distance) {
      best_distance = bv;
      best_pair = bi;
    }
  }
  if (best_pair < 0 || best_distance > 0.5) return;  // ValueError in the reference -> attempt fails
  int ca = 0, cs = best_pair;
  while (cs >= n_roots - 1 - ca) {
    cs -= n_roots - 1 - ca;
    ca++;
  }
  const int cb = ca + 1 + cs;
  // ---- calculate_transformation (skidpad_relocalizer.py:101-169), wave-uniform scalars; det_math for libm values ----
  const double c0x = S.medx[ca], c0y = S.medy[ca], c1x = S.medx[cb], c1y = S.medy[cb];
  const double yaw0 = detm::det_atan2(ody, odx);
  double sn, cs_;
  detm::det_sincos(-yaw0, sn, cs_);
  const double v0y = blas_dot2(c0x - opx, sn, c0y - opy, cs_);
  const double v1y = blas_dot2(c1x - opx, sn, c1y - opy, cs_);
  const bool r0 = v0y < 0.0, r1 = v1y < 0.0;
  if (r0 == r1) return;  // IndexError in the reference -> attempt fails
  const double rcx = r0 ? c0x : c1x, rcy = r0 ? c0y : c1y;
  const double lcx = r0 ? c1x : c0x, lcy = r0 ? c1y : c0y;
  const double reference_angle = detm::det_atan2(T.ref_left[1] - T.ref_right[1], T.ref_left[0] - T.ref_right[0]);
  const double calculated_angle = detm::det_atan2(lcy - rcy, lcx - rcx);
  __syncthreads();
  if (lane == 0) {
    st->translation[0] = T.ref_right[0] - rcx;
    st->translation[1] = T.ref_right[1] - rcy;
    st->rotation = reference_angle - calculated_angle;
    st->right_calc[0] = rcx;
    st->right_calc[1] = rcy;
    st->relocalized = 1;
    st->reloc_step = step_no;
  }
}

// ---- the path step of a planner: pieces shared by the kernels below -------------------------------------------------------
constexpr int SKID_GROUP_MAX = 16;  // steps of one planner that may share a launch
constexpr int FB_READ_PREVIOUS = 1 | 2 | 4 | 8;  // path_fallback bits whose branch reads previous_paths[-1] (16, 32: how the path was extended)
constexpr int ST_SERIAL = 298;      // internal (PathMid::status): the step is planned by its planner's own wavefront, behind its predecessor

struct SkidStep {
  const double* poses;       // (n_inst, 4) of this step
  const int32_t* status_in;  // skid_reloc_kernel's status of this step; NULL: no attempt was due (every planner relocalized): ST_OK
  double* arena;             // workspace of this step's wavefronts
  PathOut* out;
  SkidInfo* info;
};

struct SkidGroup {
  SkidStep step[SKID_GROUP_MAX];
  int32_t n_steps;
  int32_t step0;          // number of the group's first step since fsdp_skidpad_reset
  uint32_t ticket_base;   // value of the ticket counter before this launch
};

// full_pipeline.py:126-134: pose into the known map frame (the transform is read where it is used rather than held in
// registers across the path stage)
template <bool UNIFORM = true>
__device__ __forceinline__ void skid_map_pose(const SkidState* st, const SkidTables& T, const double* pose, double& px, double& py,
                                              double& dx, double& dy) {
  px = pose[0], py = pose[1], dx = pose[2], dy = pose[3];
  const double rotation = st->rotation;
  double yaw = detm::det_atan2(dy, dx);
  double sn, cs;
  detm::det_sincos(rotation, sn, cs);
  double qx = px + st->translation[0] - T.ref_right[0], qy = py + st->translation[1] - T.ref_right[1];
  double rx = blas_dot2_single_row(qx, cs, qy, -sn), ry = blas_dot2_single_row(qx, sn, qy, cs);  // (one point: gemv order)
  px = rx + T.ref_right[0];
  py = ry + T.ref_right[1];
  yaw = yaw + rotation;
  detm::det_sincos(yaw, dy, dx);
  if (UNIFORM) px = wave_uniform(px), py = wave_uniform(py), dx = wave_uniform(dx), dy = wave_uniform(dy);
}

// SkidpadCalculatePath.fit_matches_as_spline, skidpad_calculate_path.py:60-67: closest point of the known path within
// +-max_change of the index, over the lanes of a wavefront; -1: empty window
__device__ __forceinline__ int skid_closest_in_window(const SkidTables& T, int index, double px, double py) {
  const int lane = lane_id();
  const int max_change = (int)(20 / T.mean_distance);
  int lo = index - max_change;
  lo = lo < 0 ? 0 : lo;
  int hi = index + max_change;
  hi = hi > T.n_path ? T.n_path : hi;
  if (hi <= lo) return -1;
  double bv = 0.0;
  int bi = -1;
  for (int i = lo + lane; i < hi; i += WAVE) {
    double d = norm_axis(px - T.path[2 * i], py - T.path[2 * i + 1]);
    d = (d != d) ? -1.0 : d;  // np.argmin: the first NaN is the minimum (and every lane agrees on it); distances are >= 0
    if (bi < 0 || d < bv) {
      bv = d;
      bi = i;
    }
  }
  wave_argmin(bv, bi);
  return wave_uniform(bi);
}

// the path update of a step into the arena polyline [1, 1 + n1), over the G lanes of a frame: the window of the known
// path behind its closest point (skidpad_calculate_path.py:68-71) or, before relocalization, the trivial path
// (calculate_trivial_path, core_calculate_path.py:127-134: chord[1:] rotated by the car yaw, + position); returns n1
template <int G>
__device__ __forceinline__ int skid_fill_update(const Arena& A, const SkidTables& T, const double* __restrict__ chord, bool reloc,
                                                int first, double px, double py, double dx, double dy) {
  using GR = Grp<G>;
  const int lane = GR::lane();
  int n1;
  if (reloc) {
    int fin = first + (int)(25 / T.mean_distance);
    fin = fin > T.n_path ? T.n_path : fin;
    n1 = fin - first;
    for (int i = lane; i < n1; i += G) {
      A.x[1 + i] = T.path[2 * (first + i)];
      A.y[1 + i] = T.path[2 * (first + i) + 1];
    }
  } else {
    double yaw = detm::det_atan2(dy, dx);
    double sn, cs;
    detm::det_sincos(yaw, sn, cs);
    n1 = CHORD_POINTS - 1;
    for (int i = lane; i < n1; i += G) {
      double cxp = chord[2 * (i + 1)], cyp = chord[2 * (i + 1) + 1];
      A.x[1 + i] = blas_dot2(cxp, cs, cyp, -sn) + px;
      A.y[1 + i] = blas_dot2(cxp, sn, cyp, cs) + py;
    }
  }
  GR::sync();
  return n1;
}

// One step of one planner on one wavefront, from the window index index_in and the previous path prev: window lookup, the
// common MPC step (finish_path) with the shortened division / square-root sequences (spline_device.h: the same bits for
// operands inside their exponent band) and, for a step that meets an operand outside the band (ST_RETRY), once more with
// the plain ones.  Returns the step's status; *new_index = the index the step leaves (the reference moves it before the
// MPC step, skidpad_calculate_path.py:66-67: it stays moved when the step raises).
template <class PS>
__device__ __forceinline__ int skid_plan_step(PS& S, const Arena& A, const SkidTables& T, const double* __restrict__ chord, int status_in,
                                              bool reloc, int index_in, double px, double py, double dx, double dy, const double* prev,
                                              double (*out)[4], int* fallback, int* n_dense, int* new_index) {
  int status = status_in, n1 = 0;
  *fallback = *n_dense = 0;
  *new_index = index_in;
  if (status != ST_OK) return status;
  int first = 0;
  if (reloc) {
    first = skid_closest_in_window(T, index_in, px, py);
    if (first < 0) return ST_REF_UNDEFINED_PATH;
    *new_index = first;
  }
  // a car position that is not finite: the window index has moved (to the window's first point: np.argmin of NaNs), the
  // MPC step raises whatever path it is given (the car position joins the path) — the reference's exception reaches the
  // caller
  if (!(fabs(px) < INFINITY && fabs(py) < INFINITY)) return ST_REF_UNDEFINED_PATH;
  n1 = skid_fill_update<WAVE>(A, T, chord, reloc, first, px, py, dx, dy);
  status = finish_path<WAVE, true>(S, A, n1, px, py, dx, dy, prev, out, fallback, n_dense);
  if (status == ST_RETRY || status == ST_OVERFLOW_KNOTS) {
    __syncthreads();
    *fallback = *n_dense = 0;
    n1 = skid_fill_update<WAVE>(A, T, chord, reloc, first, px, py, dx, dy);
    status = finish_path<WAVE, false>(S, A, n1, px, py, dx, dy, prev, out, fallback, n_dense);
  }
  __syncthreads();
  return status;
}

// What a finished step leaves behind: previous_paths[-1] <- the result (map frame), the path xy back in the original
// frame (full_pipeline.py:178-194), the window index, the records of the step.  THROUGH: the state goes out with
// agent-scope stores (written through to where another wavefront, on whatever XCD, reads it inside the same launch).
template <bool THROUGH>
__device__ __forceinline__ void skid_finish_step(SkidState* st, const SkidTables& T, PathOut* o, SkidInfo* fi, bool reloc, int status,
                                                 int fallback, int n_dense, int index_out, int g) {
  const int lane = lane_id();
  if (status == ST_OK) {
    if (lane < PATH_POINTS) {
      double u = o->path[lane][0], x = o->path[lane][1], y = o->path[lane][2], k = o->path[lane][3];
      if (THROUGH) {
        __hip_atomic_store(&st->prev[lane][0], u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&st->prev[lane][1], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&st->prev[lane][2], y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&st->prev[lane][3], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        st->prev[lane][0] = u;
        st->prev[lane][1] = x;
        st->prev[lane][2] = y;
        st->prev[lane][3] = k;
      }
      if (reloc) {
        const double rcx = st->right_calc[0], rcy = st->right_calc[1];
        double sn, cs;
        detm::det_sincos(-st->rotation, sn, cs);
        double qx = x - st->translation[0] - rcx, qy = y - st->translation[1] - rcy;
        o->path[lane][1] = blas_dot2(qx, cs, qy, -sn) + rcx;
        o->path[lane][2] = blas_dot2(qx, sn, qy, cs) + rcy;
      }
    }
  } else if (lane < PATH_POINTS) {
    for (int q = 0; q < 4; q++) o->path[lane][q] = NAN;
  }
  if (lane == 0) {
    if (THROUGH)
      __hip_atomic_store(&st->index_along_path, index_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      st->index_along_path = index_out;
    st->index_hist[g % SKID_HIST] = index_out;
    o->status = status;
    o->fallback = fallback;
    o->n_dense = n_dense;
    o->pad = 0;
    fi->relocalized = reloc ? 1 : 0;
    fi->index_along_path = index_out;
    fi->translation[0] = fi->translation[1] = fi->rotation = NAN;
    if (reloc) {
      // RelocalizationInformation.from_transform_function: images of (0,0) and (1,0)
      const double tx = st->translation[0], ty = st->translation[1];
      double sn, cs;
      detm::det_sincos(st->rotation, sn, cs);
      double ax0 = 0.0 + tx - T.ref_right[0], ay0 = 0.0 + ty - T.ref_right[1];
      double ax1 = 1.0 + tx - T.ref_right[0];
      double o0x = blas_dot2_single_row(ax0, cs, ay0, -sn) + T.ref_right[0], o0y = blas_dot2_single_row(ax0, sn, ay0, cs) + T.ref_right[1];
      double o1x = blas_dot2_single_row(ax1, cs, ay0, -sn) + T.ref_right[0], o1y = blas_dot2_single_row(ax1, sn, ay0, cs) + T.ref_right[1];
      fi->translation[0] = o0x;
      fi->translation[1] = o0y;
      fi->rotation = atan2(o1y - o0y, o1x - o0x);
    }
  }
}

// ---- steps in flight, a wavefront per (instance, step) --------------------------------------------------------------------
// A planner's steps chain through its state (window index, previous path), but almost all of a step — the MPC refit of the
// window — depends on that state only through the window index, and the index a step leaves depends on the poses alone
// (one 400-point arg-min per step).  So one launch plans up to SKID_GROUP_MAX consecutive steps of every instance, one
// wavefront per (instance, step): the wavefront of step g
//   1. works out the index step g - 1 leaves from the index recorded before the launch (index_hist) and the poses of the
//      launch's earlier steps,
//   2. plans its step with it,
//   3. waits until step g - 1 of its instance has published its state (sync[1 + inst] >= g), and
//   4. keeps its result if the previous path was never read (every read of it sets a fallback bit) and the published
//      index is the one it used — otherwise plans the step again from the published state, exactly as a launch of its
//      own would;
//   5. writes the state and publishes it (sync[1 + inst] = g + 1).
// Results and states are therefore those of one launch per step, whatever the grouping.  Wavefronts take their
// (step, instance) from a ticket counter in start order, step-major, so the wavefront a wait is for has always started:
// no wait can be for work that is not resident or finished.  With 1024 instances three steps put three wavefronts on every
// SIMD, where one alone issues an FP64 instruction every 8.7 cycles and four together one every 4.6; the
// kernel is held to 168 registers for that (144 spilled: 1 % slower alone, 8 % faster three to a SIMD than two at 256).
#ifndef FSDP_SKID_PATH_WAVES
#define FSDP_SKID_PATH_WAVES 2  // (3: 168 registers, 214 of them spilled, 384 B of scratch per lane; 2: 256 registers, 36 spilled, 112 B — same frames/s, profiles/r06_kernel_resources.txt)
#endif
__global__ void __launch_bounds__(64, FSDP_SKID_PATH_WAVES) skid_path_kernel(int n_inst, SkidGroup G, SkidState* states, SkidTables T,
                                                          const double* __restrict__ chord, uint32_t* sync) {
  __shared__ PathShared<WAVE> S;
  __shared__ uint32_t s_ticket, s_spins;
  __shared__ double s_prev[PATH_POINTS][4];
  const int lane = lane_id();
  if (lane == 0) s_ticket = atomicAdd(&sync[0], 1u) - G.ticket_base;
  __syncthreads();
  const uint32_t ticket = s_ticket;
  if (ticket >= (uint32_t)n_inst * (uint32_t)G.n_steps) return;
  const int s = (int)(ticket / (uint32_t)n_inst), inst = (int)(ticket % (uint32_t)n_inst);
  const int g = G.step0 + s;
  const SkidStep& me = G.step[s];
  const Arena A = frame_arena(me.arena, inst, T.prm);
  SkidState* st = &states[inst];
  PathOut* o = &me.out[inst];
  // written by skid_reloc_kernel only, i.e. before this launch
  const bool latched = wave_uniform(st->relocalized) != 0;
  const int reloc_step = wave_uniform(st->reloc_step);

  // 1. the index step g - 1 leaves
  int index_in = wave_uniform(st->index_hist[(G.step0 + SKID_HIST - 1) % SKID_HIST]);
  for (int j = 0; j < s; j++) {
    if ((G.step[j].status_in && G.step[j].status_in[inst] != ST_OK) || !(latched && reloc_step <= G.step0 + j)) continue;
    double qx, qy, qdx, qdy;
    skid_map_pose(st, T, G.step[j].poses + 4 * inst, qx, qy, qdx, qdy);
    const int bi = skid_closest_in_window(T, index_in, qx, qy);
    if (bi >= 0) index_in = bi;
  }

  const bool reloc = latched && reloc_step <= g;
  double px = wave_uniform(me.poses[4 * inst + 0]), py = wave_uniform(me.poses[4 * inst + 1]);
  double dx = wave_uniform(me.poses[4 * inst + 2]), dy = wave_uniform(me.poses[4 * inst + 3]);
  const int status_in = me.status_in ? wave_uniform(me.status_in[inst]) : ST_OK;
  if (status_in == ST_OK && reloc) skid_map_pose(st, T, me.poses + 4 * inst, px, py, dx, dy);
  int status = ST_OK, fallback = 0, n_dense = 0, new_index = index_in;
  bool sync_lost = false;
  const double* prev = &st->prev[0][0];  // first round: whatever is there (a result that read it is not kept)
#pragma unroll 1
  for (int round = 0; round < 2; round++) {
    // 2. (and 4.) the step from index_in
    status = skid_plan_step(S, A, T, chord, status_in, reloc, index_in, px, py, dx, dy, prev, o->path, &fallback, &n_dense, &new_index);
    if (round == 1) break;
    // 3. step g - 1 of this instance has published its state
    // (bounded, seconds: a wait that ends this way means a broken launch order — the step reports ST_SYNC_LOST instead of
    // holding the GPU)
    if (lane == 0) {
      uint32_t spins = 0;
      while (__hip_atomic_load(&sync[1 + inst], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)g && ++spins < (1u << 24))
        __builtin_amdgcn_s_sleep(8);
      s_spins = spins;
    }
    __syncthreads();
    if (s_spins >= (1u << 24)) {
      sync_lost = true;
      break;
    }
    // the state behind the flag is read past the caches (agent-scope loads): no cache invalidation, the workspace of the
    // path stage stays where it is
    const int index_now = wave_uniform(__hip_atomic_load(&st->index_along_path, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (s == 0 || (index_now == index_in && (fallback & FB_READ_PREVIOUS) == 0)) break;  // (s == 0: the state was final before the launch)
    index_in = index_now;
    if (lane < PATH_POINTS)
      for (int q = 0; q < 4; q++) s_prev[lane][q] = __hip_atomic_load(&st->prev[lane][q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    prev = &s_prev[0][0];
    __syncthreads();
  }
  __syncthreads();  // every lane has read the published index before lane 0 replaces it
  if (sync_lost) {
    if (lane < PATH_POINTS)
      for (int q = 0; q < 4; q++) o->path[lane][q] = NAN;
    if (lane == 0) {
      o->status = ST_SYNC_LOST;
      o->fallback = o->n_dense = o->pad = 0;
      __hip_atomic_store(&sync[1 + inst], (uint32_t)(g + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (later steps do not wait in turn)
    }
    return;
  }
  // 5. state and results
  skid_finish_step<true>(st, T, o, &me.info[inst], reloc, status, fallback, n_dense, new_index, g);
  __syncthreads();
  // the state above has left the wavefront (its stores are acknowledged) before the flag goes out; no cache write-back:
  // nothing else this wavefront wrote is anybody's before the launch ends
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);
  if (lane == 0) __hip_atomic_store(&sync[1 + inst], (uint32_t)(g + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- steps in flight, many frames per wavefront ----------------------------------------------------------------------------
// With thousands of (instance, step) pairs to plan — a replay submitted far ahead — the steps go through the packed
// kernels of the autocross path stage instead (path_kernel.h: fit_kernel / path_finish_kernel, 4-16 lanes per frame, what
// makes a frame 3-4 x cheaper than on a wavefront of its own), a frame being one step of one planner:
//   skid_select_kernel  one wavefront per planner: the poses of the group's steps in the map frame and the window indices
//                       they lead to, one after the other (they depend on the poses alone),
//   skid_prep_kernel    per frame: window / trivial path into the workspace, too-far check, connect / extend / trim,
//                       chord-length parameter (what path_prep_kernel does behind its own front end),
//   fit_kernel, path_finish_kernel  as they are,
//   skid_commit_kernel  one wavefront per planner, its steps in order: the packed result stands unless the step needs the
//                       previous path (too far from the car, the ValueError retry), left the packed kernels' envelope
//                       (knots, exponent bands) or failed there — such a step is planned here, by skid_plan_step with the
//                       planner's state as the steps before it left it, exactly as skid_path_kernel would; then the
//                       state moves on.  Results do not depend on the route a step took (the packed and the
//                       one-wavefront forms of the path stage give the same bits: tests/test_gpu_parity.py).
struct SkidSel {
  int32_t status, reloc, index_in, first;  // first: the index the step leaves (a relocalized step's window starts there)
  double px, py, dx, dy;                   // pose in the map frame
};

__global__ void __launch_bounds__(64) skid_select_kernel(int n_inst, SkidGroup G, const SkidState* __restrict__ states, SkidTables T,
                                                         SkidSel* __restrict__ sel) {
  const int inst = blockIdx.x;
  if (inst >= n_inst) return;
  const int lane = lane_id();
  const SkidState* st = &states[inst];
  const bool latched = wave_uniform(st->relocalized) != 0;
  const int reloc_step = wave_uniform(st->reloc_step);
  // lane s: the pose of step s in the map frame (the steps' poses do not depend on each other) ...
  SkidSel mine;
  mine.status = ST_OK, mine.reloc = 0, mine.index_in = mine.first = 0;
  mine.px = mine.py = mine.dx = mine.dy = 0.0;
  if (lane < G.n_steps) {
    const SkidStep& me = G.step[lane];
    mine.status = me.status_in ? me.status_in[inst] : ST_OK;
    mine.reloc = (latched && reloc_step <= G.step0 + lane) ? 1 : 0;
    mine.px = me.poses[4 * inst + 0], mine.py = me.poses[4 * inst + 1], mine.dx = me.poses[4 * inst + 2], mine.dy = me.poses[4 * inst + 3];
    if (mine.status == ST_OK && mine.reloc) skid_map_pose<false>(st, T, me.poses + 4 * inst, mine.px, mine.py, mine.dx, mine.dy);
  }
  // ... then the window index walks through the steps, one arg-min over the wavefront per step
  int index = wave_uniform(st->index_along_path);
  for (int s = 0; s < G.n_steps; s++) {
    int status = wave_uniform(__shfl(mine.status, s, WAVE));
    const bool reloc = wave_uniform(__shfl(mine.reloc, s, WAVE)) != 0;
    const double px = wave_uniform(__shfl(mine.px, s, WAVE)), py = wave_uniform(__shfl(mine.py, s, WAVE));
    const int index_in = index;
    if (status == ST_OK && reloc) {
      const int first = skid_closest_in_window(T, index, px, py);
      if (first < 0)
        status = ST_REF_UNDEFINED_PATH;
      else
        index = first;
    }
    if (status == ST_OK && !(fabs(px) < INFINITY && fabs(py) < INFINITY)) status = ST_REF_UNDEFINED_PATH;
    if (lane == s) {
      mine.status = status;
      mine.index_in = index_in;
      mine.first = index;
    }
  }
  if (lane < G.n_steps) sel[(size_t)lane * n_inst + inst] = mine;
}

template <int G>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) skid_prep_kernel(int n_frames, const SkidSel* __restrict__ sel, SkidTables T,
                                                       const double* __restrict__ chord, const double* __restrict__ default_path,
                                                       double* __restrict__ arena, PathMid* __restrict__ mid) {
  using GR = Grp<G>;
  __shared__ PathShared<G, true> S_all[WAVE / G];
  const int frame = blockIdx.x * (WAVE / G) + GR::index();
  if (frame < n_frames) {
    PathShared<G, true>& S = S_all[GR::index()];
    const Arena A = frame_arena(arena, frame, T.prm);
    const SkidSel sl = sel[frame];
    int fallback = 0, off = 0, n = 0;
    bool plain = false;
    if (sl.status == ST_OK) {
      int n1 = skid_fill_update<G>(A, T, chord, sl.reloc != 0, sl.first, sl.px, sl.py, sl.dx, sl.dy);
      // (a path too far from the car is replaced by the previous one, which only the planner's own wavefront knows: the
      // constant initial path stands in here, and the fallback bit sends the step there)
      n1 = overwrite_if_too_far<G>(A, n1, sl.px, sl.py, default_path, &fallback);
      if ((fallback & FB_READ_PREVIOUS) == 0) {
        const int rc = mpc_prepare<G>(S, A, n1, sl.px, sl.py, sl.dx, sl.dy, &fallback, &off, &n);
        plain = rc == 0 && n >= 4 && (fallback & FB_READ_PREVIOUS) == 0;  // degree 3 needs 4 points
        if (plain) build_parameter<G>(S, A, off, n);
      }
    }
    if (GR::lane() == 0) {
      PathMid m;
      m.status = plain ? ST_OK : (sl.status != ST_OK ? sl.status : ST_SERIAL);
      m.fallback = fallback;
      m.off = off;
      m.n = n;
      mid[frame] = m;
    }
  }
}

__global__ void __launch_bounds__(64, 1) skid_commit_kernel(int n_inst, SkidGroup G, SkidState* states, SkidTables T,
                                                            const double* __restrict__ chord, const SkidSel* __restrict__ sel,
                                                            const PathMid* __restrict__ mid, const PathOut* __restrict__ packed,
                                                            double* __restrict__ arena, uint32_t* __restrict__ sync) {
  __shared__ PathShared<WAVE> S;
  const int inst = blockIdx.x;
  if (inst >= n_inst) return;
  const int lane = lane_id();
  SkidState* st = &states[inst];
#pragma unroll 1
  for (int s = 0; s < G.n_steps; s++) {
    const size_t frame = (size_t)s * n_inst + inst;
    const SkidStep& me = G.step[s];
    PathOut* o = &me.out[inst];
    const int packed_status = wave_uniform(mid[frame].status);
    const int sel_status = wave_uniform(sel[frame].status);
    const bool reloc = wave_uniform(sel[frame].reloc) != 0;
    int status, fallback = 0, n_dense = 0, new_index = wave_uniform(sel[frame].first);
    if (packed_status == ST_OK) {
      // the packed kernels' result stands
      const PathOut* r = &packed[frame];
      if (lane < PATH_POINTS)
        for (int q = 0; q < 4; q++) o->path[lane][q] = r->path[lane][q];
      status = ST_OK;
      fallback = wave_uniform(r->fallback);
      n_dense = wave_uniform(r->n_dense);
      __syncthreads();
    } else if (sel_status != ST_OK) {
      status = sel_status;  // decided by the relocalization attempt, the window lookup or the car position
    } else {
      const double px = wave_uniform(sel[frame].px), py = wave_uniform(sel[frame].py);
      const double dx = wave_uniform(sel[frame].dx), dy = wave_uniform(sel[frame].dy);
      const Arena A = frame_arena(arena, (int)frame, T.prm);
      status = skid_plan_step(S, A, T, chord, ST_OK, reloc, wave_uniform(sel[frame].index_in), px, py, dx, dy, &st->prev[0][0], o->path,
                              &fallback, &n_dense, &new_index);
    }
    skid_finish_step<false>(st, T, o, &me.info[inst], reloc, status, fallback, n_dense, new_index, G.step0 + s);
    __syncthreads();
  }
  if (lane == 0) sync[1 + inst] = (uint32_t)(G.step0 + G.n_steps);  // steps published, for a later skid_path_kernel launch
}

}  // namespace fsdp
