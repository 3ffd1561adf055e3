// sorting_cones step as a hand-written wave-per-frame HIP kernel (gfx950).
//
// Replaces, for a batch of independent frames, the reference call chain
//   TraceSorter.sort_left_right                     sorting_cones/trace_sorter/core_trace_sorter.py:148-216
//     select_first_k_starting_cones                 :409-465   (lanes = cones, wave arg-min)
//     create_adjacency_matrix                       adjacency_matrix.py:60-128 (lane = row, register top-5,
//                                                   mutual-kNN through LDS, ballot-frontier BFS)
//     _impl_find_all_end_configurations             end_configurations.py:320-431 (explicit LDS stack, lanes =
//                                                   candidate neighbours, ballot + popcount prefix for pushes)
//     find_all_end_configurations post-filters      :434-520   (lane = raw configuration)
//     cost_configurations                           cost_function.py:213-304 (lane = configuration; side counting
//                                                   nearby_cone_search.py:213-297 with lanes = cones + ballots)
//     calc_final_configs_for_left_and_right         combine_traces.py:21-257 (wave-uniform)
//
// LDS per frame (one wavefront): cone x/y 4 KB, type 256 B, per-cone start-cone scalars 2.3 KB, neighbour lists
// 1.5 KB, end configurations 1.5 KB, misc < 1 KB  => ~11 KB, 14 frames resident per CU.
#pragma once
#include <stddef.h>

#include "fsdp_device.h"

namespace fsdp {

// section accounting of the profiling build: PROF_MARK(k) closes the running section and opens section k
#if defined(FSDP_PROFILE) && !defined(FSDP_EMU)
#define PROF_MARK(k)                                                                          \
  do {                                                                                        \
    long long now_ = clock64();                                                               \
    if (g_prof && (threadIdx.x & 63) == 0) prof_lds()[prof_cur_] += now_ - prof_t_;            \
    prof_cur_ = (k);                                                                          \
    prof_t_ = now_;                                                                           \
  } while (0)
#define PROF_MARK_DECL(k) long long prof_t_ = clock64(); int prof_cur_ = (k)
#else
#define PROF_MARK(k)
#define PROF_MARK_DECL(k)
#endif

// The frame state of the sorting stage.  Two instantiations: the product kernel keeps it in LDS (up to 255 cones with
// one-byte indices, 64 raw end configurations per side); frames beyond either capacity are planned again by
// sort_big_kernel with the same code over a state in global memory (8192 cones, 4096 raw end configurations) — the
// reference's own buffers grow without bound (adjacency_matrix.py:21-31, end_configurations.py:74-105,346-348).
template <int CAP_, class IDX_, int ENDS_>
struct SortSharedT {
  static constexpr int CAP = CAP_;      // cones per frame
  static constexpr int ENDS = ENDS_;    // raw end configurations per side
  using idx_t = IDX_;                   // cone index in the neighbour lists
  static constexpr int NONE = (sizeof(IDX_) == 1) ? 255 : 32767;  // "no neighbour" (sorts last)
  static constexpr int MAX_N = (sizeof(IDX_) == 1 && CAP_ > 255) ? 255 : CAP_;  // largest frame (index NONE is reserved)
  double x[CAP];
  double y[CAP];
  uint8_t type[CAP];
  idx_t nbr[2][CAP][KNN];    // mutual neighbours, ascending — one adjacency per side (both DFS run together)
  uint8_t nbr_cnt[2][CAP];
  // The three phases of a frame reuse the bytes (13.0 KB instead of 18.3 KB at 256 cones: three wavefronts per SIMD):
  union {
    struct {  // ---- start cones + adjacency of both sides (sort_frame, sort_side_prepare) ----
      double dist[CAP];     // distance car->cone (start-cone selection)
      uint8_t flags[CAP];   // bit0 in ellipse, bit1 angle>0, bit2 angle<0, bit3 angle window, bit4 "in front" (skip for 2nd cone)
      idx_t knn[CAP][KNN];  // k nearest (index), NONE = none
      uint8_t knn_ok[CAP];  // bit q: knn[q] within max_dist
      uint8_t vis[CAP];
    };
    struct {  // ---- search, then evaluation ----
      int16_t ends[2][ENDS][MAX_LEN];  // raw / filtered end configurations per side, -1 padded
      union {
        struct {  // sort_dfs_both
          int16_t stack[2][MAX_STACK][2];
          double stack_ang[2][MAX_STACK];  // direction (atan2) of the edge parent -> stacked cone
          int16_t attempt[2][MAX_LEN];
          double attempt_ang[2][MAX_LEN];  // direction of the edge attempt[p-1] -> attempt[p]
        };
        struct {  // sort_side_finish
          double cost[ENDS];
          int32_t good[ENDS];
          int32_t bad[ENDS];
          uint8_t keep[ENDS];
          uint8_t keep2[ENDS];
        };
      };
    };
  };
  struct SideCtl {                 // hand-over between the per-side phases
    int32_t active;                // the side has a start cone (else: no result)
    int32_t n_first, fk0, fk1, target_length;
    int32_t adj;                   // which adjacency the DFS walks (0 for both sides of a colourless frame)
    int32_t n_ends, status;
  } ctl[2];
  int16_t all_list[ENDS * MAX_LEN > CAP ? CAP : ENDS * MAX_LEN];
  unsigned long long all_mask[CAP / 64];
  unsigned long long near_mask[CAP / 64];
  unsigned long long close_mask[CAP / 64];
  int16_t best[2][MAX_LEN];        // best configuration per side (0 = left, 1 = right)
  int32_t best_len[2];
  int32_t n_configs[2];
  double best_cost[2];
  int32_t first_k[2][2];
  int32_t adj_built;               // the mutual-kNN lists below have been built for this frame
};
using SortShared = SortSharedT<MAX_CONES, uint8_t, MAX_ENDS>;          // LDS, product kernel
using SortShared128 = SortSharedT<128, uint8_t, MAX_ENDS>;             // LDS, batches whose frames hold <= 128 cones
constexpr int BIG_CONES = 8192, BIG_ENDS = 4096;  // (indices are int16: up to 32 766 would fit; a state is ~0.7 MB of HBM)
using SortSharedBig = SortSharedT<BIG_CONES, int16_t, BIG_ENDS>;        // global memory, sort_big_kernel

// ---- trace_sorter/line_segment_intersection.py:136-200 (epsilon 1e-6) ----
__device__ inline bool segments_intersect(double a0x, double a0y, double a1x, double a1y, double b0x, double b0y, double b1x,
                                          double b1y) {
  const double eps = 1e-6;
  // homogeneous lines: cross(h0,h1), cross(h2,h3), then their cross (numpy.cross component order)
  double la0 = a0y * 1.0 - 1.0 * a1y, la1 = 1.0 * a1x - a0x * 1.0, la2 = a0x * a1y - a0y * a1x;
  double lb0 = b0y * 1.0 - 1.0 * b1y, lb1 = 1.0 * b1x - b0x * 1.0, lb2 = b0x * b1y - b0y * b1x;
  double ix = la1 * lb2 - la2 * lb1;
  double iy = la2 * lb0 - la0 * lb2;
  double iz = la0 * lb1 - la1 * lb0;
  if (fabs(iz) < eps) {
    // _handle_line_segment_intersection_parallel_case :34-72
    double dx = a1x - a0x, dy = a1y - a0y;
    bool maybe;
    double slope;
    if (dx < eps) {
      maybe = fabs(a0x - b0x) < eps;
      slope = INFINITY;
    } else {
      slope = dy / dx;
      double ia = a0y - slope * a0x;
      double ib = b0y - slope * b0x;
      maybe = fabs(ia - ib) < eps;
    }
    if (!maybe) return false;
    bool use_y = slope > 1;
    double as = use_y ? a0y : a0x, ae = use_y ? a1y : a1x, bs = use_y ? b0y : b0x, be = use_y ? b1y : b1x;
    double left_end, right_start;
    if (as < bs) {
      left_end = ae;
      right_start = fmin(bs, be);
    } else {
      left_end = be;
      right_start = fmin(as, ae);
    }
    return left_end >= right_start;
  }
  double px = ix / iz, py = iy / iz;
  double al = fmin(a0x, a1x), ar = fmax(a0x, a1x), bl = fmin(b0x, b1x), br = fmax(b0x, b1x);
  double ab = fmin(a0y, a1y), at = fmax(a0y, a1y), bb = fmin(b0y, b1y), bt = fmax(b0y, b1y);
  return (al - eps <= px && px <= ar + eps) && (bl - eps <= px && px <= br + eps) && (ab - eps <= py && py <= at + eps) &&
         (bb - eps <= py && py <= bt + eps);
}

// utils/math_utils.py:493-530; ang = atan2(diry, dirx) of the ellipse's direction vector
__device__ inline bool inside_ellipse_at(double px, double py, double cx, double cy, double ang, double major, double minor) {
  Rot2 r = make_rot(-ang);
  double qx, qy;
  rot_apply(r, px - cx, py - cy, qx, qy);
  double crit = (qx * qx) / (major * major) + (qy * qy) / (minor * minor);
  return crit < 1;
}

// The same test for the search (end_configurations.py:282-300), where the direction is an edge of the attempt: (ex, ey) =
// last cone - cone before it, ang = atan2(ey, ex) as stored when the last cone was a candidate.  cos / sin of that angle
// are ex / |e| and ey / |e| up to a few ulp, and the criterion is compared with 1: it is formed from the normalised edge
// (reciprocal square root with one Newton step, no libm call), which decides unless it lies within 1e-6 of the boundary —
// ten orders of magnitude above the difference between the two formulations; only then the reference's own arithmetic
// (cos / sin of the stored angle) is evaluated.  FSDP_EXACT_ELLIPSE: always the latter (A/B builds).
__device__ inline bool inside_ellipse_of_edge(double px, double py, double cx, double cy, double ex, double ey, double ang, double major,
                                              double minor) {
#ifndef FSDP_EXACT_ELLIPSE
  const double n2 = ex * ex + ey * ey;
#ifdef FSDP_EMU
  const double rn = 1.0 / sqrt(n2);
#else
  double rn = __builtin_amdgcn_rsq(n2);  // (n2 = 0: inf / NaN below -> the exact path)
  rn = rn * (1.5 - 0.5 * n2 * rn * rn);
#endif
  const double c = ex * rn, s = ey * rn;
  const double x = px - cx, y = py - cy;
  const double qx = x * c + y * s, qy = y * c - x * s;
  const double crit = (qx * qx) / (major * major) + (qy * qy) / (minor * minor);
  if (crit < 1.0 - 1e-6) return true;
  if (crit > 1.0 + 1e-6) return false;
#endif
  return inside_ellipse_at(px, py, cx, cy, ang, major, minor);
}

// end_configurations.py:108-223 for ONE candidate neighbour `cand` of the popped node.
// check_if_neighbor_lies_between_last_in_attempt_and_candidate (:226-257) for one (candidate, neighbour) pair
template <class SH>
__device__ __forceinline__ bool neighbour_lies_between(const SH& S, int node, int cand, int nb) {  // (cone indices only)
  if (nb == cand) return false;
  const double lx = S.x[node], ly = S.y[node];
  const double cx = S.x[cand], cy = S.y[cand];
  double vlx = lx - S.x[nb], vly = ly - S.y[nb];
  double vcx = cx - S.x[nb], vcy = cy - S.y[nb];
  double dc = norm_blas(vcx, vcy), dl = norm_blas(vlx, vly);
  return dc < 6.0 && dl < 6.0 && acos_greater(cos_between(vlx, vly, vcx, vcy), 150 * FSDP_DEG, COS_150DEG);
}

// `between` = some neighbour of `node` lies between it and the candidate (evaluated by the caller, one (candidate,
// neighbour) pair per lane); all tests are pure predicates, so their order does not matter.  The directions of the last
// two edges of the attempt (ang_sl: attempt[pos-1] -> node, ang_tl: attempt[pos-2] -> attempt[pos-1]) are the atan2
// values computed when those cones were candidates themselves (same operands, same bits); ang_cand returns the direction
// of the edge node -> candidate for the candidate's own children.
template <class SH>
__device__ inline bool candidate_can_be_added(const SH& S, const Params& P, int side, int cone_type, int pos, int node, int cand, bool between,
                                              double px, double py, double dx, double dy, double dnx, double dny, double a_car,
                                              double ang_sl, double ang_tl, double& ang_cand) {
  const double lx = S.x[node], ly = S.y[node];
  const double cx = S.x[cand], cy = S.y[cand];
  const double l2cx = cx - lx, l2cy = cy - ly;
  ang_cand = atan2(l2cy, l2cx);
  if (between) return false;
  for (int q = 0; q <= pos; q++)
    if (S.attempt[side][q] == cand) return false;
  int sl = (pos >= 1) ? S.attempt[side][pos - 1] : 0;
  if (pos >= 1) {
    // direction of the ellipse = the edge attempt[pos-1] -> node, whose atan2 is ang_sl (same operands)
    if (!inside_ellipse_of_edge(cx, cy, lx, ly, lx - S.x[sl], ly - S.y[sl], ang_sl, 6, 3)) return false;
  }
  if (pos == 0) {
    double a_n = atan2(cy - py, cx - px);
    double diff = angle_difference(a_n, a_car);
    double want = (cone_type == T_LEFT) ? 1.0 : -1.0;
    if (!((sign_of(diff) == want) || (fabs(diff) < 5 * FSDP_DEG))) return false;
  }
  bool can = true;
  if (pos >= 1) {
    double angle_1 = ang_sl;
    double angle_2 = ang_cand;
    double difference = angle_difference(angle_2, angle_1);
    double len = norm_blas(l2cx, l2cy);
    if (fabs(difference) > P.threshold_absolute_angle)
      can = false;
    else if (cone_type == T_LEFT)
      can = (difference < P.threshold_directional_angle) || (len < 4.0);
    else
      can = (difference > -P.threshold_directional_angle) || (len < 4.0);
    if (pos >= 2) {
      double angle_3 = ang_tl;
      double difference_2 = angle_difference(angle_1, angle_3);
      if (sign_of(difference) != sign_of(difference_2) && fabs(difference - difference_2) > 1.3) can = false;
    }
  }
  if (can && pos == 1) {
    int st = S.attempt[side][0];
    can = acos_less(cos_between(dx, dy, cx - S.x[st], cy - S.y[st]), FSDP_PI / 2, 0.0);
  }
  if (can) {
    const double car_size = 2.1;
    double csx = px - dnx * car_size / 2, csy = py - dny * car_size / 2;
    double cex = px + dnx * car_size, cey = py + dny * car_size;
    can = !segments_intersect(lx, ly, cx, cy, csx, csy, cex, cey);
  }
  return can;
}

// NumPy pairwise sum of the first n (<= MAX_LEN - 1 <= 15: below the 16 of the unrolled loop's second block) entries of a
// register array, static indexing only
__device__ __forceinline__ double np_sum_reg(const double (&a)[MAX_LEN], int n) {
  if (n < 8) {
    double r = 0.0;
#pragma unroll
    for (int i = 0; i < 7; i++)
      if (i < n) r += a[i];
    return r;
  }
  double res = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
#pragma unroll
  for (int i = 8; i < MAX_LEN - 1; i++)
    if (i < n) res += a[i];
  return 0.0 + res;
}

// one side (cone_type LEFT or RIGHT); side = 0 (left) / 1 (right).  All lanes call.
// Returns status (wave-uniform).
// Phase 1 of a side (S4-S7): start cones, mutual-kNN adjacency, reachability -> S.ctl[side].
// reuse_adjacency: the mutual-kNN lists were built by the other side's call and no cone of the frame carries a side
// colour, so they are the same for this side (no-colour mode builds them once per frame).
template <class SH>
__device__ __forceinline__ void sort_side_prepare(SH& S, const Params& P, int n, int cone_type, int side, double px, double py, double dx,
                                         double dy, bool reuse_adjacency) {
  const int lane = lane_id();
  const int other_type = (cone_type == T_LEFT) ? T_RIGHT : T_LEFT;
  if (lane == 0) {
    S.n_configs[side] = 0;
    S.best_len[side] = 0;
    S.first_k[side][0] = -1;
    S.first_k[side][1] = -1;
    S.best_cost[side] = 0.0;
    S.ctl[side].active = 0;
    S.ctl[side].n_ends = 0;
    S.ctl[side].status = ST_OK;
  }
  __syncthreads();
  if (n < 3) return;  // core_trace_sorter.py:272-273
  PROF_MARK_DECL(1);

  // ---------------- S4: start cones (core_trace_sorter.py:344-465) ----------------
  const int want_bit = (cone_type == T_LEFT) ? 2 : 4;  // flags bit1 (angle>0) / bit2 (angle<0)
  double bv = 0.0;
  int bi = -1;
  for (int i = lane; i < n; i += WAVE) {
    int f = S.flags[i];
    int t = S.type[i];
    bool mask_side = ((f & want_bit) && (f & 8)) || (t == cone_type);
    bool valid = (f & 1) && mask_side && (t != other_type);
    if (valid && (bi < 0 || S.dist[i] < bv)) {
      bv = S.dist[i];
      bi = i;
    }
  }
  wave_argmin(bv, bi);
  int index_1 = bi;
  if (index_1 >= 0 && bv > P.max_dist_to_first) index_1 = -1;
  if (index_1 < 0) return;  // no start cone -> side has no result
  bv = 0.0;
  bi = -1;
  for (int i = lane; i < n; i += WAVE) {
    int f = S.flags[i];
    int t = S.type[i];
    bool mask_side = ((f & want_bit) && (f & 8)) || (t == cone_type);
    bool valid = (f & 1) && mask_side && (t != other_type) && !(f & 16) && (i != index_1);
    if (valid && (bi < 0 || S.dist[i] < bv)) {
      bv = S.dist[i];
      bi = i;
    }
  }
  wave_argmin(bv, bi);
  int index_2 = bi;
  if (index_2 >= 0 && bv > P.max_dist_to_first) index_2 = -1;
  int fk0, fk1 = -1, n_first = 1;
  if (index_2 < 0) {
    fk0 = index_1;
  } else {
    double d1x = S.x[index_1] - S.x[index_2], d1y = S.y[index_1] - S.y[index_2];
    double angle_1 = angle_between(d1x, d1y, dx, dy);
    double angle_2 = angle_between(-d1x, -d1y, dx, dy);
    // cone_dir_2 = cones[index_2] - cones[index_1]: computed, not negated, in the reference
    angle_2 = angle_between(S.x[index_2] - S.x[index_1], S.y[index_2] - S.y[index_1], dx, dy);
    if (angle_1 > angle_2) {
      int t = index_1;
      index_1 = index_2;
      index_2 = t;
    }
    double dist = norm_blas(d1x, d1y);
    if (dist > P.max_dist * 1.1 || dist < 1.4) {
      fk0 = index_1;
    } else {
      fk0 = index_2;
      fk1 = index_1;
      n_first = 2;
    }
  }
  if (lane == 0) {
    S.first_k[side][0] = fk0;
    S.first_k[side][1] = fk1;
  }
  const int start_idx = fk0;

  // ---------------- S5: mutual-kNN adjacency (adjacency_matrix.py:60-128) ----------------
  PROF_MARK(2);
  const int adj = reuse_adjacency ? 0 : side;
  if (!reuse_adjacency) {
    const int k_nn = (n - 1 < P.max_n_neighbors) ? (n - 1) : P.max_n_neighbors;
    // Cones of the other side's colour neither have neighbours nor are neighbours (their rows and columns of the distance
    // matrix are inf, adjacency_matrix.py:93-99): the rows and columns that count are compacted first (ascending, in
    // S.all_list, free until S12), so a coloured frame searches half the matrix.
    int ncand = 0;
    for (int w = 0; w * WAVE < n; w++) {
      const int i = w * WAVE + lane;
      const bool c = i < n && S.type[i] != other_type;
      const unsigned long long cm = __ballot(c);
      if (c) S.all_list[ncand + __popcll(cm & ((1ull << lane) - 1ull))] = (int16_t)i;
      if (i < n) {
#pragma unroll
        for (int q = 0; q < KNN; q++) S.knn[i][q] = (typename SH::idx_t)SH::NONE;
        S.knn_ok[i] = 0;
      }
      ncand += __popcll(cm);
    }
    __syncthreads();
    // Only neighbours within max_dist survive the d^2 > max_dist^2 drop (adjacency_matrix.py:105-107), and the k nearest
    // of a cone that lie within max_dist are the k nearest of its in-range cones: a first sweep over the candidates
    // marks the in-range ones of every lane's cone (a bit per candidate position), a second one inserts just those — a
    // handful per cone — into the sorted list.
    const double md2 = P.max_dist * P.max_dist;
    constexpr int MW = SH::CAP / WAVE;
    for (int ci = lane; ci < ncand; ci += WAVE) {
      const int i = S.all_list[ci];
      double bd[KNN];
      int bj[KNN];
#pragma unroll
      for (int q = 0; q < KNN; q++) {
        bd[q] = INFINITY;
        bj[q] = SH::NONE;
      }
      const double xi = S.x[i], yi = S.y[i];
      unsigned long long inr[MW];
#pragma unroll
      for (int w = 0; w < MW; w++) inr[w] = 0ull;
      for (int j0 = 0; j0 < ncand; j0 += 8) {  // operands eight at a time (group-uniform LDS reads, one round trip)
        double xj[8], yj[8];
        int jj[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
          jj[e] = S.all_list[(j0 + e < ncand) ? j0 + e : ncand - 1];
          xj[e] = S.x[jj[e]];
          yj[e] = S.y[jj[e]];
        }
        unsigned long long bits8 = 0ull;
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const bool in = j0 + e < ncand && jj[e] != i && cdist_sq(xi, yi, xj[e], yj[e]) <= md2;
          bits8 |= (unsigned long long)in << e;
        }
#pragma unroll
        for (int w = 0; w < MW; w++)
          if (w == (j0 >> 6)) inr[w] |= bits8 << (j0 & 63);
      }
      for (;;) {
        int pos = -1;
#pragma unroll
        for (int w = 0; w < MW; w++)
          if (pos < 0 && inr[w] != 0ull) {
            pos = w * WAVE + (__ffsll(inr[w]) - 1);
            inr[w] &= inr[w] - 1ull;
          }
        if (pos < 0) break;
        const int j = S.all_list[pos];
        double d = cdist_sq(xi, yi, S.x[j], S.y[j]);
        if (d < bd[KNN - 1]) {
          // sorted insertion through registers: strict '<' finds the slot (the earlier index stays first on an exact
          // tie), from there on every element moves down one slot — a stable order, lowest index first.  (The
          // reference's np.argsort is unstable: on exact ties its pick depends on the NumPy build)
          int cj = j;
          bool ins = false;
#pragma unroll
          for (int q = 0; q < KNN; q++) {
            bool lt = ins || d < bd[q];
            ins = lt;
            double td = lt ? bd[q] : d;
            int tj = lt ? bj[q] : cj;
            bd[q] = lt ? d : bd[q];
            bj[q] = lt ? cj : bj[q];
            d = td;
            cj = tj;
          }
        }
      }
      int okm = 0;
#pragma unroll
      for (int q = 0; q < KNN; q++) {
        bool in_k = (q < k_nn) && (bj[q] != SH::NONE);
        S.knn[i][q] = in_k ? (typename SH::idx_t)bj[q] : (typename SH::idx_t)SH::NONE;
        if (in_k) okm |= (1 << q);  // (every listed neighbour lies within max_dist)
      }
      S.knn_ok[i] = (uint8_t)okm;
    }
    __syncthreads();
    for (int i = lane; i < n; i += WAVE) {
      int cnt = 0;
      int lst[KNN];
#pragma unroll
      for (int q = 0; q < KNN; q++) lst[q] = SH::NONE;
      int okm = S.knn_ok[i];
#pragma unroll
      for (int q = 0; q < KNN; q++) {
        int j = S.knn[i][q];
        if (j == SH::NONE || !(okm & (1 << q))) continue;
        int okj = S.knn_ok[j];
        bool mutual = false;
#pragma unroll
        for (int r = 0; r < KNN; r++)
          if (S.knn[j][r] == i && (okj & (1 << r))) mutual = true;
        if (mutual) {
          // ascending insertion (NONE = empty sorts last)
          int v = j;
#pragma unroll
          for (int p = 0; p < KNN; p++) {
            bool lt = v < lst[p];
            int tv = lt ? lst[p] : v;
            lst[p] = lt ? v : lst[p];
            v = tv;
          }
          cnt++;
        }
      }
#pragma unroll
      for (int q = 0; q < KNN; q++) S.nbr[adj][i][q] = (typename SH::idx_t)lst[q];
      S.nbr_cnt[adj][i] = (uint8_t)cnt;
    }
    __syncthreads();
    if (lane == 0) S.adj_built = 1;
  }
  for (int i = lane; i < n; i += WAVE) S.vis[i] = (i == start_idx) ? 1 : 0;
  __syncthreads();
  // BFS reachability from start_idx (common.py:36-67); only min(len, 12) is consumed
  int reach = 1;
  // this lane's cones keep their neighbour lists in registers for the sweeps (product kernel: <= 4 cones per lane); a
  // sweep then costs one round of independent LDS reads instead of a dependent chain per neighbour
  constexpr int NWR = (SH::CAP / WAVE <= 4) ? SH::CAP / WAVE : 1;
  constexpr bool PRELOAD = SH::CAP / WAVE <= 4;
  int nbq[NWR][KNN];
  if constexpr (PRELOAD) {
#pragma unroll
    for (int w = 0; w < NWR; w++) {
      const int i = w * WAVE + lane;
#pragma unroll
      for (int q = 0; q < KNN; q++) nbq[w][q] = (i < n && q < S.nbr_cnt[adj][i]) ? (int)S.nbr[adj][i][q] : i < n ? i : 0;
    }
  }
  for (int it = 0; it < SH::CAP && reach < P.max_length; it++) {
    int add = 0;
    unsigned long long newbits[SH::CAP / WAVE];
    if constexpr (PRELOAD) {
#pragma unroll
      for (int w = 0; w < NWR; w++) {
        const int i = w * WAVE + lane;
        bool nv = false;
        if (w * WAVE < n) {
          int any = 0;
#pragma unroll
          for (int q = 0; q < KNN; q++) any |= S.vis[nbq[w][q]];  // (padding entries point at the cone itself: not visited)
          nv = i < n && !S.vis[i] && any != 0;
        }
        newbits[w] = __ballot(nv);
        add += __popcll(newbits[w]);
      }
    } else {
    for (int w = 0; w * WAVE < n; w++) {
      int i = w * WAVE + lane;
      bool nv = false;
      if (i < n && !S.vis[i]) {
        int c = S.nbr_cnt[adj][i];
        for (int q = 0; q < c; q++)
          if (S.vis[S.nbr[adj][i][q]]) nv = true;
      }
      newbits[w] = __ballot(nv);
      add += __popcll(newbits[w]);
    }
    }
    if (add == 0) break;
    __syncthreads();
    for (int w = 0; w * WAVE < n; w++) {
      int i = w * WAVE + lane;
      if (i < n && ((newbits[w] >> lane) & 1ull)) S.vis[i] = 1;
    }
    __syncthreads();
    reach += add;
  }
  const int target_length = reach < P.max_length ? reach : P.max_length;
  PROF_MARK(7);  // closes section 2
  if (lane == 0) {
    S.ctl[side].active = 1;
    S.ctl[side].n_first = n_first;
    S.ctl[side].fk0 = fk0;
    S.ctl[side].fk1 = fk1;
    S.ctl[side].target_length = target_length;
    S.ctl[side].adj = adj;
  }
  __syncthreads();
}

template <bool WIDE>
struct PairMask {
  using type = unsigned;
};
template <>
struct PairMask<true> {
  using type = unsigned long long;
};

// Phase 2 (S8): DFS over the cost tree (end_configurations.py:320-431) of BOTH sides at once, one half-wavefront per
// side.  A pop keeps at most 5 candidate lanes and 25 (candidate, neighbour) lanes busy, so the two independent searches
// share every instruction; the loop runs until both stacks are empty.
template <class SH>
__device__ inline void sort_dfs_both(SH& S, const Params& P, double px, double py, double dx, double dy) {
  const int lane = lane_id();
  const int side = lane >> 5, sl = lane & 31;
  const int cone_type = (side == 0) ? T_LEFT : T_RIGHT;
  const bool active = S.ctl[side].active != 0;
  const int n_first = S.ctl[side].n_first, fk0 = S.ctl[side].fk0, fk1 = S.ctl[side].fk1;
  const int target_length = S.ctl[side].target_length, adj = S.ctl[side].adj;
  const double nrm = norm_blas(dx, dy);
  const double dnx = dx / nrm, dny = dy / nrm;
  const double a_car = atan2(dny, dnx);
  int sp = 0, n_ends = 0;
  int status = ST_OK;
  PROF_MARK_DECL(3);
  if (active && sl < MAX_LEN) S.attempt[side][sl] = -1;
  __syncthreads();
  if (active) {
    if (n_first == 2) {
      if (target_length < 1) status = ST_REF_UNDEFINED_DFS_OOB;
      if (sl == 0) {
        S.attempt[side][0] = (int16_t)fk0;
        S.stack[side][0][0] = (int16_t)fk1;
        S.stack[side][0][1] = 1;
        S.stack_ang[side][0] = atan2(S.y[fk1] - S.y[fk0], S.x[fk1] - S.x[fk0]);
      }
    } else if (sl == 0) {
      S.stack[side][0][0] = (int16_t)fk0;
      S.stack[side][0][1] = 0;
      S.stack_ang[side][0] = 0.0;  // unused at position 0
    }
    sp = 1;
  }
  __syncthreads();
  for (;;) {
    const bool run = active && sp > 0 && status == ST_OK;
    if (__ballot(run) == 0ull) break;
    int node = 0, pos = 0;
    double node_ang = 0.0;
    if (run) {
      sp--;
      node = S.stack[side][sp][0];
      pos = S.stack[side][sp][1];
      node_ang = S.stack_ang[side][sp];
      if (pos >= target_length) status = ST_REF_UNDEFINED_DFS_OOB;  // numpy IndexError at current_attempt[position_in_stack]
    }
    const bool go = run && status == ST_OK;
    __syncthreads();
    if (go && sl < MAX_LEN) {
      if (sl == pos) {
        S.attempt[side][sl] = (int16_t)node;
        S.attempt_ang[side][sl] = node_ang;
      }
      if (sl > pos) S.attempt[side][sl] = -1;
    }
    __syncthreads();
    const int n_nb = go ? S.nbr_cnt[adj][node] : 0;
    // up to KNN x KNN (candidate, neighbour) pairs, one per lane (5 x 5: one round of the side's 32 lanes; the wide build's
    // 8 x 8: two); candidate c owns bits [c * n_nb, (c + 1) * n_nb)
    constexpr int PAIR_ROUNDS = (KNN * KNN + 31) / 32;
    using pair_mask_t = typename PairMask<(PAIR_ROUNDS > 1)>::type;
    pair_mask_t bm = 0;
#pragma unroll
    for (int r = 0; r < PAIR_ROUNDS; r++) {
      const int pr = sl + 32 * r;
      bool btw = false;
      if (pr < n_nb * n_nb) btw = neighbour_lies_between(S, node, S.nbr[adj][node][pr / n_nb], S.nbr[adj][node][pr % n_nb]);
      bm |= (pair_mask_t)(unsigned)(__ballot(btw) >> (32 * side)) << (32 * r);
    }
    bool can = false;
    double cand_ang = 0.0;
    if (sl < n_nb) {
      const bool between = ((bm >> (sl * n_nb)) & (((pair_mask_t)1 << n_nb) - 1u)) != 0u;
      const double ang_tl = (pos >= 2) ? S.attempt_ang[side][pos - 1] : 0.0;
      can = candidate_can_be_added(S, P, side, cone_type, pos, node, S.nbr[adj][node][sl], between, px, py, dx, dy, dnx, dny, a_car,
                                   node_ang, ang_tl, cand_ang);
    }
    const unsigned m = (unsigned)(__ballot(can) >> (32 * side));
    if (go) {
      const bool has_valid = (pos < target_length - 1) && (m != 0u);
      if (has_valid) {
        if (can) {
          int slot = sp + __popc(m & ((1u << sl) - 1u));
          S.stack[side][slot][0] = (int16_t)S.nbr[adj][node][sl];
          S.stack[side][slot][1] = (int16_t)(pos + 1);
          S.stack_ang[side][slot] = cand_ang;
        }
        sp += __popc(m);
      } else if (n_ends >= SH::ENDS) {
        status = ST_OVERFLOW_ENDS;
      } else {
        if (sl < MAX_LEN) S.ends[side][n_ends][sl] = (sl < target_length) ? S.attempt[side][sl] : (int16_t)-1;
        n_ends++;
      }
    }
    __syncthreads();
  }
  PROF_MARK(7);  // closes section 3
  if (sl == 0) {
    S.ctl[side].n_ends = n_ends;
    S.ctl[side].status = status;
  }
  __syncthreads();
}

// Phase 3 of a side (S10-S12): post filters, side counting, costs, pick.  Returns the frame status of this side.
template <class SH>
__device__ __forceinline__ int sort_side_finish(SH& S, int n, int cone_type, int side, double px, double py, double dx,
                                       double dy) {
  const int lane = lane_id();
  const int other_type = (cone_type == T_LEFT) ? T_RIGHT : T_LEFT;
  if (S.ctl[side].status != ST_OK) return S.ctl[side].status;
  if (!S.ctl[side].active) return ST_OK;
  const int n_first = S.ctl[side].n_first, fk0 = S.ctl[side].fk0, fk1 = S.ctl[side].fk1;
  const int target_length = S.ctl[side].target_length, n_ends = S.ctl[side].n_ends;
  (void)other_type;
  (void)px;
  (void)py;
  PROF_MARK_DECL(4);
  // ---------------- S10: post filters (end_configurations.py:420-515), lane = raw configuration ----------------
  // (configurations in chunks of 64: one chunk in the product kernel, more in sort_big_kernel)
  const int L = target_length;
  for (int c0 = 0; c0 < n_ends; c0 += WAVE) {
    const int c = c0 + lane;
    bool keep = false;
    if (c < n_ends) {
      int16_t cfg[MAX_LEN];
      int len = 0;
#pragma unroll
      for (int l = 0; l < MAX_LEN; l++) {
        cfg[l] = S.ends[side][c][l];
        len += (cfg[l] != -1);
      }
      keep = len > 2;
      if (keep && n_first == 2) keep = (cfg[0] == fk0) && (L > 1) && (cfg[1] == fk1);
      if (keep) {
        // drop the last cone if it is not of the side's colour (:491-500)
        int am = 0;
        bool found = false;
        for (int l = 0; l < L; l++)
          if (cfg[l] == -1) {
            am = l;
            found = true;
            break;
          }
        if (!found) am = 0;
        int last_idx = ((am - 1) % L + L) % L;
        int last_cone = 0;
#pragma unroll
        for (int l = 0; l < MAX_LEN; l++)
          if (l == last_idx) last_cone = cfg[l];
        if (S.type[last_cone] != cone_type) {
#pragma unroll
          for (int l = 0; l < MAX_LEN; l++)
            if (l == last_idx) cfg[l] = -1;
          len--;
        }
        keep = len >= 3;
      }
      for (int l = 0; l < MAX_LEN; l++) S.ends[side][c][l] = cfg[l];
      S.keep[c] = keep ? 1 : 0;
    }
  }
  __syncthreads();
  // np.unique(axis=0) + prefix removal (:507-515)
  int C = 0;
  for (int c0 = 0; c0 < n_ends; c0 += WAVE) {
    const int c = c0 + lane;
    bool keep = c < n_ends && S.keep[c] != 0;
    if (keep) {
      // rows as 64-bit words (four positions each; a row starts on an 8-byte boundary): "same" = no word differs, "this row is a
      // prefix of the other" = they differ nowhere this row holds a cone (its -1 positions masked out) — the per-position
      // predicates (a == b for all l; a == b or b == -1 for all l) three or four words at a time
      constexpr int CW = MAX_LEN / 4;
      static_assert(MAX_LEN % 4 == 0 && sizeof(S.ends[0][0]) == 8 * CW && offsetof(SH, ends) % 8 == 0, "a configuration row is CW aligned 64-bit words");
      const unsigned long long* mine_row = reinterpret_cast<const unsigned long long*>(&S.ends[side][c][0]);
      unsigned long long mine[CW], holds[CW];
#pragma unroll
      for (int w = 0; w < CW; w++) {
        mine[w] = mine_row[w];
        // 0xFFFF at the positions whose entry is not -1 (0xFFFF): per 16-bit lane "~entry != 0", smeared over the lane
        const unsigned long long t = ~mine[w];
        const unsigned long long h = (((t & 0x7FFF7FFF7FFF7FFFull) + 0x7FFF7FFF7FFF7FFFull) | t) & 0x8000800080008000ull;
        holds[w] = (h - (h >> 15)) | h;
      }
      bool drop = false;
      for (int o = 0; o < n_ends && !drop; o++) {
        if (o == c || !S.keep[o]) continue;
        const unsigned long long* other = reinterpret_cast<const unsigned long long*>(&S.ends[side][o][0]);
        unsigned long long diff = 0ull, pdiff = 0ull;
#pragma unroll
        for (int w = 0; w < CW; w++) {
          const unsigned long long x = other[w] ^ mine[w];
          diff |= x;
          pdiff |= x & holds[w];
        }
        const bool same = diff == 0ull, prefix = pdiff == 0ull;
        if (same && o < c) drop = true;       // duplicate of an earlier row
        if (!same && prefix) drop = true;     // strict prefix of another row
      }
      keep = !drop;
    }
    if (c < n_ends) S.keep2[c] = keep ? 1 : 0;
    C += __popcll(__ballot(keep));
  }
  __syncthreads();
  for (int c = lane; c < n_ends; c += WAVE) S.keep[c] = S.keep2[c];
  __syncthreads();
  if (lane == 0) S.n_configs[side] = C;
  if (C == 0) return ST_OK;  // NoPathError -> side has no result

  // ---------------- S12: cones on either side (nearby_cone_search.py:213-297) ----------------
  PROF_MARK(5);
  const int n_words = (n + WAVE - 1) / WAVE;
  if (lane < SH::CAP / WAVE) {
    S.all_mask[lane] = 0ull;
  }
  __syncthreads();
  for (int c = lane; c < n_ends; c += WAVE)
    if (S.keep[c]) {
      for (int l = 0; l < MAX_LEN; l++) {
        const int v = S.ends[side][c][l];
        if (v != -1) atomicOr(&S.all_mask[v >> 6], 1ull << (v & 63));
      }
    }
  __syncthreads();
  // all_list: ascending indices of all_mask (wave-uniform build)
  int n_all = 0;
  for (int w = 0; w < n_words; w++) {
    unsigned long long mw = S.all_mask[w];
    if ((mw >> lane) & 1ull) S.all_list[n_all + __popcll(mw & ((1ull << lane) - 1ull))] = (int16_t)(w * WAVE + lane);
    n_all += __popcll(mw);
  }
  __syncthreads();
  // near_mask[j]: any i in all with D[i][j] < 36 (diag 1e7) — the 'all' cones eight at a time (indices, then their
  // coordinates: two LDS round trips per eight instead of two per cone)
  constexpr int NW = SH::CAP / WAVE;
  // The same sweep leaves, per 'all' cone a, the set of cones within 6 m of it (near6[a]: the lanes' hits as one ballot per
  // word) in the bytes of the adjacency lists, which nothing reads after the search: the counting loop below then walks
  // the handful of candidates near its cone instead of testing every candidate's distance again.
  unsigned long long* const near6 = reinterpret_cast<unsigned long long*>(&S.nbr[0][0][0]);
  constexpr int NEAR_CAP = (int)(sizeof(S.nbr) / (sizeof(unsigned long long) * NW));
  static_assert(sizeof(S.x) % 8 == 0 && (2 * sizeof(S.x) + sizeof(S.type)) % 8 == 0, "near6 must be 8-byte aligned");
  const bool tabled = n_all <= NEAR_CAP;
  unsigned long long near_w[NW], close_w[NW];
#pragma unroll
  for (int w = 0; w < NW; w++) near_w[w] = 0ull;
  for (int w = 0; w < n_words; w++) {
    int j = w * WAVE + lane;
    bool nr = false;
    const int jc = j < n ? j : 0;
    const double xj = S.x[jc], yj = S.y[jc];
    for (int a0 = 0; a0 < n_all; a0 += 8) {
      int ia[8];
      double xa[8], ya[8];
#pragma unroll
      for (int e = 0; e < 8; e++) ia[e] = S.all_list[(a0 + e < n_all) ? a0 + e : n_all - 1];
#pragma unroll
      for (int e = 0; e < 8; e++) {
        xa[e] = S.x[ia[e]];
        ya[e] = S.y[ia[e]];
      }
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const bool hit = a0 + e < n_all && ia[e] != j && cdist_sq(xa[e], ya[e], xj, yj) < 36.0;
        if (hit) nr = true;
        if (tabled && a0 + e < n_all) {
          const unsigned long long hm = __ballot(hit && j < n);
          if (lane == 0) near6[(size_t)(a0 + e) * NW + w] = hm;
        }
      }
    }
    const unsigned long long mw = __ballot(nr && j < n);
#pragma unroll
    for (int q = 0; q < NW; q++)
      if (q == w) near_w[q] = mw;
  }
  // sorted_set_diff(near_all, all) with the searchsorted quirk (:88-94) on register-resident words (every lane computes
  // the same): for every b of `all` in ascending order, drop the first element of near_all that is >= b
  if constexpr (NW <= 4) {
    // LDS kernels (two or four words): one element b of `all` per lane.  Every b looks for its element in the UNCHANGED near set
    // (near_w is not modified below), so the b are independent: each lane finds its position and clears that bit of the close set
    // (an atomic AND in LDS; two b may clear the same bit) — a handful of instructions instead of a wave-uniform loop over `all`.
    if (lane == 0) {
#pragma unroll
      for (int w = 0; w < NW; w++) {
        S.near_mask[w] = near_w[w];
        S.close_mask[w] = near_w[w];
      }
    }
    __syncthreads();
    bool undefined = false;
    for (int a0 = 0; a0 < n_all; a0 += WAVE) {
      const int a = a0 + lane;
      if (a < n_all) {
        const int b = S.all_list[a];
        const int bw = b >> 6;
        int pos = -1;
#pragma unroll
        for (int w = 0; w < NW; w++) {
          unsigned long long mw = near_w[w];
          if (w == bw) mw &= ~((1ull << (b & 63)) - 1ull);
          if (w >= bw && pos < 0 && mw) pos = w * WAVE + (__ffsll(mw) - 1);
        }
        if (pos < 0)
          undefined = true;
        else
          atomicAnd(&S.close_mask[pos >> 6], ~(1ull << (pos & 63)));
      }
    }
    if (__ballot(undefined) != 0ull) return ST_REF_UNDEFINED_SET_DIFF;
    __syncthreads();
  } else {
#pragma unroll
    for (int w = 0; w < NW; w++) close_w[w] = near_w[w];
    bool undefined = false;
    for (int a = 0; a < n_all; a++) {
      const int b = S.all_list[a];
      const int bw = b >> 6;
      int pos = -1;
#pragma unroll
      for (int w = 0; w < NW; w++) {
        unsigned long long mw = near_w[w];
        if (w == bw) mw &= ~((1ull << (b & 63)) - 1ull);
        if (w >= bw && pos < 0 && mw) pos = w * WAVE + (__ffsll(mw) - 1);
      }
      if (pos < 0) {
        undefined = true;
      } else {
#pragma unroll
        for (int w = 0; w < NW; w++)
          if (w == (pos >> 6)) close_w[w] &= ~(1ull << (pos & 63));
      }
    }
    if (undefined) return ST_REF_UNDEFINED_SET_DIFF;
    if (lane == 0) {
#pragma unroll
      for (int w = 0; w < NW; w++) {
        S.near_mask[w] = near_w[w];
        S.close_mask[w] = close_w[w];
      }
    }
    __syncthreads();
  }
  // counts per kept configuration: lane = (configuration, position in it); every lane walks the candidate cones of
  // its pair ("other" = close ∪ (all \ configuration), a few dozen bits) and tests the few that lie within 6 m.
  // (Integer counts of order-independent predicates: same values as the reference's per-cone loops.)
  for (int c = lane; c < n_ends; c += WAVE) {
    S.good[c] = 0;
    S.bad[c] = 0;
  }
  __syncthreads();
  for (int p0 = 0; p0 < n_ends * MAX_LEN; p0 += WAVE) {
    const int p = p0 + lane;
    const int c = p / MAX_LEN, j = p - c * MAX_LEN;
    if (c < n_ends && S.keep[c]) {
      const int16_t* e = S.ends[side][c];
      int clen = 0;
      for (int l = 0; l < MAX_LEN; l++) clen += (e[l] != -1);
      if (j < clen) {
        int a, b;
        if (j == 0) {
          a = e[0];
          b = e[1];
        } else if (j == clen - 1) {
          a = e[j - 1];
          b = e[j];
        } else {
          a = e[j - 1];
          b = e[j + 1];
        }
        double sdx, sdy;
        search_direction(S.x[a], S.y[a], S.x[b], S.y[b], cone_type, sdx, sdy);
        const int cj = e[j];
        const double xc = S.x[cj], yc = S.y[cj];
        // row of cj in near6 = its position in all_list = the 'all' cones below it (cj is a cone of a kept configuration)
        int rk = 0;
        if (tabled)
          for (int w = 0; w <= (cj >> 6); w++) {
            const unsigned long long mw = S.all_mask[w];
            rk += __popcll(w < (cj >> 6) ? mw : (mw & ((1ull << (cj & 63)) - 1ull)));
          }
        int good = 0, bad = 0;
        for (int w = 0; w < n_words; w++) {
          unsigned long long cm = 0ull;
          for (int l = 0; l < clen; l++) {
            int v = e[l];
            if ((v >> 6) == w) cm |= (1ull << (v & 63));
          }
          unsigned long long om = S.close_mask[w] | (S.all_mask[w] & ~cm);
          if ((cj >> 6) == w) om &= ~(1ull << (cj & 63));
          if (tabled) om &= near6[(size_t)rk * NW + w];  // (the same predicate on the same operands: cdist_sq(cone, other) < 36)
          while (om) {
            const int idx = w * WAVE + (__ffsll(om) - 1);
            om &= om - 1ull;
            if (idx < n && (tabled || cdist_sq(xc, yc, S.x[idx], S.y[idx]) < 36.0)) {
              double vx = S.x[idx] - xc, vy = S.y[idx] - yc;
              // the cosine towards -dir is the exact negative of the cosine towards +dir (IEEE negation commutes with
              // every operation of cos_between)
              const double cg = cos_between(vx, vy, sdx, sdy);
              good += acos_less(cg, (FSDP_PI / 1.5) / 2, COS_60DEG);
              bad += acos_less(-cg, (FSDP_PI / 1.5) / 2, COS_60DEG);
            }
          }
        }
        if (good) atomicAdd(&S.good[c], good);
        if (bad) atomicAdd(&S.bad[c], bad);
      }
    }
  }
  __syncthreads();
  int mval = 0x7fffffff;
  for (int c = 0; c < n_ends; c++)
    if (S.keep[c]) {
      int d = S.good[c] - S.bad[c];
      mval = d < mval ? d : mval;
    }

  // ---------------- S11: cost per configuration (cost_function.py:213-304) ----------------
  // WAVE / MAX_LEN (five) kept configurations at a time, lane = (configuration k, position a): every lane evaluates the terms of
  // its position (one turn angle, one segment length, one segment direction), then the MAX_LEN (twelve) lanes of a configuration
  // fetch each other's terms and add them up in the reference's order (all of them redundantly; lane a = 0 stores the cost).
  PROF_MARK(6);
  for (int c0 = 0; c0 < n_ends; c0 += WAVE) {
    unsigned long long km = __ballot(c0 + lane < n_ends && S.keep[(c0 + lane < n_ends) ? c0 + lane : 0]);
    while (km) {
      constexpr int PER = WAVE / MAX_LEN;                    // configurations per round: 5 (4 in the wide build)
      constexpr unsigned long long ROW = (1ull << MAX_LEN) - 1ull;  // the lanes of one configuration in a ballot
      const int k = lane / MAX_LEN, a = lane - k * MAX_LEN;  // k = PER: lanes 60..63 idle (standard build)
      int cid = -1;
#pragma unroll
      for (int q = 0; q < PER; q++)
        if (km) {
          if (q == k) cid = c0 + (__ffsll(km) - 1);
          km &= km - 1ull;
        }
      const bool act = cid >= 0;
      const int base = (k < PER) ? k * MAX_LEN : 0;
      const int16_t* e = S.ends[side][act ? cid : 0];
      // this position's cone and the two after it (rows are -1 padded to MAX_LEN; -1 wraps to the last cone like NumPy)
      const int e0 = e[a], e1 = (a + 1 < MAX_LEN) ? e[a + 1] : -1, e2 = (a + 2 < MAX_LEN) ? e[a + 2] : -1;
      const int i0 = e0 < 0 ? n + e0 : e0, i1 = e1 < 0 ? n + e1 : e1, i2 = e2 < 0 ? n + e2 : e2;
      const double x0 = S.x[i0], y0 = S.y[i0], x1 = S.x[i1], y1 = S.y[i1], x2 = S.x[i2], y2 = S.y[i2];
      const unsigned grp_in = (unsigned)((__ballot(act && a < L && e0 != -1) >> base) & ROW);
      const int clen = __popc(grp_in);
      const int na = L - 2;
      // angle cost term :41-79
      double t_ang = 0.0;
      bool is_part = false, under = false;
      if (act && a < na) {
        double n0x = x0 - x1, n0y = y0 - y1;
        if (e1 == -1) n0x = n0y = 100.0;
        double n1x = x1 - x2, n1y = y1 - y2;
        if (e2 == -1) n1x = n1y = 100.0;
        const double ang = angle_between(n1x, n1y, -n0x, -n0y);
        is_part = e2 != -1;
        const double as_cost = (FSDP_PI - ang) / FSDP_PI;
        t_ang = as_cost * (is_part ? 1.0 : 0.0);
        under = ang < 40 * FSDP_DEG && is_part;
      }
      // residual distance term (cone_distance_cost.py:15-32)
      double t_dist = 0.0;
      if (act && a < L - 1) {
        const double ddx = x1 - x0, ddy = y1 - y0;
        double d = sqrt(ddx * ddx + ddy * ddy);
        d = d * ((e1 != -1) ? 1.0 : 0.0);
        t_dist = fmax(0.0, d - 3.0);
      }
      // direction of the segment a -> a + 1, and the turn at a against the unwanted direction :149-188
      double seg_ang = 0.0;
      if (act && a < clen - 1) seg_ang = atan2(y1 - y0, x1 - x0);
      const double prev_seg = __shfl(seg_ang, (lane + WAVE - 1) & (WAVE - 1));
      double turn = 0.0;
      bool wrong = false;
      if (act && a >= 1 && a < clen - 1) {
        const double unwanted = (cone_type == T_LEFT) ? 1.0 : -1.0;
        turn = angle_difference(prev_seg, seg_ang);
        wrong = sign_of(turn) == unwanted && fabs(turn) > 40 * FSDP_DEG;
      }
      double init_cost = 0.0;
      if (act && a == 0) init_cost = angle_between(x1 - x0, y1 - y0, dx, dy);
      const unsigned part_bits = (unsigned)((__ballot(is_part) >> base) & ROW);
      const unsigned under_bits = (unsigned)((__ballot(under) >> base) & ROW);
      const unsigned wrong_bits = (unsigned)((__ballot(wrong) >> base) & ROW);
      // the configuration's terms to every one of its lanes, then the sums in the reference's order
      double tmp[MAX_LEN];  // static indexing only (fully unrolled loops) so that it stays in registers
#pragma unroll
      for (int q = 0; q < MAX_LEN; q++) tmp[q] = (q < MAX_LEN - 2) ? __shfl(t_ang, (base + q) & (WAVE - 1)) : 0.0;
      const double angle_cost = np_sum_reg(tmp, na) / (double)__popc(part_bits) * (double)(__popc(under_bits) + 1);
#pragma unroll
      for (int q = 0; q < MAX_LEN; q++) tmp[q] = (q < MAX_LEN - 1) ? __shfl(t_dist, (base + q) & (WAVE - 1)) : 0.0;
      const double dist_cost = np_sum_reg(tmp, L - 1);
      const double ncones_cost = 1.0 / (double)clen;
      double wrong_cost = 0.0;
      {
        int ns = 0;
#pragma unroll
        for (int l = 1; l < MAX_LEN - 1; l++) {
          const double tl = __shfl(turn, (base + l) & (WAVE - 1));
          if ((wrong_bits >> l) & 1u) {
            // compacting append with static indexing
#pragma unroll
            for (int q = 0; q < MAX_LEN - 2; q++)
              if (q == ns) tmp[q] = tl;
            ns++;
          }
        }
        if (clen != 3) wrong_cost = fabs(np_sum_reg(tmp, ns));
      }
      if (act && a == 0) {
        double either_cost;
        {
          int d = S.good[cid] - S.bad[cid];
          d += (mval < 0 ? -mval : mval) + 1;
          either_cost = 1.0 / (double)d;
        }
        const double f0 = 1000.0 / 9200.0, f1 = 200.0 / 9200.0, f2 = 5000.0 / 9200.0, f3 = 1000.0 / 9200.0, f4 = 0.0 / 9200.0;
        // np.sum over the 7 weighted columns (n < 8: sequential from 0); weights [1000,200,5000,1000,0,1000,1000] / 9200
        double my_cost = 0.0;
        my_cost += angle_cost * f0;
        my_cost += dist_cost * f1;
        my_cost += ncones_cost * f2;
        my_cost += init_cost * f3;
        my_cost += 0.0 * f4;
        my_cost += either_cost * f3;
        my_cost += wrong_cost * f3;
        S.cost[cid] = my_cost;
      }
    }
  }
  // argmin with np.unique's lexicographic row order as tie-break (argsort is stable for the short arrays here)
  {
    int best = -1;
    double bc = 0.0;
    __syncthreads();
    for (int c = 0; c < n_ends; c++) {
      if (!S.keep[c]) continue;
      double cc = S.cost[c];
      bool take = best < 0 || cc < bc;
      if (!take && cc == bc) {
        // lexicographic comparison of rows (ints, -1 padded)
        bool less = false;
        for (int l = 0; l < MAX_LEN; l++) {
          int a = S.ends[side][c][l], b = S.ends[side][best][l];
          if (a != b) {
            less = a < b;
            break;
          }
        }
        take = less;
      }
      if (take) {
        best = c;
        bc = cc;
      }
    }
    int blen = 0;
    for (int l = 0; l < MAX_LEN; l++) blen += (S.ends[side][best][l] != -1);
    if (lane < MAX_LEN) S.best[side][lane] = S.ends[side][best][lane];
    if (lane == 0) {
      S.best_len[side] = blen;
      S.best_cost[side] = bc;
    }
  }
  __syncthreads();
  PROF_MARK(7);  // closes section 6 (slot 7 unused)
  return ST_OK;
}

// combine_traces.py:115-275 (wave-uniform; every lane computes the same scalars)
template <class SH>
__device__ inline void combine_sides(SH& S, int& nl, int& nr) {
  nl = S.best_len[0];
  nr = S.best_len[1];
  if (nl == 0 || nr == 0) return;
  int li = -1, ri = -1;
  for (int i = 0; i < nl && li < 0; i++)
    for (int j = 0; j < nr; j++)
      if (S.best[1][j] == S.best[0][i]) {
        li = i;
        break;
      }
  if (li < 0) return;
  for (int i = 0; i < nr && ri < 0; i++)
    for (int j = 0; j < nl; j++)
      if (S.best[0][j] == S.best[1][i]) {
        ri = i;
        break;
      }
  int ls = -1, rs = -1;
  bool have = false;
  auto X = [&](int idx) { return S.x[idx]; };
  auto Y = [&](int idx) { return S.y[idx]; };
  if (li > 0 && ri > 0) {
    int pl = S.best[0][li - 1], pr = S.best[1][ri - 1], ic = S.best[0][li];
    double dl = norm_blas(X(ic) - X(pl), Y(ic) - Y(pl));
    double dr = norm_blas(X(ic) - X(pr), Y(ic) - Y(pr));
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
  auto angle_change = [&](int side, int pos) {
    int p = S.best[side][pos - 1], c = S.best[side][pos], nx = S.best[side][pos + 1];
    double a_next = atan2(Y(nx) - Y(c), X(nx) - X(c));
    double a_prev = atan2(Y(p) - Y(c), X(p) - X(c));
    return angle_difference(a_next, a_prev);
  };
  if (!have && S.best[0][li] == S.best[1][ri] && (li >= 1 && li < nl - 1) && (ri >= 1 && ri < nr - 1)) {
    double al = angle_change(0, li), ar = angle_change(1, ri);
    double sl = sign_of(al), sr = sign_of(ar);
    double absdiff = fabs(fabs(al) - fabs(ar));
    int ndiff = nl > nr ? nl - nr : nr - nl;
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
    } else if (absdiff > 5 * FSDP_DEG) {
      if (fabs(al) > fabs(ar)) {
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
  nl = ls;
  nr = rs;
}

// A ticket whose input buffers are page-locked host memory (fsdp_submit): the sorting kernel is the batch's way onto the
// device.  Every frame's wavefront reads its offsets, pose and cone block straight from the caller's buffers over PCIe —
// the cone block is read exactly once anyway, on its way into LDS — and leaves the copies in the slot's device buffers that
// the kernels behind it (matching, path stage, the route kernels) read.  No separate copy command or copy kernel in front
// of a pass: its ~0.25 ms of link time hides under the kernel's own work.  src_off == NULL: the inputs are on the device.
struct StageIn {
  const int32_t* src_off = nullptr;   // host views (the device's address of the page-locked buffers)
  const double* src_cones = nullptr;
  const double* src_poses = nullptr;
  const double* src_prev = nullptr;   // optional previous paths (n_frames,40,4)
  int32_t* dst_off = nullptr;         // = the kernel's cone_offsets / cones_xyt / poses arguments, writable
  double* dst_cones = nullptr;
  double* dst_poses = nullptr;
  double* dst_prev = nullptr;
  int n_frames = 0;
  int32_t base = 0;                   // src_off[0]: src_cones points at row `base` of the caller's array, the device copies start at 0
};

// The sorting stage of one frame on one wavefront; S = the frame state (LDS or global memory).
template <class SH>
__device__ inline void sort_frame(SH& S, const Params& P, int frame, const int32_t* __restrict__ cone_offsets, const double* __restrict__ cones_xyt,
                                  const double* __restrict__ poses, SortOut* __restrict__ out, const StageIn& stage = StageIn()) {
  const int lane = lane_id();
  const bool staging = stage.src_off != nullptr;
  const int32_t* offs = staging ? stage.src_off : cone_offsets;
  const double* pose_src = staging ? stage.src_poses : poses;
  const int off = offs[frame] - (staging ? stage.base : 0);  // (a slice of a larger batch keeps its offsets: fsdp.h fsdp_submit)
  int n = offs[frame + 1] - offs[frame];
  const int n_all = n;
  const double px = pose_src[4 * frame + 0], py = pose_src[4 * frame + 1], dx = pose_src[4 * frame + 2], dy = pose_src[4 * frame + 3];
  if (staging) {
    if (lane == 0) {
      stage.dst_off[frame] = off;
      if (frame == stage.n_frames - 1) stage.dst_off[frame + 1] = off + n;
      stage.dst_poses[4 * frame + 0] = px;
      stage.dst_poses[4 * frame + 1] = py;
      stage.dst_poses[4 * frame + 2] = dx;
      stage.dst_poses[4 * frame + 3] = dy;
    }
    if (stage.src_prev != nullptr)
      for (int e = lane; e < PATH_POINTS * 4; e += WAVE)
        stage.dst_prev[(size_t)frame * (PATH_POINTS * 4) + e] = stage.src_prev[(size_t)frame * (PATH_POINTS * 4) + e];
  }
  SortOut* o = &out[frame];
  int status = ST_OK;
  if (n > SH::MAX_N) {
    status = ST_OVERFLOW_CONES;
    n = 0;
  }
  // stage the frame's cones: coalesced loads of the (n,3) row-major block (lane = consecutive doubles)
  {
    const double* src = (staging ? stage.src_cones : cones_xyt) + 3 * (size_t)off;
    double* copy = staging ? stage.dst_cones + 3 * (size_t)off : nullptr;
    for (int e = lane; e < 3 * n; e += WAVE) {
      double v = src[e];
      if (staging) copy[e] = v;
      int i = e / 3, c = e - 3 * i;
      if (c == 0)
        S.x[i] = v;
      else if (c == 1)
        S.y[i] = v;
      else
        S.type[i] = (uint8_t)(int)v;
    }
    // (a frame beyond this state's capacity is planned by sort_big_kernel from the device copy: it needs all of its cones there)
    if (staging && n != n_all)
      for (int e = lane; e < 3 * n_all; e += WAVE) copy[e] = src[e];
  }
  __syncthreads();
  // per-cone scalars of mask_cone_can_be_first_in_config (core_trace_sorter.py:379-407)
  {
    double car_ang = atan2(dy, dx);
    Rot2 r = make_rot(-car_ang);
    for (int i = lane; i < n; i += WAVE) {
      double rx, ry;
      rot_apply(r, S.x[i] - px, S.y[i] - py, rx, ry);
      double ang = atan2(ry, rx);
      S.dist[i] = sqrt(rx * rx + ry * ry);
      // points_inside_ellipse(major = max_dist_to_first * 1.5, minor = max_dist_to_first / 1.5) — core_trace_sorter.py:388-394
      const double emaj = P.max_dist_to_first * 1.5, emin = P.max_dist_to_first / 1.5;
      double crit = (rx * rx) / (emaj * emaj) + (ry * ry) / (emin * emin);
      int f = 0;
      if (crit < 1) f |= 1;
      if (ang > 0) f |= 2;
      if (ang < 0) f |= 4;
      if (fabs(ang) < FSDP_PI - FSDP_PI / 5 && fabs(ang) > FSDP_PI / 10) f |= 8;
      if (acos_less(cos_between(S.x[i] - px, S.y[i] - py, dx, dy), FSDP_PI / 2, 0.0)) f |= 16;  // |angle to the car| < 90 deg
      S.flags[i] = (uint8_t)f;
    }
  }
  __syncthreads();
  if (lane == 0) S.adj_built = 0;
  // no cone with a side colour: both sides see the same distance matrix and hence the same mutual-kNN adjacency
  bool coloured = false;
  for (int i = lane; i < n; i += WAVE) coloured = coloured || S.type[i] == T_LEFT || S.type[i] == T_RIGHT;
  const bool colourless = __ballot(coloured) == 0ull;
#ifndef FSDP_SORT_STOP
#define FSDP_SORT_STOP 9  // (instruction accounting builds stop the stage after phase 1..4: tools/sort_phase_insts.sh)
#endif
  if (status == ST_OK && FSDP_SORT_STOP > 1) {
    sort_side_prepare(S, P, n, T_LEFT, 0, px, py, dx, dy, false);
    // (the left call returns before building the adjacency when it finds no start cone or n < 3)
    const bool left_built = S.adj_built != 0;
    sort_side_prepare(S, P, n, T_RIGHT, 1, px, py, dx, dy, colourless && left_built);
    if (FSDP_SORT_STOP > 2) sort_dfs_both(S, P, px, py, dx, dy);
    if (FSDP_SORT_STOP > 3) {
      status = sort_side_finish(S, n, T_LEFT, 0, px, py, dx, dy);
      __syncthreads();
      if (status == ST_OK) status = sort_side_finish(S, n, T_RIGHT, 1, px, py, dx, dy);
    }
  }
  __syncthreads();
  int nl = 0, nr = 0;
  if (status == ST_OK && FSDP_SORT_STOP > 4) combine_sides(S, nl, nr);
  if (lane == 0) {
    o->status = status;
    o->n_left = nl;
    o->n_right = nr;
    o->n_configs_left = S.n_configs[0];
    o->n_configs_right = S.n_configs[1];
    o->first_k_left[0] = S.first_k[0][0];
    o->first_k_left[1] = S.first_k[0][1];
    o->first_k_right[0] = S.first_k[1][0];
    o->first_k_right[1] = S.first_k[1][1];
    o->best_cost_left = S.best_cost[0];
    o->best_cost_right = S.best_cost[1];
  }
  if (lane < MAX_LEN) {
    o->left_idx[lane] = (status == ST_OK && lane < nl) ? (int32_t)S.best[0][lane] : -1;
    o->right_idx[lane] = (status == ST_OK && lane < nr) ? (int32_t)S.best[1][lane] : -1;
  }
  __syncthreads();
}

// One workgroup (= one wavefront) per frame, frame state in LDS.  big (optional): [0] = counter, [1..] = frames beyond
// the LDS capacities (more cones than the state holds, more than 64 raw end configurations), planned again by
// sort_big_kernel.
template <class SH>
__device__ __forceinline__ void sort_kernel_body(SH& S, int n_frames, const int32_t* __restrict__ cone_offsets,
                                                 const double* __restrict__ cones_xyt, const double* __restrict__ poses,
                                                 SortOut* __restrict__ out, int* __restrict__ big, const Params* __restrict__ prm,
                                                 const StageIn& stage) {
  const int frame = blockIdx.x;
  if (frame >= n_frames) return;
  PROF_INIT();
  sort_frame(S, *prm, frame, cone_offsets, cones_xyt, poses, out, stage);
  if (big != nullptr && lane_id() == 0 && (out[frame].status == ST_OVERFLOW_CONES || out[frame].status == ST_OVERFLOW_ENDS))
    big[1 + atomicAdd(&big[0], 1)] = frame;
  PROF_FLUSH();
}
// Up to 255 cones per frame.  Three wavefronts per SIMD (168 registers, 13.0 KB of LDS per frame): the stage is a chain of
// short dependent sections, and the third resident wavefront fills issue slots the other two leave open (+8 % frames/s over
// two; 40 bytes of spill).
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3)))
sort_kernel(int n_frames, const int32_t* __restrict__ cone_offsets, const double* __restrict__ cones_xyt,
            const double* __restrict__ poses, SortOut* __restrict__ out, int* __restrict__ big, const Params* __restrict__ prm,
            StageIn stage = StageIn()) {
  __shared__ SortShared S;
  sort_kernel_body(S, n_frames, cone_offsets, cones_xyt, poses, out, big, prm, stage);
}
// The same code over a state for up to 128 cones (the host launches it when no frame of the batch holds more): the cone
// arrays, neighbour lists and bit masks are half as long, which makes a frame SORT128_LDS and lets a SIMD hold
// SORT128_WAVES wavefronts.
#ifndef SORT128_WAVES
#ifdef FSDP_WIDE_SHAPES
#define SORT128_WAVES 3  // (the wide shapes' frame state is 12.4 KB: three wavefronts per SIMD by LDS anyway — at four by registers it spilled 48 of them)
#else
#define SORT128_WAVES 4
#endif
#endif
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SORT128_WAVES)))
sort_kernel_128(int n_frames, const int32_t* __restrict__ cone_offsets, const double* __restrict__ cones_xyt,
                const double* __restrict__ poses, SortOut* __restrict__ out, int* __restrict__ big, const Params* __restrict__ prm,
                StageIn stage = StageIn()) {
  __shared__ SortShared128 S;
  sort_kernel_body(S, n_frames, cone_offsets, cones_xyt, poses, out, big, prm, stage);
}

// The frames sort_kernel could not hold in LDS, with the frame state in global memory (one SortSharedBig per block).
__global__ void __launch_bounds__(64) sort_big_kernel(const int32_t* __restrict__ cone_offsets, const double* __restrict__ cones_xyt,
                                                      const double* __restrict__ poses, SortOut* __restrict__ out,
                                                      const int* __restrict__ big, SortSharedBig* __restrict__ state,
                                                      const Params* __restrict__ prm) {
  const int n = big[0];
  SortSharedBig& S = state[blockIdx.x];
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    sort_frame(S, *prm, big[1 + i], cone_offsets, cones_xyt, poses, out);
    __syncthreads();
  }
}

}  // namespace fsdp
