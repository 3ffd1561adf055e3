// cone_matching step as a HIP kernel (gfx950): G lanes per frame (a side holds at most 24 cones with its virtual ones, so
// G = 32 puts two frames into a wavefront and every serial section advances both).
//
// Replaces ConeMatching.run_cone_matching -> calculate_virtual_cones_for_both_sides
// (reference cone_matching/core_cone_matching.py:87-124, functional_cone_matching.py:479-588):
//   search directions (match_directions.py:23-44)                 lane = cone
//   candidate test + nearest other-side cone (:73-175)            lane = own cone, loop over the other side
//   virtual cone synthesis (:178-192)                             ballot-compacted
//   ordered insertion (:195-261)                                  serial over <= 12 inserts; each insert uses a
//                                                                 wave arg-min over the existing cones
//   re-match with virtual cones (:443-476)
// Parameters are the reference defaults (config.py:124-129,162; core_cone_matching.py:101-102):
// min_track_width 3, major radius 5*1.5, minor radius 3, max search angle 50 deg, non-monotonic.
#pragma once
#include "fsdp_device.h"

#ifndef FSDP_MATCH_WAVES
#define FSDP_MATCH_WAVES 4
#endif
namespace fsdp {

struct MatchShared {
  double lx[MAX_LEN], ly[MAX_LEN];      // sorted left (after the discard rule)
  double rx[MAX_LEN], ry[MAX_LEN];
  double ax[MAX_MATCH], ay[MAX_MATCH];  // left with virtual
  double bx[MAX_MATCH], by[MAX_MATCH];  // right with virtual
  double ex[MAX_MATCH + 1], ey[MAX_MATCH + 1];  // insertion work list
  double tx[MAX_MATCH], ty[MAX_MATCH];  // cones to insert
  double d1x[MAX_MATCH], d1y[MAX_MATCH];  // search directions (own side)
  double d2x[MAX_MATCH], d2y[MAX_MATCH];  // search directions (other side)
  double key[MAX_MATCH];
  double rc[MAX_MATCH], rs[MAX_MATCH];  // cos / sin of the rotation into each own cone's search frame
  int32_t anyok[MAX_MATCH];            // some other-side cone passes the candidate test
  int32_t match[MAX_MATCH];
  int32_t order[MAX_MATCH];
};

// match_directions.py:23-44
template <int G>
__device__ inline void match_dirs(const double* px, const double* py, int n, int cone_type, double* dx, double* dy) {
  const int lane = Grp<G>::lane();
  if (lane < n && n > 1) {
    int a, b;
    if (lane == 0) {
      a = 0;
      b = 1;
    } else if (lane == n - 1) {
      a = n - 2;
      b = n - 1;
    } else {
      a = lane - 1;
      b = lane + 1;
    }
    search_direction(px[a], py[a], px[b], py[b], cone_type, dx[lane], dy[lane]);
  }
  Grp<G>::sync();
}

// functional_cone_matching.py:340-384 (+ :73-175).  Result in S.match[0..n).  own dirs left in S.d1.
template <int G>
__device__ inline void matches_for_side(MatchShared& S, const Params& P, const double* px, const double* py, int n, int cone_type,
                                        const double* qx, const double* qy, int m) {
  const int lane = Grp<G>::lane();
  if (lane < MAX_MATCH) S.match[lane] = -1;
  Grp<G>::sync();
  if (n <= 1) return;
  match_dirs<G>(px, py, n, cone_type, S.d1x, S.d1y);
  const int other_type = (cone_type == T_RIGHT) ? T_LEFT : T_RIGHT;
  if (m > 1) match_dirs<G>(qx, qy, m, other_type, S.d2x, S.d2y);
  if (m == 0) return;
  // per own cone: rotation into its search frame (:100-104)
  if (lane < n) {
    const double ang = atan2(S.d1y[lane], S.d1x[lane]);
    const Rot2 rot = make_rot(-ang);
    S.rc[lane] = rot.c;
    S.rs[lane] = rot.s;
    S.anyok[lane] = 0;
  }
  Grp<G>::sync();
  // candidate test, one (own cone, other-side cone) pair per lane: ellipse, search angle, opposing directions
  for (int p = lane; p < n * m; p += G) {
    const int i = p / m, j = p - i * m;
    const double sx = px[i], sy = py[i];
    const Rot2 rot{S.rc[i], S.rs[i]};
    // radii_square = [major_radius, minor_radius] ** 2 with major = max_search_range * 1.5, minor = min_track_width
    // (core_cone_matching.py:101-102)
    const double rmaj = P.max_search_range * 1.5, rmin = P.min_track_width;
    const double r0 = rmaj * rmaj, r1 = rmin * rmin;
    double vx, vy;
    rot_apply(rot, qx[j] - sx, qy[j] - sy, vx, vy);
    double sc = (vx * vx) / r0 + (vy * vy) / r1;
    bool ok = sc < 1;
    double a = atan2(vy, vx);
    if (fabs(a / 2) > P.max_search_angle) ok = false;
    if (m > 1) {  // with a single other-side cone the reference's direction mask is empty
      if (acos_less(cos_between(S.d1x[i], S.d1y[i], S.d2x[j], S.d2y[j]), FSDP_PI / 2, 0.0)) ok = false;
    }
    if (ok) atomicOr(&S.anyok[i], 1);
  }
  Grp<G>::sync();
  // nearest other-side cone (first smallest), lane = own cone
  if (lane < n) {
    const double sx = px[lane], sy = py[lane];
    int best = 0;
    double bd = 0.0;
    for (int j = 0; j < m; j++) {
      double d = cdist_sq(sx, sy, qx[j], qy[j]);
      if (j == 0 || d < bd) {
        bd = d;
        best = j;
      }
    }
    S.match[lane] = best;
  }
  Grp<G>::sync();
  // matches_should_be_monotonic (functional_cone_matching.py:164-171): entry i survives iff its (raw) match index equals the
  // running maximum of the raw indices up to i; then entries without any potential match go (:174)
  int raw = -1;
  bool keep = false;
  if (lane < n) {
    raw = S.match[lane];
    keep = S.anyok[lane] != 0;
    if (P.matches_should_be_monotonic) {
      int mx = S.match[0];
      for (int j = 1; j <= lane; j++) mx = S.match[j] > mx ? S.match[j] : mx;
      keep = keep && raw == mx;
    }
  }
  Grp<G>::sync();
  if (lane < n) S.match[lane] = keep ? raw : -1;
  Grp<G>::sync();
}

// functional_cone_matching.py:195-261; existing/to-insert chosen by the caller.  Result in S.ex/ey, returns count.
template <int G>
__device__ inline int insert_virtual(MatchShared& S, int ne, int nt, double carx, double cary) {
  const int lane = Grp<G>::lane();
  // order_to_insert = cdist(to_insert, existing).min(axis=1).argsort()
  if (lane < nt) {
    double b = 0.0;
    for (int j = 0; j < ne; j++) {
      double d = cdist_sq(S.tx[lane], S.ty[lane], S.ex[j], S.ey[j]);
      if (j == 0 || d < b) b = d;
    }
    S.key[lane] = b;
  }
  Grp<G>::sync();
  if (lane < nt) {
    int rank = 0;
    double k = S.key[lane];
    for (int o = 0; o < nt; o++) {
      double ko = S.key[o];
      if (ko < k || (ko == k && o < lane)) rank++;
    }
    S.order[rank] = lane;
  }
  Grp<G>::sync();
  for (int r = 0; r < nt; r++) {
    const int ti = S.order[r];
    const double cx = S.tx[ti], cy = S.ty[ti];
    // two nearest existing cones (argsort of norms, stable)
    double v = 0.0;
    int idx = -1;
    if (lane < ne) {
      double ddx = S.ex[lane] - cx, ddy = S.ey[lane] - cy;
      v = sqrt(ddx * ddx + ddy * ddy);
      idx = lane;
    }
    double v1 = v;
    int i1 = idx;
    Grp<G>::argmin(v1, i1);
    int index_to_insert = -1;
    if (ne == 1) {
      // calculate_insert_index_for_one_cone :264-282
      double d_other = norm_blas(cx - carx, cy - cary);
      double d_exist = norm_blas(S.ex[0] - carx, S.ey[0] - cary);
      index_to_insert = (d_other < d_exist) ? 0 : 1;
    } else {
      double v2 = v;
      int i2 = (idx == i1) ? -1 : idx;
      Grp<G>::argmin(v2, i2);
      const int closest = i1, second = i2;
      int diff = closest - second;
      if (diff == 1 || diff == -1) {
        double ax = S.ex[closest] - cx, ay = S.ey[closest] - cy;
        double bx = S.ex[second] - cx, by = S.ey[second] - cy;
        bool between = acos_greater(cos_between(ax, ay, bx, by), FSDP_PI / 2, 0.0);
        if (between)
          index_to_insert = (closest < second ? closest : second) + 1;
        else
          index_to_insert = (closest < second) ? closest : closest + 1;
      }
    }
    if (index_to_insert >= 0) {
      double mx = 0, my = 0;
      bool mv = lane < ne && lane >= index_to_insert;
      if (mv) {
        mx = S.ex[lane];
        my = S.ey[lane];
      }
      Grp<G>::sync();
      if (mv) {
        S.ex[lane + 1] = mx;
        S.ey[lane + 1] = my;
      }
      if (lane == 0) {
        S.ex[index_to_insert] = cx;
        S.ey[index_to_insert] = cy;
      }
      ne++;
    }
    Grp<G>::sync();
  }
  // drop interior cones whose trace angle is < 85 deg (:252-259)
  if (ne >= 3) {
    bool low = false;
    double mx = 0, my = 0;
    if (lane < ne) {
      mx = S.ex[lane];
      my = S.ey[lane];
    }
    if (lane >= 1 && lane < ne - 1) {
      double nx = S.ex[lane + 1] - mx, ny = S.ey[lane + 1] - my;
      double qx = -(mx - S.ex[lane - 1]), qy = -(my - S.ey[lane - 1]);
      low = acos_less(cos_between(nx, ny, qx, qy), 85 * FSDP_DEG, COS_85DEG);
    }
    unsigned long long lowm = Grp<G>::ballot(low);
    if (lowm) {
      unsigned long long keepm = Grp<G>::ballot(lane < ne && !low);
      Grp<G>::sync();
      if (lane < ne && !low) {
        int p = __popcll(keepm & ((1ull << lane) - 1ull));
        S.ex[p] = mx;
        S.ey[p] = my;
      }
      ne = __popcll(keepm);
    }
    Grp<G>::sync();
  }
  return ne;
}

// functional_cone_matching.py:387-440: result written to (ox, oy), returns its length
template <int G>
__device__ __forceinline__ int cones_for_other_side(MatchShared& S, const Params& P, const double* px, const double* py, int n, int cone_type,
                                           const double* qx, const double* qy, int m, double carx, double cary, double* ox,
                                           double* oy) {
  const int lane = Grp<G>::lane();
  matches_for_side<G>(S, P, px, py, n, cone_type, qx, qy, m);
  bool unmatched = lane < n && S.match[lane] == -1;
  unsigned long long um = Grp<G>::ballot(unmatched);
  const int nv = __popcll(um);
  double vx = 0, vy = 0;
  if (unmatched) {
    vx = px[lane] + S.d1x[lane] * P.min_track_width;
    vy = py[lane] + S.d1y[lane] * P.min_track_width;
  }
  int no;
  Grp<G>::sync();
  if (m == 0) {
    if (unmatched) {
      int p = __popcll(um & ((1ull << lane) - 1ull));
      S.ex[p] = vx;
      S.ey[p] = vy;
    }
    no = nv;
  } else if (nv == 0) {
    if (lane < m) {
      S.ex[lane] = qx[lane];
      S.ey[lane] = qy[lane];
    }
    no = m;
  } else {
    // existing = the longer list (other side if strictly longer), to-insert = the other one
    const bool other_is_base = m > nv;
    if (unmatched) {
      int p = __popcll(um & ((1ull << lane) - 1ull));
      if (other_is_base) {
        S.tx[p] = vx;
        S.ty[p] = vy;
      } else {
        S.ex[p] = vx;
        S.ey[p] = vy;
      }
    }
    if (lane < m) {
      if (other_is_base) {
        S.ex[lane] = qx[lane];
        S.ey[lane] = qy[lane];
      } else {
        S.tx[lane] = qx[lane];
        S.ty[lane] = qy[lane];
      }
    }
    Grp<G>::sync();
    no = insert_virtual<G>(S, other_is_base ? m : nv, other_is_base ? nv : m, carx, cary);
  }
  Grp<G>::sync();
  if (no < 2) {  // :436-438 keep the originals
    if (lane < m) {
      ox[lane] = qx[lane];
      oy[lane] = qy[lane];
    }
    no = m;
  } else if (lane < no) {
    ox[lane] = S.ex[lane];
    oy[lane] = S.ey[lane];
  }
  Grp<G>::sync();
  return no;
}

constexpr int MATCH_G = (MAX_MATCH <= 32) ? 32 : 64;  // lanes per frame: the lists (<= MAX_MATCH cones incl. the virtual ones) are walked one cone per lane
static_assert(MATCH_G >= MAX_MATCH, "matching walks its lists one cone per lane");
// (four wavefronts per SIMD, 126 registers: measured +2 % frames/s over the two the allocator takes unasked)
template <int G>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FSDP_MATCH_WAVES))) match_kernel(int n_frames, const int32_t* __restrict__ cone_offsets,
                                                   const double* __restrict__ cones_xyt, const double* __restrict__ poses,
                                                   const SortOut* __restrict__ sorted, MatchOut* __restrict__ out,
                                                   const Params* __restrict__ prm) {
  __shared__ MatchShared S_all[WAVE / G];
  const Params& P = *prm;
  const int frame = blockIdx.x * (WAVE / G) + Grp<G>::index();
  if (frame >= n_frames) return;
  MatchShared& S = S_all[Grp<G>::index()];
  const int lane = Grp<G>::lane();
  const SortOut* so = &sorted[frame];
  MatchOut* o = &out[frame];
  const double carx = poses[4 * frame + 0], cary = poses[4 * frame + 1];
  int nl = so->n_left, nr = so->n_right;
  const int st_in = so->status;
  if (st_in != ST_OK) {
    nl = 0;
    nr = 0;
  }
  const double* base = cones_xyt + 3 * (size_t)cone_offsets[frame];
  if (lane < nl) {
    int idx = so->left_idx[lane];
    S.lx[lane] = base[3 * idx];
    S.ly[lane] = base[3 * idx + 1];
  }
  if (lane < nr) {
    int idx = so->right_idx[lane];
    S.rx[lane] = base[3 * idx];
    S.ry[lane] = base[3 * idx + 1];
  }
  Grp<G>::sync();
  int na = 0, nb = 0;
  if (!(nl < 2 && nr < 2)) {
    // discard the shorter side if empty or the length ratio exceeds 2 (:513-520)
    int mn = nl < nr ? nl : nr, mx = nl < nr ? nr : nl;
    bool discard = (mn == 0) || (((double)mx / (double)mn) > 2);
    if (discard) {
      if (nl < nr)
        nl = 0;
      else
        nr = 0;
    }
    // right cones with virtual: driven by the left side
    if (nl >= 2) {
      nb = cones_for_other_side<G>(S, P, S.lx, S.ly, nl, T_LEFT, S.rx, S.ry, nr, carx, cary, S.bx, S.by);
    } else {
      if (lane < nr) {
        S.bx[lane] = S.rx[lane];
        S.by[lane] = S.ry[lane];
      }
      nb = nr;
    }
    Grp<G>::sync();
    if (nr >= 2) {
      na = cones_for_other_side<G>(S, P, S.rx, S.ry, nr, T_RIGHT, S.lx, S.ly, nl, carx, cary, S.ax, S.ay);
    } else {
      if (lane < nl) {
        S.ax[lane] = S.lx[lane];
        S.ay[lane] = S.ly[lane];
      }
      na = nl;
    }
    Grp<G>::sync();
  }
  // final matching on the lists with virtual cones (:443-476)
  matches_for_side<G>(S, P, S.ax, S.ay, na, T_LEFT, S.bx, S.by, nb);
  if (lane < MAX_MATCH) {
    o->l2r[lane] = (lane < na) ? S.match[lane] : -1;
    o->left_v[lane][0] = (lane < na) ? S.ax[lane] : 0.0;
    o->left_v[lane][1] = (lane < na) ? S.ay[lane] : 0.0;
  }
  Grp<G>::sync();
  matches_for_side<G>(S, P, S.bx, S.by, nb, T_RIGHT, S.ax, S.ay, na);
  if (lane < MAX_MATCH) {
    o->r2l[lane] = (lane < nb) ? S.match[lane] : -1;
    o->right_v[lane][0] = (lane < nb) ? S.bx[lane] : 0.0;
    o->right_v[lane][1] = (lane < nb) ? S.by[lane] : 0.0;
  }
  if (lane == 0) {
    o->status = st_in;
    o->n_left_v = na;
    o->n_right_v = nb;
    o->pad = 0;
  }
}

}  // namespace fsdp
