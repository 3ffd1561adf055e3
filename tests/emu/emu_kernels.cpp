// TEST INFRASTRUCTURE — builds the product's kernel *sources* for the host SIMT emulator
// (tests/emu/hip_emu.h).  Never loaded by the package.
#include "hip_emu.h"

#include "../../ft-fsd-path-planning_amd/csrc/sort_kernel.h"
#include "../../ft-fsd-path-planning_amd/csrc/match_kernel.h"
#include "../../ft-fsd-path-planning_amd/csrc/path_kernel.h"
#include "../../ft-fsd-path-planning_amd/csrc/skidpad_kernel.h"
#include "../../ft-fsd-path-planning_amd/csrc/filter_kernel.h"

#include <mutex>
#include <vector>

// 64-byte aligned scratch like hipMalloc's (the basis records of the arena are alignas(64))
struct AlignedArena {
  double* p;
  explicit AlignedArena(size_t n) : p((double*)aligned_alloc(64, ((n * sizeof(double) + 63) / 64) * 64)) { memset(p, 0, n * sizeof(double)); }
  ~AlignedArena() { free(p); }
  double* data() { return p; }
};

// configuration constants handed to the kernels (emu_set_params; defaults = fsd_path_planning/config.py)
static fsdp::Params g_prm = {5, 12, 6.5, 6.0, 40 * FSDP_DEG, 65 * FSDP_DEG, 3.0, 5.0, 50 * FSDP_DEG, 0.2, 0.1, 5.0, 20.0, 3, 40, 0, 1};
static double g_default_path[fsdp::PATH_POINTS * 4];
static const double* g_prev_paths = nullptr;
static const double* g_gpath = nullptr;
static int g_n_gpath = 0;
static std::once_flag g_once;
static int g_last_retries = 0;
static int g_last_big = 0;
static bool g_no_sort128 = false;  // emu_set_no_sort128: the 255-cone state also for small frames (the library's option "no_sort128")
static std::vector<double> g_refit;  // FitRec per frame of the last three-kernel path launch
static void build_default() {
  double chord[fsdp::CHORD_POINTS][2];
  fsdp::default_chord_points(chord);
  AlignedArena arena(fsdp::ARENA_DOUBLES);
  emu::launch(1, 64, [&]() { fsdp::default_path_kernel(&chord[0][0], arena.data(), g_default_path, &g_prm); });
}

namespace fsdp {
// test kernel: fit one polyline (m <= PATH_CAP) and dump knots / coefficients
__global__ void fit_test_kernel(const double* xy, int m, double smoothing, double* arena, double* t_out, double* c_out, int* info, double* fp_out) {
  __shared__ PathShared<WAVE> S;
  const int lane = lane_id();
  const Arena A = frame_arena(arena, 0, &g_prm);
  for (int i = lane; i < m; i += WAVE) {
    A.x[i] = xy[2 * i];
    A.y[i] = xy[2 * i + 1];
  }
  __syncthreads();
  SplineFit f;
  double max_u;
  int rc = fit_polyline<WAVE, true>(S, A, 0, m, smoothing, f, max_u);
  if (lane == 0) {
    info[0] = rc;
    info[1] = f.n;
    info[2] = f.ier;
    info[3] = f.k;
    fp_out[0] = f.fp;
    fp_out[1] = max_u;
    constexpr int NK = SplineWS<WAVE>::NK;
    for (int i = 1; i <= f.n && i <= NK; i++) {
      t_out[i - 1] = S.ws.t[i];
      c_out[i - 1] = S.ws.c[i];
      c_out[NK + i - 1] = S.ws.c[i + f.n];
    }
  }
}
}  // namespace fsdp

template <int G>
static void emu_path_launch(int n_frames, const double* poses, const fsdp::MatchOut* matched, fsdp::PathOut* out) {
  AlignedArena arena((size_t)fsdp::ARENA_DOUBLES * n_frames);
  const unsigned per = 64 / G;
  std::vector<int> retry((size_t)n_frames + 1, 0);
  emu::launch(((unsigned)n_frames + per - 1) / per, 64, [&]() {
    fsdp::path_kernel<G>(n_frames, poses, matched, g_default_path, g_prev_paths, g_gpath, g_n_gpath, arena.data(), out,
                         G != 64 ? retry.data() : nullptr, &g_prm);
  });
  if (G != 64)  // like fsdp_lib.hip launch_path: frames beyond the packed kernels' knot capacity go through the G = 64 code
    emu::launch(8, 64, [&]() {
      fsdp::path_retry_kernel(poses, matched, g_default_path, g_prev_paths, g_gpath, g_n_gpath, arena.data(), out, retry.data(), &g_prm);
    });
}

// the path stage as the library launches it for large batches: prep -> fit -> finish -> exact re-plan of the retry list
template <int GF, int NKC, int G = fsdp::PATH_G_SPLIT, int NKP = fsdp::FIT_KNOTS>  // NKP: knots per fit of the prep / finish workspaces
static void emu_path_split_launch(int n_frames, const double* poses, const fsdp::MatchOut* matched, fsdp::PathOut* out) {
  AlignedArena arena((size_t)fsdp::ARENA_DOUBLES * n_frames);
  std::vector<fsdp::PathMid> mid(n_frames);
  std::vector<int> retry((size_t)n_frames + 1, 0);
  const unsigned per = 64 / G, perf = 64 / GF;
  emu::launch(((unsigned)n_frames + per - 1) / per, 64, [&]() {
    fsdp::path_prep_kernel<G, NKP>(n_frames, poses, matched, g_default_path, g_prev_paths, g_gpath, g_n_gpath, arena.data(), out,
                                   mid.data(), retry.data(), &g_prm);
  });
  emu::launch(((unsigned)n_frames + perf - 1) / perf, 64, [&]() { fsdp::fit_kernel<GF, NKC>(n_frames, arena.data(), mid.data(), retry.data(), &g_prm, nullptr, nullptr); });
  emu::launch(((unsigned)n_frames + per - 1) / per, 64, [&]() { fsdp::path_finish_kernel<G, NKP>(n_frames, arena.data(), mid.data(), out, retry.data(), &g_prm); });
  g_last_retries = retry[0];
  // the refit records (knots / coefficients fit_kernel handed to path_finish_kernel), kept for emu_last_refit
  g_refit.assign((size_t)n_frames * fsdp::FITREC_DOUBLES, 0.0);
  for (int f = 0; f < n_frames; f++) {
    const fsdp::Arena A = fsdp::frame_arena(arena.data(), f, &g_prm);
    memcpy(&g_refit[(size_t)f * fsdp::FITREC_DOUBLES], A.fit, sizeof(fsdp::FitRec));
  }
  emu::launch(8, 64, [&]() {
    fsdp::path_retry_kernel(poses, matched, g_default_path, g_prev_paths, g_gpath, g_n_gpath, arena.data(), out, retry.data(), &g_prm);
  });
}

// the same steps through the packed kernels (csrc/skidpad_kernel.h "steps in flight, many frames per wavefront"):
// select -> prep -> fit -> finish -> commit; lanes = lanes per frame of prep / finish (16, or 8 with a 4-lane fit)
template <int G, int GF>
static void emu_skid_packed_kernels(int frames, const fsdp::SkidSel* sel, const fsdp::SkidTables& T, const double* chord, double* arena,
                                    fsdp::PathMid* mid, fsdp::PathOut* pout, int* retry) {
  const unsigned per = 64 / G, perf = 64 / GF;
  emu::launch(((unsigned)frames + per - 1) / per, 64, [&]() { fsdp::skid_prep_kernel<G>(frames, sel, T, chord, g_default_path, arena, mid); });
  emu::launch(((unsigned)frames + perf - 1) / perf, 64, [&]() { fsdp::fit_kernel<GF, fsdp::FIT_KNOTS>(frames, arena, mid, retry, &g_prm, nullptr, nullptr); });
  emu::launch(((unsigned)frames + per - 1) / per, 64, [&]() { fsdp::path_finish_kernel<G>(frames, arena, mid, pout, retry, &g_prm); });
}

extern "C" {
int emu_last_retries() { return g_last_retries; }
// refit record of frame f of the last three-kernel launch: n knots, then knots (34), then coefficients (68); returns n
int emu_last_refit(int f, double* knots34, double* coeffs68) {
  if ((size_t)(f + 1) * fsdp::FITREC_DOUBLES > g_refit.size()) return -1;
  const fsdp::FitRec* r = (const fsdp::FitRec*)&g_refit[(size_t)f * fsdp::FITREC_DOUBLES];
  memcpy(knots34, r->t, sizeof(r->t));
  memcpy(coeffs68, r->c, sizeof(r->c));
  return r->n;
}
int emu_last_big() { return g_last_big; }
void emu_set_no_sort128(int on) { g_no_sort128 = on != 0; }
void emu_fit(const double* xy, int m, double smoothing, double* t_out, double* c_out, int* info, double* fp_out) {
  AlignedArena arena(fsdp::ARENA_DOUBLES);
  emu::launch(1, 64, [&]() { fsdp::fit_test_kernel(xy, m, smoothing, arena.data(), t_out, c_out, info, fp_out); });
}
int emu_sizeof_sort_out() { return (int)sizeof(fsdp::SortOut); }
int emu_sizeof_match_out() { return (int)sizeof(fsdp::MatchOut); }
int emu_sizeof_path_out() { return (int)sizeof(fsdp::PathOut); }

static void emu_sort_plain(int n_frames, const int32_t* offsets, const double* cones, const double* poses, fsdp::SortOut* out) {
  std::vector<int> big((size_t)n_frames + 1, 0);
  // the host library's choice (fsdp_lib.hip launch_sort): the 128-cone state when no frame of the batch holds more
  int max_cones = 0;
  for (int f = 0; f < n_frames; f++) max_cones = std::max(max_cones, (int)(offsets[f + 1] - offsets[f]));
  if (max_cones <= fsdp::SortShared128::MAX_N && !g_no_sort128)
    emu::launch((unsigned)n_frames, 64, [&]() { fsdp::sort_kernel_128(n_frames, offsets, cones, poses, out, big.data(), &g_prm); });
  else
    emu::launch((unsigned)n_frames, 64, [&]() { fsdp::sort_kernel(n_frames, offsets, cones, poses, out, big.data(), &g_prm); });
  g_last_big = big[0];
  if (big[0] > 0) {
    std::vector<fsdp::SortSharedBig> state(2);
    emu::launch(2, 64, [&]() { fsdp::sort_big_kernel(offsets, cones, poses, out, big.data(), state.data(), &g_prm); });
  }
}
// use_unknown_cones = False: the filter kernels in front (fsdp_lib.hip launch_filter); f_off / f_cones describe the batch
// the other kernels plan, f_map the way back for indices
static std::vector<int32_t> g_f_off, g_f_map;
static std::vector<double> g_f_cones;
static void emu_filter(int n_frames, const int32_t* offsets, const double* cones) {
  std::vector<int32_t> cnt((size_t)n_frames + 1, 0);
  g_f_off.assign((size_t)n_frames + 1, 0);
  const size_t total = (size_t)offsets[n_frames];
  g_f_cones.assign(3 * total + 3, 0.0);
  g_f_map.assign(total + 1, 0);
  emu::launch((unsigned)n_frames, 64, [&]() { fsdp::filter_count_kernel(n_frames, offsets, cones, cnt.data()); });
  emu::launch(1, 64, [&]() { fsdp::filter_scan_kernel(n_frames, cnt.data(), g_f_off.data()); });
  emu::launch((unsigned)n_frames, 64, [&]() { fsdp::filter_scatter_kernel(n_frames, offsets, cones, g_f_off.data(), g_f_cones.data(), g_f_map.data()); });
}
void emu_sort(int n_frames, const int32_t* offsets, const double* cones, const double* poses, fsdp::SortOut* out) {
  if (g_prm.use_unknown_cones) {
    emu_sort_plain(n_frames, offsets, cones, poses, out);
    return;
  }
  emu_filter(n_frames, offsets, cones);
  emu_sort_plain(n_frames, g_f_off.data(), g_f_cones.data(), poses, out);
}
// indices of a filtered sort back into the caller's array (assemble_kernel's remap); no-op with use_unknown_cones on
void emu_sort_remap(int n_frames, fsdp::SortOut* out) {
  if (g_prm.use_unknown_cones) return;
  for (int f = 0; f < n_frames; f++) {
    auto back = [&](int32_t& v) {
      if (v >= 0) v = g_f_map[(size_t)g_f_off[f] + v];
    };
    for (int k = 0; k < fsdp::MAX_LEN; k++) {
      back(out[f].left_idx[k]);
      back(out[f].right_idx[k]);
    }
    for (int k = 0; k < 2; k++) {
      back(out[f].first_k_left[k]);
      back(out[f].first_k_right[k]);
    }
  }
}
void emu_match(int n_frames, const int32_t* offsets, const double* cones, const double* poses, const fsdp::SortOut* sorted,
               fsdp::MatchOut* out) {
  constexpr unsigned per = 64 / fsdp::MATCH_G;
  if (!g_prm.use_unknown_cones) {  // (call after emu_sort of the same batch, before emu_sort_remap)
    offsets = g_f_off.data();
    cones = g_f_cones.data();
  }
  emu::launch(((unsigned)n_frames + per - 1) / per, 64, [&]() { fsdp::match_kernel<fsdp::MATCH_G>(n_frames, offsets, cones, poses, sorted, out, &g_prm); });
}
int emu_sizeof_skid_state() { return (int)sizeof(fsdp::SkidState); }
int emu_sizeof_skid_info() { return (int)sizeof(fsdp::SkidInfo); }

// n_steps consecutive skidpad steps for n_inst planner instances (states updated in place): the relocalization attempts
// of all of them first, then ONE skid_path_kernel launch with a wavefront per (instance, step) — the order of commands
// fsdp_skidpad_submit produces for a caller that submits ahead.  sync: n_inst + 1 words kept by the caller across calls
// (zero at the planners' start), *ticket_base likewise; step0 = number of the first step since the start.
void emu_skidpad_steps(int n_inst, int n_steps, int step0, const int32_t* const* offsets, const double* const* cones,
                       const double* const* poses, fsdp::SkidState* states, const double* half_table, int n_path, const double* noise,
                       int n_noise, const double* ref4, double mean_distance, fsdp::PathOut* const* out, fsdp::SkidInfo* const* info,
                       uint32_t* sync, uint32_t* ticket_base) {
  std::call_once(g_once, build_default);
  fsdp::SkidTables T;
  T.path = half_table;
  T.n_path = n_path;
  T.noise = noise;
  T.n_noise = n_noise;
  T.ref_right[0] = ref4[0];
  T.ref_right[1] = ref4[1];
  T.ref_left[0] = ref4[2];
  T.ref_left[1] = ref4[3];
  T.mean_distance = mean_distance;
  T.prm = &g_prm;
  double chord[fsdp::CHORD_POINTS][2];
  fsdp::default_chord_points(chord);
  AlignedArena arena((size_t)fsdp::ARENA_DOUBLES * n_inst * n_steps);
  std::vector<int32_t> status((size_t)n_inst * n_steps, 0);
  fsdp::SkidGroup G;
  memset(&G, 0, sizeof(G));
  for (int k = 0; k < n_steps; k++) {
    double* ar = arena.data() + (size_t)fsdp::ARENA_DOUBLES * n_inst * k;
    int32_t* stat = status.data() + (size_t)n_inst * k;
    emu::launch((unsigned)n_inst, 64, [&]() { fsdp::skid_reloc_kernel(n_inst, offsets[k], cones[k], poses[k], states, T, ar, stat, step0 + k); });
    G.step[k] = fsdp::SkidStep{poses[k], stat, ar, out[k], info[k]};
  }
  G.n_steps = n_steps;
  G.step0 = step0;
  G.ticket_base = *ticket_base;
  *ticket_base += (uint32_t)n_inst * (uint32_t)n_steps;
  emu::launch((unsigned)(n_inst * n_steps), 64, [&]() { fsdp::skid_path_kernel(n_inst, G, states, T, &chord[0][0], sync); });
}

// n_steps consecutive skidpad steps through the packed kernels (emu_skid_packed_kernels); returns the number of steps the
// planners' own wavefronts planned
int emu_skidpad_steps_packed(int lanes, int n_inst, int n_steps, int step0, const int32_t* const* offsets, const double* const* cones,
                             const double* const* poses, fsdp::SkidState* states, const double* half_table, int n_path, const double* noise,
                             int n_noise, const double* ref4, double mean_distance, fsdp::PathOut* const* out, fsdp::SkidInfo* const* info, uint32_t* sync) {
  std::call_once(g_once, build_default);
  fsdp::SkidTables T;
  T.path = half_table;
  T.n_path = n_path;
  T.noise = noise;
  T.n_noise = n_noise;
  T.ref_right[0] = ref4[0];
  T.ref_right[1] = ref4[1];
  T.ref_left[0] = ref4[2];
  T.ref_left[1] = ref4[3];
  T.mean_distance = mean_distance;
  T.prm = &g_prm;
  double chord[fsdp::CHORD_POINTS][2];
  fsdp::default_chord_points(chord);
  const int frames = n_inst * n_steps;
  AlignedArena arena((size_t)fsdp::ARENA_DOUBLES * frames);
  std::vector<int32_t> status((size_t)frames, 0);
  std::vector<fsdp::PathMid> mid(frames);
  std::vector<fsdp::PathOut> pout(frames);
  std::vector<fsdp::SkidSel> sel(frames);
  std::vector<int> retry((size_t)frames + 1, 0);
  fsdp::SkidGroup G;
  memset(&G, 0, sizeof(G));
  for (int k = 0; k < n_steps; k++) {
    double* ar = arena.data() + (size_t)fsdp::ARENA_DOUBLES * n_inst * k;
    int32_t* stat = status.data() + (size_t)n_inst * k;
    emu::launch((unsigned)n_inst, 64, [&]() { fsdp::skid_reloc_kernel(n_inst, offsets[k], cones[k], poses[k], states, T, ar, stat, step0 + k); });
    G.step[k] = fsdp::SkidStep{poses[k], stat, ar, out[k], info[k]};
  }
  G.n_steps = n_steps;
  G.step0 = step0;
  emu::launch((unsigned)n_inst, 64, [&]() { fsdp::skid_select_kernel(n_inst, G, states, T, sel.data()); });
  if (lanes == 8)
    emu_skid_packed_kernels<8, 4>(frames, sel.data(), T, &chord[0][0], arena.data(), mid.data(), pout.data(), retry.data());
  else
    emu_skid_packed_kernels<16, 16>(frames, sel.data(), T, &chord[0][0], arena.data(), mid.data(), pout.data(), retry.data());
  int serial = 0;
  for (int f = 0; f < frames; f++) serial += mid[f].status != fsdp::ST_OK && sel[f].status == fsdp::ST_OK;
  emu::launch((unsigned)n_inst, 64, [&]() {
    fsdp::skid_commit_kernel(n_inst, G, states, T, &chord[0][0], sel.data(), mid.data(), pout.data(), arena.data(), sync);
  });
  return serial;  // steps the planners' own wavefronts planned
}

// calculate_reference_centers_for_skidpad_path + table spacing as the device derives them (out5)
void emu_skidpad_constants(const double* table_xy, int n_table, double* out5) {
  std::vector<double> scratch((size_t)3 * n_table);
  emu::launch(1, 64, [&]() { fsdp::skid_centers_kernel(table_xy, n_table, scratch.data(), out5); });
}

// 17 values in the order of fsdp::Params (ints as doubles); resets the cached default path
void emu_set_params(const double* v) {
  g_prm = fsdp::Params{(int32_t)v[0], (int32_t)v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11], v[12],
                       (int32_t)v[13], (int32_t)v[14], (int32_t)v[15], (int32_t)v[16]};
  build_default();
}
void emu_set_prev_paths(const double* p) { g_prev_paths = p; }
void emu_set_global_path(const double* xy, int n) {
  g_gpath = xy;
  g_n_gpath = n;
}

void emu_default_path(double* out) {
  std::call_once(g_once, build_default);
  for (int i = 0; i < fsdp::PATH_POINTS * 4; i++) out[i] = g_default_path[i];
}
// lanes per frame G: 8 / 16 / 64 (the three instantiations the product library launches)
int emu_path_g(int G, int n_frames, const double* poses, const fsdp::MatchOut* matched, fsdp::PathOut* out) {
  std::call_once(g_once, build_default);
  if (G == 8)
    emu_path_launch<8>(n_frames, poses, matched, out);
  else if (G == 16)
    emu_path_launch<16>(n_frames, poses, matched, out);
  else if (G == 64)
    emu_path_launch<64>(n_frames, poses, matched, out);
  else if (G == 1004)  // split pipeline, fit kernel with 4 lanes per frame
    emu_path_split_launch<4, fsdp::FIT_KNOTS>(n_frames, poses, matched, out);
  else if (G == 1008)
    emu_path_split_launch<8, fsdp::FIT_KNOTS>(n_frames, poses, matched, out);
  else if (G == 2008)  // the WIDE instantiations (32 knots per fit: contexts with a global path)
    emu_path_split_launch<8, fsdp::WIDE_KNOTS, 8, fsdp::WIDE_KNOTS>(n_frames, poses, matched, out);
  else if (G == 1016)  // the three kernels with 16 lanes per frame (one pass of a mid-size batch)
    emu_path_split_launch<16, fsdp::FIT_KNOTS, 16>(n_frames, poses, matched, out);
  else
    return 1;
  return 0;
}
void emu_path(int n_frames, const double* poses, const fsdp::MatchOut* matched, fsdp::PathOut* out) {
  emu_path_g(fsdp::PATH_G_THROUGHPUT, n_frames, poses, matched, out);
}
}
