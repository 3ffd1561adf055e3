// libfsdp_hip.so — host side of the C ABI declared in include/fsdp.h.
//
// A context owns one GPU: up to FSDP_MAX_OVERLAP *pass slots*, each with its own HIP stream, its own copy of a batch's
// inputs, its own intermediates and its own result block — so several DIFFERENT batches are in flight at once
// (fsdp_submit / fsdp_collect: H2D, the kernels of a pass and the D2H of one batch run on the slot's stream and overlap
// with the other slots') — plus one resident batch that fsdp_run / fsdp_time_runs replay (the benchmark's form).
// Launch geometry: one 64-lane workgroup (= one wavefront) per frame or per 2 / 4 / 8 / 16 frames, so a 4096-frame batch
// is thousands of workgroups over 256 CUs / 8 XCDs (consecutive frames land on consecutive XCDs; frames are independent,
// no inter-workgroup traffic).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fsdp.h"
#include "sort_kernel.h"
#include "match_kernel.h"
#include "path_kernel.h"
#include "skidpad_kernel.h"
#include "assemble_kernel.h"
#include "filter_kernel.h"
#include "fsdp_comm.h"

using namespace fsdp;

static_assert(FSDP_MAX_LEN == MAX_LEN && FSDP_MAX_MATCH == MAX_MATCH && FSDP_PATH_POINTS == PATH_POINTS &&
                  FSDP_MAX_NEIGHBORS == KNN && FSDP_MAX_CONES == BIG_CONES,
              "header/device constant mismatch");

static thread_local std::string g_create_error;

constexpr int SORT_BIG_BLOCKS = 32;

// one batch of frames on the device (CSR offsets, flattened cones, poses, optional previous paths)
struct Inputs {
  int32_t* d_off = nullptr;
  double* d_cones = nullptr;
  double* d_poses = nullptr;
  double* d_prev = nullptr;  // (n_frames,40,4), allocated on first use
  int cap_frames = 0;
  size_t cap_cones = 0;
  int cap_prev = 0;
  int n_frames = 0;
  int max_cones = 0;      // most cones in a frame (picks the sorting kernel's state size)
  bool use_prev = false;  // d_prev holds this batch's previous paths
  // the next pass's sorting kernel brings the batch onto the device itself (sort_kernel.h StageIn): device views of the
  // caller's page-locked buffers, valid for that one launch
  const int32_t* h_off = nullptr;
  const double* h_cones = nullptr;
  const double* h_poses = nullptr;
  const double* h_prev = nullptr;
  int32_t h_base = 0;              // cone_offsets[0] of the caller's batch (the device copies are rebased to 0)
  std::vector<int32_t> off_rebased;  // offsets - cone_offsets[0] for the copy paths (kept until the slot's next batch)
};

// A batch whose offsets do not start at 0 (a slice of a larger batch): the copy paths move offsets rebased to 0 and the
// cones from the slice's first row.  Returns true when `off` now points at the slot's own (pageable) rebased copy.
static bool rebase_batch(Inputs& in, int n_frames, const int32_t*& off, const double*& cones) {
  if (n_frames <= 0 || off[0] == 0) return false;
  const int32_t b = off[0];
  in.off_rebased.resize((size_t)n_frames + 1);
  for (int i = 0; i <= n_frames; i++) in.off_rebased[(size_t)i] = off[i] - b;
  if (cones) cones += 3 * (size_t)b;
  off = in.off_rebased.data();
  return true;
}

// Tickets queue up behind each other on a slot's stream (stream order protects the slot's buffers), so a slot always has
// its next batch waiting when the current one ends — the host's collect / submit round trip is off the GPU's critical path.
constexpr int SLOT_QUEUE = 2;   // tickets per slot
constexpr int N_TRAILERS = SLOT_QUEUE + 1;  // pass trailers per slot: one per ticket entry (a ticket may stay uncollected while the
                                            // slot's other entry turns over many times) + one for resident / blocking passes

// one pass slot: a stream, the inputs of the batch submitted to it, the intermediates of a pass and its results
struct Work {
  int index = 0;
  hipStream_t stream = nullptr;
  Inputs in;
  SortOut* d_sort = nullptr;
  MatchOut* d_match = nullptr;
  PathOut* d_path = nullptr;
  double* d_arena = nullptr;  // per-frame working polyline + basis cache (ARENA_DOUBLES doubles), HBM/L2 scratch
  int* d_big = nullptr;       // [0] counter + frames beyond sort_kernel's LDS capacities (n + 1 ints)
  int* d_retry = nullptr;     // [0] counter + frames for the exact re-plan kernel (n + 1 ints)
  PathMid* d_mid = nullptr;   // hand-over records of the three-kernel path stage
  fsdp_frame_result* d_result = nullptr;  // the pass's results in the ABI's layout (assemble_kernel)
  SkidInfo* d_skid_info = nullptr;        // skidpad contexts
  int32_t* d_skid_status = nullptr;       // skid_reloc_kernel's status of the step this slot holds
  bool skid_attempted = true;             // ... which ran for that step (not once every planner is relocalized)
  SortSharedBig* d_sort_big = nullptr;    // frame states of sort_big_kernel, allocated when the route is first needed
  int cap_frames = 0;
  // use_unknown_cones = False (filter_kernel.h): the batch without its UNKNOWN cones, and the way back for the indices
  int32_t* f_cnt = nullptr;
  int32_t* f_off = nullptr;
  double* f_cones = nullptr;
  int32_t* f_map = nullptr;
  int f_cap_frames = 0;
  size_t f_cap_cones = 0;
  PassTrailer* h_trailer = nullptr;  // N_TRAILERS of them: pinned, host-coherent, written by assemble_kernel
  int trailer_idx = SLOT_QUEUE;      // which one the next pass writes: a ticket's entry index, or SLOT_QUEUE (resident / blocking passes)
  PassTrailer* d_trailer = nullptr;  // their device address
  int seq = 0;                       // passes launched on this slot
  // the most recent pass launched on the slot (verify_pass re-runs it with the route kernels when they were needed)
  const Inputs* pass_in = nullptr;
  bool ran_big = false, ran_retry = false, unverified = false, pass_skid = false;
  fsdp_frame_result* result_dst = nullptr;  // where assemble_kernel writes the next pass's results: NULL = d_result; a ticket with a
                                            // page-locked result buffer: that buffer, straight over PCIe (no copy command at all)
  bool result_compact = false;              // ... as fsdp_compact_result records (fsdp_submit_compact)
  // tickets of fsdp_submit / fsdp_skidpad_submit queued on this slot's stream (id -1: free entry)
  struct Ticket {
    long long id = -1;
    int n = 0;
    bool skid = false;
    bool pending = false;  // skidpad step whose path kernel waits for the rest of its group (flush_skid)
    int seq = 0;                       // the slot's pass counter of this ticket's pass (checked against its trailer)
    bool ran_big = false, ran_retry = false;
    // the caller's buffers: valid and untouched until fsdp_collect (a pass that has to be repeated reads them again)
    const int32_t* off = nullptr;
    const double* cones = nullptr;
    const double* poses = nullptr;
    const double* prev = nullptr;
    size_t total = 0;
    int max_cones = 0;
    fsdp_frame_result* user_results = nullptr;
    fsdp_skidpad_info* user_info = nullptr;
    bool via_stage = false;                // results go through h_stage (the caller's buffer is pageable)
    bool compact = false;                  // user_results holds compact records: fsdp_path_result (skidpad step) or
                                           // fsdp_compact_result (fsdp_submit_compact)
    size_t rec_bytes() const { return !compact ? sizeof(fsdp_frame_result) : (skid ? sizeof(PathOut) : sizeof(fsdp_compact_result)); }
    fsdp_frame_result* h_stage = nullptr;  // pinned + mapped: the assembly kernel writes a pageable caller's results here
    char* h_in = nullptr;                  // pinned + mapped: a small pageable batch is packed here and read by the sorting kernel itself
    size_t cap_in = 0;
    SkidInfo* h_info = nullptr;            // pinned
    int cap_stage = 0, cap_info = 0;
    hipEvent_t done = nullptr;             // recorded behind the ticket's last command
  } tk[SLOT_QUEUE];
};

struct fsdp_ctx {
  int device = 0;
  int mission = 0;
  hipStream_t stream = nullptr;  // = slot[0].stream
  hipEvent_t ev[8] = {};
  std::string err;
  Work slot[FSDP_MAX_OVERLAP];
  Inputs res;             // the resident batch of fsdp_upload (every slot's fsdp_run pass reads it)
  bool resident = false;  // res describes a batch fsdp_run may plan
  bool res_checked = false;  // a verified pass over the resident batch has set expect_big / expect_retry exactly
  int last_n = 0;         // frames of the most recent full pass (what fsdp_download writes)
  double* d_default_path = nullptr;  // (40,4)
  Params params;                     // configuration constants (fsdp_params) ...
  Params* d_params = nullptr;        // ... and their device copy, read by every kernel
  double* d_chord = nullptr;         // (40,2) almost-straight chord (trivial path of the skidpad mission)
  double* d_gpath = nullptr;         // PathPlanner.global_path (n_gpath,2), or NULL
  int n_gpath = 0;
  // fsdp_set_option (include/fsdp.h): what a test or a measurement may pin; 0 = the library's own choice
  int force_path_mode = 0;    // "path_mode": 0 = by batch size; 1 = one kernel (64 lanes per frame); 2 = three kernels
  int fit_g = 4;              // "fit_g": lanes per frame of fit_kernel when frames are packed: 4 = exactly the Givens quad, sixteen
                              // frames per wavefront (+1.6 % frames/s over 8 since the basis records are 32 bytes)
  int force_pack = 0;         // "pack": 0 = by frames in flight; 1 = 4 frames per wavefront; 2 = packed
  std::string stage_names;    // kernels of the most recent pass, comma-separated
  bool profile_sort = false;  // profiling build: which kernel fsdp_profile_path runs
  int overlap = 1;
  unsigned turn = 0;
  int last_slot = 0;
  long long next_ticket = 0;
  int last_ticket_slot = -1;  // slot of the most recent ticket
  int outstanding = 0;  // tickets submitted and not yet collected
  // The route kernels (sort_big_kernel, path_retry_kernel) are launched only when a pass is expected to need them: a pass
  // that turns out to need a kernel it did not get is re-run with it before anybody sees its results (verify_pass), and
  // from then on the kernel is part of every pass until ROUTE_DECAY (4096) passes in a row came back with an empty list.
  bool expect_big = false, expect_retry = false;
  int retry_hint = 0;  // the longest retry list a recent pass reported (decays by an eighth per pass): sizes path_retry_kernel's grid
  int clean_big = 0, clean_retry = 0;
  bool poison = false;        // "poison": every pass first fills its intermediates and scratch with 0xFF bytes (tests: no result depends on what a buffer held before)
  int plan_chunks = 0;        // "plan_chunks": most chunks a blocking fsdp_plan_batch call is pipelined in (0: up to 4; 1: never cut)
  bool always_route = false;  // "always_route": both route kernels with every pass (tests: results never depend on the prediction)
  long long reruns = 0;  // passes re-run by verify_pass (diagnostics: fsdp_route_stats)
  bool no_sort128 = false;  // "no_sort128": always the 255-cone state of the sorting kernel (tests)
  std::vector<hipEvent_t> tev;  // per-launch timing events of fsdp_time_runs
  int timed_iters = 0, timed_stages = 0;  // the most recent fsdp_time_runs (fsdp_time_results reads its events)
  bool time_main_only = false;            // fsdp_time_detail: events only around the path stage's main kernel
  bool time_kernel_clock = false;         // fsdp_time_detail bit 1: the refit kernel's launches note their own start / end clock
  std::vector<unsigned> tev_recorded;     // per pass of the most recent fsdp_time_runs: which of its events were recorded
  unsigned long long* d_kclock = nullptr;  // [2 * kclock_cap]: first-wavefront-start | last-wavefront-end of the refit kernel, per timed pass
  int kclock_cap = 0;
  bool primed[FSDP_MAX_OVERLAP] = {};     // slot i has executed a pass of the current packing (its stream / hardware queue is set up)
  // skidpad mission
  double* d_table = nullptr;
  double* d_noise = nullptr;
  SkidTables tables = {};
  fsdp_comm::Comm comm;  // RCCL communicator of this rank (fsdp_comm_init), or none
  double skid_consts[5] = {};  // reference centres (right xy, left xy) + table spacing, computed on the device
  bool have_tables = false;
  SkidState* d_skid = nullptr;
  SkidState* d_skid_backup = nullptr;
  uint32_t* d_skid_sync = nullptr;   // [0] ticket counter of skid_path_kernel, [1 + i] steps instance i has published
  uint32_t skid_ticket_base = 0;
  int skid_step_no = 0;              // steps submitted since fsdp_skidpad_reset
  bool skid_all_reloc = false;       // a collected step reported every planner relocalized: cones have no reader any more
  int skid_group_env = 0;            // "skid_group": steps per launch when the caller submits ahead (0: chosen from the instance count)
  // fsdp_skidpad_time_groups: HIP events around the packed kernels of every group of steps (select | prep | fit | finish | commit)
  bool skid_time_groups = false;
  std::vector<hipEvent_t> skid_group_ev;  // six per group
  std::vector<int> skid_group_frames;     // (instance, step) pairs per group
  std::string skid_group_names;
  int skid_pack_min = 2048;          // (instance, step) pairs from which a group goes through the packed kernels (one step of
                                     // 2048 planners: 1.49 M frames/s packed, 1.40 M a wavefront each; 1024: 0.98 / 1.08 M)
  // workspace of a group that goes through the packed kernels, frame = step * n_instances + instance
  double* d_g_arena = nullptr;
  PathMid* d_g_mid = nullptr;
  PathOut* d_g_out = nullptr;
  int* d_g_retry = nullptr;
  SkidSel* d_g_sel = nullptr;
  size_t g_cap = 0;
  int skid_pending[SKID_GROUP_MAX] = {};  // slots whose step waits for its group's launch, oldest first
  int n_skid_pending = 0;
  int n_instances = 0;
  // pinned host staging of the stage-level entry points (hipHostMalloc; grown by ensure_staging)
  SortOut* h_sort = nullptr;
  MatchOut* h_match = nullptr;
  PathOut* h_path = nullptr;
  int cap_staging = 0;
};
// (an empty route launch costs a stream ~1 % of a pass; a pass repeated because the kernel was missing costs a whole pass and
// stalls the caller's collect: once needed, a route stays for a long time)
constexpr int ROUTE_DECAY = 4096;

#define HIP_TRY(ctx, call)                                                                       \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess) {                                                                      \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                            \
      return 2;                                                                                  \
    }                                                                                            \
  } while (0)

// Synchronous copies go through the context's own stream: the library never touches the null stream (every stream the
// process uses takes one of the runtime's GPU_MAX_HW_QUEUES hardware queues, and overlapped passes need theirs).
static hipError_t copy_sync(fsdp_ctx* c, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
  hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, c->stream);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(c->stream);
}

template <class T>
static hipError_t regrow(T*& p, size_t count) {
  if (p) (void)hipFree(p);
  p = nullptr;
  return hipMalloc(&p, sizeof(T) * (count ? count : 1));
}

// room for a batch's inputs (buffers only grow; the caller has made sure nothing in flight reads them)
static int ensure_inputs(fsdp_ctx* c, Inputs& in, int n_frames, size_t n_cones, bool with_prev) {
  if (n_frames > in.cap_frames) {
    HIP_TRY(c, regrow(in.d_off, (size_t)n_frames + 1));
    HIP_TRY(c, regrow(in.d_poses, 4 * (size_t)n_frames));
    in.cap_frames = n_frames;
  }
  if (n_cones > in.cap_cones || !in.d_cones) {
    // (with headroom: a replay's cone count creeps up from step to step, and hipFree synchronises the whole device)
    const size_t want = n_cones + n_cones / 2 + 64;
    HIP_TRY(c, regrow(in.d_cones, 3 * want));
    in.cap_cones = want;
  }
  if (with_prev && n_frames > in.cap_prev) {
    HIP_TRY(c, regrow(in.d_prev, (size_t)PATH_POINTS * 4 * (size_t)n_frames));
    in.cap_prev = n_frames;
  }
  return 0;
}
static void free_inputs(Inputs& in) {
  (void)hipFree(in.d_off);
  (void)hipFree(in.d_cones);
  (void)hipFree(in.d_poses);
  (void)hipFree(in.d_prev);
  in = Inputs();
}

// stream, trailer and intermediates of slot w for passes of up to n frames
static int ensure_work(fsdp_ctx* c, Work& w, int n) {
  if (!w.stream) HIP_TRY(c, hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
  if (!w.h_trailer) {
    HIP_TRY(c, hipHostMalloc((void**)&w.h_trailer, sizeof(PassTrailer) * N_TRAILERS, hipHostMallocMapped | hipHostMallocCoherent));
    memset(w.h_trailer, 0, sizeof(PassTrailer) * N_TRAILERS);
    HIP_TRY(c, hipHostGetDevicePointer((void**)&w.d_trailer, w.h_trailer, 0));
  }
  if (n <= w.cap_frames) return 0;
  HIP_TRY(c, hipStreamSynchronize(w.stream));
  const size_t m = (size_t)n;
  HIP_TRY(c, regrow(w.d_sort, m));
  HIP_TRY(c, regrow(w.d_match, m));
  HIP_TRY(c, regrow(w.d_path, m));
  HIP_TRY(c, regrow(w.d_arena, (size_t)ARENA_DOUBLES * m));
  HIP_TRY(c, regrow(w.d_big, m + 1));
  HIP_TRY(c, regrow(w.d_retry, m + 1));
  HIP_TRY(c, regrow(w.d_mid, m));
  HIP_TRY(c, regrow(w.d_result, m));
  if (c->mission == 2) {
    HIP_TRY(c, regrow(w.d_skid_info, m));
    HIP_TRY(c, regrow(w.d_skid_status, m));
  }
  // the list counters are zero between passes: assemble_kernel resets them at the end of every pass
  HIP_TRY(c, hipMemsetAsync(w.d_big, 0, sizeof(int), w.stream));
  HIP_TRY(c, hipMemsetAsync(w.d_retry, 0, sizeof(int), w.stream));
  w.cap_frames = n;
  return 0;
}
static void free_work(Work& w) {
  if (w.stream) (void)hipStreamSynchronize(w.stream);
  free_inputs(w.in);
  (void)hipFree(w.d_sort);
  (void)hipFree(w.d_match);
  (void)hipFree(w.d_path);
  (void)hipFree(w.d_arena);
  (void)hipFree(w.d_big);
  (void)hipFree(w.d_retry);
  (void)hipFree(w.d_mid);
  (void)hipFree(w.d_result);
  (void)hipFree(w.d_skid_info);
  (void)hipFree(w.d_skid_status);
  (void)hipFree(w.d_sort_big);
  (void)hipFree(w.f_cnt);
  (void)hipFree(w.f_off);
  (void)hipFree(w.f_cones);
  (void)hipFree(w.f_map);
  if (w.h_trailer) (void)hipHostFree(w.h_trailer);
  for (Work::Ticket& t : w.tk) {
    if (t.h_stage) (void)hipHostFree(t.h_stage);
    if (t.h_in) (void)hipHostFree(t.h_in);
    if (t.h_info) (void)hipHostFree(t.h_info);
    if (t.done) (void)hipEventDestroy(t.done);
  }
}

// pinned result staging of the stage-level entry points, n frames
static int ensure_staging(fsdp_ctx* c, int n) {
  if (n <= c->cap_staging) return 0;
  if (c->h_sort) (void)hipHostFree(c->h_sort);
  if (c->h_match) (void)hipHostFree(c->h_match);
  if (c->h_path) (void)hipHostFree(c->h_path);
  c->h_sort = nullptr;
  c->h_match = nullptr;
  c->h_path = nullptr;
  c->cap_staging = 0;
  const size_t want = (size_t)(n < 64 ? 64 : n);
  HIP_TRY(c, hipHostMalloc((void**)&c->h_sort, sizeof(SortOut) * want, hipHostMallocDefault));
  HIP_TRY(c, hipHostMalloc((void**)&c->h_match, sizeof(MatchOut) * want, hipHostMallocDefault));
  HIP_TRY(c, hipHostMalloc((void**)&c->h_path, sizeof(PathOut) * want, hipHostMallocDefault));
  c->cap_staging = (int)want;
  return 0;
}

// p as the GPU addresses it if p is page-locked host memory (fsdp_host_alloc, fsdp_host_register: mapped into the device's
// address space, reachable by kernels and by asynchronous copies), else NULL
static void* device_view(const void* p) {
  if (!p) return nullptr;
  hipPointerAttribute_t a;
  memset(&a, 0, sizeof(a));
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // (an unregistered pointer is an error in older runtimes: not ours to keep)
    return nullptr;
  }
  return a.type == hipMemoryTypeHost ? a.devicePointer : nullptr;
}
static bool is_pinned(const void* p) { return device_view(p) != nullptr; }
// the whole extent [p, p + bytes) is page-locked and one mapping: a view that starts inside a registered range and runs past
// its end would let a kernel read / write unmapped host memory (a device fault instead of an error code) — such a buffer
// takes the pageable path (round-3 advisor)
static bool is_pinned(const void* p, size_t bytes) {
  const char* dv = (const char*)device_view(p);
  if (!dv) return false;
  if (bytes <= 1) return true;
  const char* de = (const char*)device_view((const char*)p + bytes - 1);
  return de != nullptr && de - dv == (ptrdiff_t)(bytes - 1);
}

// ---- the kernels of one pass ----------------------------------------------------------------------------------------------
// sort_kernel -> [sort_big_kernel] -> match_kernel -> path stage -> [path_retry_kernel] -> assemble_kernel, all on the
// slot's stream.  The bracketed ones are the *routes* for what the fast kernels hand on (device lists): frames beyond the
// sorting kernel's LDS capacities, frames for the exact one-frame-per-wavefront path kernel.  On the bench workload both
// lists are empty in every pass, and a launch that finds its list empty still costs its place in the stream (round 2: two
// route kernels + two counter memsets = 7 % of the overlapped kernel time): they are launched only when expected
// (fsdp_ctx::expect_*), assemble_kernel reports the list lengths, and verify_pass re-runs a pass that needed a route it
// did not get.  Results never depend on the route or on the prediction.
constexpr int MAX_STAGES = FSDP_MAX_STAGES;
struct StageEvents {  // optional timing: ev[k] is recorded before stage k, ev[n_stages] after the last
  hipEvent_t* ev = nullptr;
  int n = 0;
  bool main_only = false;   // record only the events around the path stage's main kernel (and the last one of the pass)
  unsigned recorded = 0;    // bit k: ev[k] was recorded
  unsigned long long* clock_first = nullptr;  // device words for the refit kernel's own clock readings (fit_kernel)
  unsigned long long* clock_last = nullptr;
};
enum MarkKind { MARK_PLAIN = 0, MARK_MAIN = 1, MARK_LAST = 2 };
static void mark(const Work& q, StageEvents* t, MarkKind kind = MARK_PLAIN) {
  if (!t || !t->ev) return;
  if (!t->main_only || kind != MARK_PLAIN) {
    (void)hipEventRecord(t->ev[t->n], q.stream);
    t->recorded |= 1u << t->n;
  }
  t->n++;
}

static bool sort128(const fsdp_ctx* c, const Inputs& in) { return in.max_cones <= SortShared128::MAX_N && !c->no_sort128; }

static void launch_sort(fsdp_ctx* c, Work& q, const Inputs& in) {
  StageIn st;
  if (in.h_off) {
    st.src_off = in.h_off;
    st.base = in.h_base;
    st.src_cones = in.h_cones;
    st.src_poses = in.h_poses;
    st.src_prev = in.h_prev;
    st.dst_off = in.d_off;
    st.dst_cones = in.d_cones;
    st.dst_poses = in.d_poses;
    st.dst_prev = in.d_prev;
    st.n_frames = in.n_frames;
  }
  if (sort128(c, in))
    hipLaunchKernelGGL(sort_kernel_128, dim3(in.n_frames), dim3(WAVE), 0, q.stream, in.n_frames, in.d_off, in.d_cones,
                       in.d_poses, q.d_sort, q.d_big, c->d_params, st);
  else
    hipLaunchKernelGGL(sort_kernel, dim3(in.n_frames), dim3(WAVE), 0, q.stream, in.n_frames, in.d_off, in.d_cones, in.d_poses,
                       q.d_sort, q.d_big, c->d_params, st);
}
static int launch_sort_big(fsdp_ctx* c, Work& q, const Inputs& in) {
  if (!q.d_sort_big) HIP_TRY(c, hipMalloc(&q.d_sort_big, sizeof(SortSharedBig) * SORT_BIG_BLOCKS));
  hipLaunchKernelGGL(sort_big_kernel, dim3(SORT_BIG_BLOCKS), dim3(WAVE), 0, q.stream, in.d_off, in.d_cones, in.d_poses, q.d_sort, q.d_big,
                     q.d_sort_big, c->d_params);
  return 0;
}
static void launch_match(fsdp_ctx* c, Work& q, const Inputs& in) {
  hipLaunchKernelGGL(match_kernel<MATCH_G>, dim3((in.n_frames + WAVE / MATCH_G - 1) / (WAVE / MATCH_G)), dim3(WAVE), 0, q.stream,
                     in.n_frames, in.d_off, in.d_cones, in.d_poses, q.d_sort, q.d_match, c->d_params);
}
// ---- the path stage of one pass ------------------------------------------------------------------------------------------
// Small batches (<= PATH_SMALL_BATCH frames: single-frame calls, latency): one kernel, one frame per wavefront.
// Large batches: three kernels (path_kernel.h: path_prep_kernel -> fit_kernel -> path_finish_kernel).  Either way the
// frames the fast kernels hand on (retry list on the device) are planned by the exact kernel in the same stream.
template <int GF, int NKC = FIT_KNOTS>
static void launch_fit(fsdp_ctx* c, Work& q, int n, const StageEvents* t = nullptr) {
  hipLaunchKernelGGL((fit_kernel<GF, NKC>), dim3((n + WAVE / GF - 1) / (WAVE / GF)), dim3(WAVE), 0, q.stream, n,
                     q.d_arena, q.d_mid, q.d_retry, c->d_params, t ? t->clock_first : nullptr, t ? t->clock_last : nullptr);
}
template <int G, int NKC = FIT_KNOTS>
static void launch_prep(fsdp_ctx* c, Work& q, const Inputs& in, const double* prev) {
  const int n = in.n_frames;
  hipLaunchKernelGGL((path_prep_kernel<G, NKC>), dim3((n + WAVE / G - 1) / (WAVE / G)), dim3(WAVE), 0, q.stream, n, in.d_poses, q.d_match,
                     c->d_default_path, prev, c->d_gpath, c->n_gpath, q.d_arena, q.d_path, q.d_mid, q.d_retry, c->d_params);
}
template <int G, int NKC = FIT_KNOTS>
static void launch_finish(fsdp_ctx* c, Work& q, int n) {
  hipLaunchKernelGGL((path_finish_kernel<G, NKC>), dim3((n + WAVE / G - 1) / (WAVE / G)), dim3(WAVE), 0, q.stream, n, q.d_arena, q.d_mid, q.d_path,
                     q.d_retry, c->d_params);
}

// the same steps through the packed kernels (csrc/skidpad_kernel.h "steps in flight, many frames per wavefront")
static void skid_group_mark(fsdp_ctx* c) {  // (timing of the groups' kernels on request: fsdp_skidpad_time_groups)
  if (!c->skid_time_groups) return;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  (void)hipEventRecord(e, c->stream);
  c->skid_group_ev.push_back(e);
}
template <int G, int GF>
static void launch_skid_packed_kernels(fsdp_ctx* c, int frames) {
  hipStream_t xs = c->stream;
  if (c->skid_time_groups && c->skid_group_names.empty())  // (the first group's instantiations: a replay's last, shorter group may take the 16-lane ones)
    c->skid_group_names = "skid_select_kernel,skid_prep_kernel<" + std::to_string(G) + ">,fit_kernel<" + std::to_string(GF) + ">,path_finish_kernel<" +
                          std::to_string(G) + ">,skid_commit_kernel";
  skid_group_mark(c);
  hipLaunchKernelGGL(skid_prep_kernel<G>, dim3((frames + WAVE / G - 1) / (WAVE / G)), dim3(WAVE), 0, xs, frames, c->d_g_sel, c->tables, c->d_chord,
                     c->d_default_path, c->d_g_arena, c->d_g_mid);
  skid_group_mark(c);
  hipLaunchKernelGGL((fit_kernel<GF, FIT_KNOTS>), dim3((frames + WAVE / GF - 1) / (WAVE / GF)), dim3(WAVE), 0, xs, frames, c->d_g_arena, c->d_g_mid,
                     c->d_g_retry, c->d_params, (unsigned long long*)nullptr, (unsigned long long*)nullptr);
  skid_group_mark(c);
  hipLaunchKernelGGL(path_finish_kernel<G>, dim3((frames + WAVE / G - 1) / (WAVE / G)), dim3(WAVE), 0, xs, frames, c->d_g_arena, c->d_g_mid, c->d_g_out,
                     c->d_g_retry, c->d_params);
  skid_group_mark(c);
}

// Lanes per frame: a serial instruction costs its issue cycles whatever the number of active lanes, so the more frames
// share a wavefront the cheaper a frame gets — as long as there are enough wavefronts for every SIMD.  With at least
// PACK_FRAMES frames in flight (batch size x passes overlapped) the fit kernel packs 16 frames into a wavefront (4 lanes
// each: exactly the Givens quad) and the kernels around it 8; below that, 4 frames per wavefront everywhere.
constexpr int PACK_FRAMES = 12288;

// How many frames are in flight on the GPU decides the packing.  A resident batch replayed through `overlap` slots keeps
// overlap x n frames in flight; a ticket of fsdp_submit counts what is really queued (the frames of the uncollected tickets
// plus its own): a lone batch submitted to a context of depth 10 is a lone batch, and gets the lanes of one.
static long long frames_in_flight(const fsdp_ctx* c, int n, bool ticket) {
  if (!ticket) return (long long)n * c->overlap;
  long long sum = n;
  for (int i = 0; i < FSDP_MAX_OVERLAP; i++)
    for (const Work::Ticket& t : c->slot[i].tk)
      if (t.id >= 0 && !t.skid) sum += t.n;
  return sum;
}

// the path stage's fast kernels (no route, no assembly); returns whether it was the three-kernel form
static bool launch_path(fsdp_ctx* c, Work& q, const Inputs& in, StageEvents* t, std::string& names, long long in_flight) {
  const double* prev = in.use_prev ? in.d_prev : nullptr;
  const int n = in.n_frames;
  // (the packed kernels hold degree-3 fits only: a context with max_deg < 3 plans every batch with the one-kernel stage)
  // (by the frames in flight, like the packing: a 1024-frame chunk of a 4096-frame call is not a small batch)
  const bool split = c->params.max_deg != 3 ? false : (c->force_path_mode ? c->force_path_mode == 2 : in_flight > PATH_SMALL_BATCH);
  if (!split) {
    mark(q, t, MARK_MAIN);
    // A context whose fits may be of degree 1 or 2 (max_deg < 3: utils/spline_fit.py:113) has no three-kernel form (those kernels
    // hold cubic fits only); its large batches run the one-kernel stage with FOUR frames per wavefront (16 lanes each, all
    // degrees, 32 knots per fit, the scaling-free divisions; what that form cannot hold goes to the exact kernel like any other
    // frame the packed kernels hand on) instead of one frame per wavefront.
    const bool mono16 = c->force_path_mode != 1 && c->params.max_deg != 3 && in_flight > PATH_SMALL_BATCH;
    if (mono16)
      hipLaunchKernelGGL(path_kernel<PATH_G_LATENCY>, dim3((n + WAVE / PATH_G_LATENCY - 1) / (WAVE / PATH_G_LATENCY)), dim3(WAVE), 0, q.stream, n, in.d_poses,
                         q.d_match, c->d_default_path, prev, c->d_gpath, c->n_gpath, q.d_arena, q.d_path, q.d_retry, c->d_params);
    else
      hipLaunchKernelGGL(path_kernel<PATH_G_SMALL>, dim3(n), dim3(WAVE), 0, q.stream, n, in.d_poses, q.d_match, c->d_default_path, prev,
                         c->d_gpath, c->n_gpath, q.d_arena, q.d_path, q.d_retry, c->d_params);
    names += mono16 ? "path_kernel<16>," : "path_kernel<64>,";
    return split;
  }
  mark(q, t);
  // A context with a global path (set_global_path; the acceleration / ebs_test missions run on their known path) fits 100+ m
  // polylines that need 17-32 knots (87 % of such frames; never more than 32 on the reference's tables): the WIDE instantiations
  // of the three kernels — 32 knots per fit in the frame's LDS workspace, eight lanes per frame — keep them on the packed kernels
  // instead of sending every frame to the exact one.
  if (c->n_gpath > 0) {
    launch_prep<8, WIDE_KNOTS>(c, q, in, prev);
    mark(q, t, MARK_MAIN);
    launch_fit<8, WIDE_KNOTS>(c, q, n, t);
    mark(q, t, MARK_MAIN);
    launch_finish<8, WIDE_KNOTS>(c, q, n);
    names += "path_prep_kernel<8,32>,fit_kernel<8,32>,path_finish_kernel<8,32>,";
    return split;
  }
  const bool packed = c->force_pack ? c->force_pack == 2 : in_flight >= PACK_FRAMES;
  const int gf = packed ? c->fit_g : 16;
  if (packed)
    launch_prep<8>(c, q, in, prev);
  else
    launch_prep<16>(c, q, in, prev);
  mark(q, t, MARK_MAIN);
  if (gf == 4)
    launch_fit<4>(c, q, n, t);
  else if (gf == 8)
    launch_fit<8>(c, q, n, t);
  else
    launch_fit<16>(c, q, n, t);
  mark(q, t, MARK_MAIN);
  if (packed)
    launch_finish<8>(c, q, n);
  else
    launch_finish<16>(c, q, n);
  const std::string g = packed ? "8" : "16";
  names += "path_prep_kernel<" + g + ">,fit_kernel<" + std::to_string(gf) + ">,path_finish_kernel<" + g + ">,";
  return split;
}
// sized: by the retry lists the context's recent passes reported (fsdp_ctx::retry_hint) instead of for the worst case.  A pass that is
// only EXPECTED to need the kernel (one batch in thousands had a frame for it) used to launch 1024 workgroups that found an empty list:
// nothing to compute, but on a chip full of other passes' wavefronts the last of them was dispatched ~1.7 ms later, and the pass's
// assembly waits for it (the stream of different batches: 355 such launches, 12 % of the summed kernel time of the trace).  The kernel
// walks its list grid-stride: any grid plans any list.
static void launch_path_retry(fsdp_ctx* c, Work& q, const Inputs& in, bool sized = false) {
  const double* prev = in.use_prev ? in.d_prev : nullptr;
  int rb = in.n_frames < 1024 ? in.n_frames : 1024;  // one wavefront per SIMD at most; blocks beyond the list's length return at once
  if (sized) rb = std::max(1, std::min(rb, 16 + 2 * c->retry_hint));
  hipLaunchKernelGGL(path_retry_kernel, dim3(rb), dim3(WAVE), 0, q.stream, in.d_poses, q.d_match, c->d_default_path, prev, c->d_gpath,
                     c->n_gpath, q.d_arena, q.d_path, q.d_retry, c->d_params);
}
static void launch_assemble(fsdp_ctx* c, Work& q, int n, bool skid, fsdp_frame_result* dst = nullptr, hipStream_t stream = nullptr,
                            const SkidInfo* info_src = nullptr, SkidInfo* info_dst = nullptr, const int32_t* remap = nullptr,
                            const int32_t* remap_off = nullptr, bool compact = false) {
  long long blocks = ((long long)n + 3) / 4;  // one wavefront per frame, four per workgroup (grid-stride beyond the cap)
  // results that go straight to host memory leave at the link's pace: a few hundred wavefronts keep it busy, more would
  // only sit on the SIMDs' wavefront slots with their stores pending while the other slots' kernels wait for a place
  (void)c;
  const long long cap = dst ? 128 : 16384;  // (32 / 128 / 512 workgroups towards host memory: 5.1 / 5.2 / 5.4 M frames/s streamed, inside the noise: profiles/r06_streaming_probe.txt)
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  q.seq++;
  if (compact) {  // fsdp_compact_result records (into the slot's result block or the caller's page-locked buffer)
    hipLaunchKernelGGL(assemble_compact_kernel, dim3((unsigned)blocks), dim3(256), 0, stream ? stream : q.stream, n, q.d_sort, q.d_match, q.d_path,
                       (fsdp_compact_result*)(dst ? dst : q.d_result), q.d_big, q.d_retry, q.d_trailer + q.trailer_idx, q.seq, remap, remap_off);
    return;
  }
  hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)blocks), dim3(256), 0, stream ? stream : q.stream, n, skid ? (const SortOut*)nullptr : q.d_sort,
                     skid ? (const MatchOut*)nullptr : q.d_match, q.d_path, dst ? dst : q.d_result, q.d_big, q.d_retry, q.d_trailer + q.trailer_idx, q.seq,
                     (const int32_t*)info_src, (int32_t*)info_dst, info_dst ? (int)(sizeof(SkidInfo) / 4) * n : 0, remap, remap_off);
}

// use_unknown_cones = False: the batch without its UNKNOWN cones into the slot's filter buffers; returns the view the
// stage kernels plan (same poses / previous paths)
static int launch_filter(fsdp_ctx* c, Work& q, const Inputs& in, Inputs* view) {
  if (in.cap_frames > q.f_cap_frames || !q.f_off) {
    HIP_TRY(c, hipStreamSynchronize(q.stream));
    HIP_TRY(c, regrow(q.f_cnt, (size_t)in.cap_frames));
    HIP_TRY(c, regrow(q.f_off, (size_t)in.cap_frames + 1));
    q.f_cap_frames = in.cap_frames;
  }
  if (in.cap_cones > q.f_cap_cones || !q.f_cones) {
    HIP_TRY(c, hipStreamSynchronize(q.stream));
    HIP_TRY(c, regrow(q.f_cones, 3 * in.cap_cones));
    HIP_TRY(c, regrow(q.f_map, in.cap_cones));
    q.f_cap_cones = in.cap_cones;
  }
  const int n = in.n_frames;
  hipLaunchKernelGGL(filter_count_kernel, dim3(n), dim3(WAVE), 0, q.stream, n, in.d_off, in.d_cones, q.f_cnt);
  hipLaunchKernelGGL(filter_scan_kernel, dim3(1), dim3(1024), 0, q.stream, n, q.f_cnt, q.f_off);
  hipLaunchKernelGGL(filter_scatter_kernel, dim3(n), dim3(WAVE), 0, q.stream, n, in.d_off, in.d_cones, q.f_off, q.f_cones, q.f_map);
  *view = in;
  view->d_off = q.f_off;
  view->d_cones = q.f_cones;
  return 0;
}

// sorting -> matching -> path stage -> result assembly of batch `in` on slot q
// in_flight: frames on the GPU while this pass runs, its own included (launch_path); < 0: a resident batch replayed through
// every slot of the overlap depth
static int launch_pass(fsdp_ctx* c, Work& q, const Inputs& in_, StageEvents* t = nullptr, bool force_routes = false, long long in_flight = -1) {
  c->primed[q.index] = true;
  const bool with_big = force_routes || c->always_route || c->expect_big;
  const bool with_retry = force_routes || c->always_route || c->expect_retry;
  Inputs fin;
  const bool filtered = !c->params.use_unknown_cones;
  if (filtered)
    if (int rc = launch_filter(c, q, in_, &fin)) return rc;
  const Inputs& in = filtered ? fin : in_;
  std::string names = std::string(sort128(c, in) ? "sort_kernel_128" : "sort_kernel") + ",";
  if (c->poison) {
    // (tests) whatever a previous pass, another batch or the allocator left in the slot's buffers is gone: 0xFF bytes = NaNs, -1 indices
    const size_t m = (size_t)in.n_frames;
    (void)hipMemsetAsync(q.d_sort, 0xff, sizeof(SortOut) * m, q.stream);
    (void)hipMemsetAsync(q.d_match, 0xff, sizeof(MatchOut) * m, q.stream);
    (void)hipMemsetAsync(q.d_path, 0xff, sizeof(PathOut) * m, q.stream);
    (void)hipMemsetAsync(q.d_mid, 0xff, sizeof(PathMid) * m, q.stream);
    (void)hipMemsetAsync(q.d_arena, 0xff, sizeof(double) * (size_t)ARENA_DOUBLES * m, q.stream);
    (void)hipMemsetAsync(q.d_result, 0xff, sizeof(fsdp_frame_result) * m, q.stream);
    (void)hipMemsetAsync(q.d_big + 1, 0xff, sizeof(int) * m, q.stream);
    (void)hipMemsetAsync(q.d_retry + 1, 0xff, sizeof(int) * m, q.stream);
  }
  mark(q, t);
  launch_sort(c, q, in);
  if (with_big) {
    mark(q, t);
    if (int rc = launch_sort_big(c, q, in)) return rc;
    names += "sort_big_kernel,";
  }
  mark(q, t);
  launch_match(c, q, in);
  names += "match_kernel<" + std::to_string(MATCH_G) + ">,";
  const bool split = launch_path(c, q, in, t, names, in_flight >= 0 ? in_flight : frames_in_flight(c, in.n_frames, false));
  MarkKind after_path = split ? MARK_PLAIN : MARK_MAIN;  // (the one-kernel path stage is the main kernel: close its bracket)
  if (with_retry) {
    mark(q, t, after_path);
    after_path = MARK_PLAIN;
    launch_path_retry(c, q, in, !force_routes || c->retry_hint > 0);
    names += "path_retry_kernel,";
  }
  mark(q, t, after_path);
  launch_assemble(c, q, in.n_frames, false, q.result_dst, nullptr, nullptr, nullptr, filtered ? q.f_map : nullptr, filtered ? q.f_off : nullptr,
                  q.result_compact);
  names += "assemble_kernel";
  mark(q, t, MARK_LAST);
  c->stage_names = names;
  q.pass_in = &in_;
  q.ran_big = with_big;
  q.ran_retry = with_retry;
  q.unverified = true;
  q.pass_skid = false;
  c->last_n = in.n_frames;
  return 0;
}

static PassTrailer read_trailer(const Work& q, int idx) {
  const PassTrailer* h = q.h_trailer + idx;
  PassTrailer tr;
  tr.seq = __atomic_load_n(&h->seq, __ATOMIC_ACQUIRE);
  tr.n_big = __atomic_load_n(&h->n_big, __ATOMIC_RELAXED);
  tr.n_retry = __atomic_load_n(&h->n_retry, __ATOMIC_RELAXED);
  tr.pad = 0;
  return tr;
}

// The slot's stream is idle: did its most recent pass get the route kernels it needed?  If not, the pass runs again with
// both (same inputs, same slot; the caller copies results afterwards).  Also keeps the expectations up to date.
// *rerun (optional) reports whether the pass was repeated.
static int verify_pass(fsdp_ctx* c, Work& q, bool* rerun = nullptr) {
  if (rerun) *rerun = false;
  if (!q.unverified || q.pass_skid) {
    q.unverified = false;
    return 0;
  }
  q.unverified = false;
  const PassTrailer tr = read_trailer(q, SLOT_QUEUE);
  if (tr.seq != q.seq) {
    c->err = "internal: pass trailer out of date (slot " + std::to_string(q.index) + ")";
    return 2;
  }
  auto track = [](bool needed, bool& expect, int& clean) {
    if (needed) {
      expect = true;
      clean = 0;
    } else if (expect && ++clean >= ROUTE_DECAY) {
      expect = false;
      clean = 0;
    }
  };
  c->retry_hint = std::max(tr.n_retry, c->retry_hint - (c->retry_hint + 7) / 8);
  if (q.pass_in == &c->res) {  // the resident batch: from now on its passes carry exactly the routes it needs
    c->expect_big = tr.n_big > 0;
    c->expect_retry = tr.n_retry > 0;
    c->clean_big = c->clean_retry = 0;
    c->res_checked = true;
  } else {
    track(tr.n_big > 0, c->expect_big, c->clean_big);
    track(tr.n_retry > 0, c->expect_retry, c->clean_retry);
  }
  if ((tr.n_big > 0 && !q.ran_big) || (tr.n_retry > 0 && !q.ran_retry)) {
    c->reruns++;
    if (int rc = launch_pass(c, q, *q.pass_in, nullptr, true)) return rc;
    HIP_TRY(c, hipStreamSynchronize(q.stream));
    HIP_TRY(c, hipGetLastError());
    q.unverified = false;
    if (rerun) *rerun = true;
  }
  return 0;
}

// wait for every pass in flight (all slots) and settle their routes
static int flush_skid(fsdp_ctx* c);
static int sync_all(fsdp_ctx* c) {
  if (int rc = flush_skid(c)) return rc;
  for (int i = 0; i < FSDP_MAX_OVERLAP; i++) {
    Work& w = c->slot[i];
    if (!w.stream) continue;
    HIP_TRY(c, hipStreamSynchronize(w.stream));
    if (int rc = verify_pass(c, w)) return rc;  // (a ticket's pass is settled by its fsdp_collect: it never sets `unverified`)
  }
  return 0;
}

static int ensure_slots(fsdp_ctx* c, int n_frames) {
  for (int i = 0; i < c->overlap; i++)
    if (int rc = ensure_work(c, c->slot[i], n_frames)) return rc;
  return 0;
}

static int busy_error(fsdp_ctx* c, const char* who) {
  c->err = std::string(who) + ": " + std::to_string(c->outstanding) + " ticket(s) of fsdp_submit not collected yet (fsdp_collect them first)";
  return 1;
}

static void assemble(const SortOut* s, const MatchOut* m, const PathOut* p, fsdp_frame_result* r) {
  // stage-level entry points: r may already hold fields from earlier stages when only part of the pipeline ran
  if (s) {
    r->status = s->status;
    r->n_left = s->n_left;
    r->n_right = s->n_right;
    memcpy(r->left_idx, s->left_idx, sizeof(r->left_idx));
    memcpy(r->right_idx, s->right_idx, sizeof(r->right_idx));
    r->n_configs_left = s->n_configs_left;
    r->n_configs_right = s->n_configs_right;
    memcpy(r->first_k_left, s->first_k_left, sizeof(r->first_k_left));
    memcpy(r->first_k_right, s->first_k_right, sizeof(r->first_k_right));
    r->best_cost_left = s->best_cost_left;
    r->best_cost_right = s->best_cost_right;
  }
  if (m) {
    if (m->status != 0) r->status = m->status;
    r->n_left_v = m->n_left_v;
    r->n_right_v = m->n_right_v;
    memcpy(r->left_v, m->left_v, sizeof(r->left_v));
    memcpy(r->right_v, m->right_v, sizeof(r->right_v));
    memcpy(r->l2r, m->l2r, sizeof(r->l2r));
    memcpy(r->r2l, m->r2l, sizeof(r->r2l));
  }
  if (p) {
    if (p->status != 0) r->status = p->status;
    memcpy(r->path, p->path, sizeof(r->path));
    r->path_fallback = p->fallback;
    r->n_dense = p->n_dense;
  }
}

// validate a batch description; fills max_cones
static int check_batch(fsdp_ctx* c, int n_frames, const int32_t* off, const double* cones, const double* poses, size_t* total, int* max_cones) {
  if (n_frames < 0 || (n_frames > 0 && (!off || !poses))) {
    c->err = "batch: NULL offsets / poses";
    return 1;
  }
  // cone_offsets[0] = b >= 0: the batch's cones are the rows [b, cone_offsets[n_frames]) of cones_xyt — a slice of a larger
  // batch is handed over by pointing at its offsets, without rebasing or copying anything (include/fsdp.h)
  if (n_frames > 0 && off[0] < 0) {
    c->err = "cone_offsets[0] must be >= 0";
    return 1;
  }
  *total = n_frames > 0 ? (size_t)(off[n_frames] - off[0]) : 0;
  *max_cones = 0;
  if (n_frames > 0 && off[n_frames] < off[0]) {
    c->err = "cone_offsets must be non-decreasing";
    return 1;
  }
  for (int i = 0; i < n_frames; i++) {
    const int d = off[i + 1] - off[i];
    if (d < 0) {
      c->err = "cone_offsets must be non-decreasing";
      return 1;
    }
    *max_cones = std::max(*max_cones, d);
  }
  if (*total > 0 && !cones) {
    c->err = "cones_xyt is NULL";
    return 1;
  }
  return 0;
}

// host -> device of a batch on `stream` (asynchronous for page-locked sources)
static int upload_inputs(fsdp_ctx* c, Inputs& in, hipStream_t stream, int n_frames, const int32_t* off, const double* cones, const double* poses,
                         const double* prev, size_t total, int max_cones) {
  if (int rc = ensure_inputs(c, in, n_frames > 0 ? n_frames : 1, total, prev != nullptr)) return rc;
  in.n_frames = n_frames;
  in.max_cones = max_cones;
  in.use_prev = prev != nullptr;
  if (n_frames == 0) return 0;
  (void)rebase_batch(in, n_frames, off, cones);
  HIP_TRY(c, hipMemcpyAsync(in.d_off, off, sizeof(int32_t) * ((size_t)n_frames + 1), hipMemcpyHostToDevice, stream));
  if (total) HIP_TRY(c, hipMemcpyAsync(in.d_cones, cones, sizeof(double) * 3 * total, hipMemcpyHostToDevice, stream));
  HIP_TRY(c, hipMemcpyAsync(in.d_poses, poses, sizeof(double) * 4 * (size_t)n_frames, hipMemcpyHostToDevice, stream));
  if (prev) HIP_TRY(c, hipMemcpyAsync(in.d_prev, prev, sizeof(double) * PATH_POINTS * 4 * (size_t)n_frames, hipMemcpyHostToDevice, stream));
  return 0;
}

// the same through stage_in_kernel: every source is page-locked host memory
static int stage_inputs(fsdp_ctx* c, Inputs& in, hipStream_t stream, int n_frames, const int32_t* off, const double* cones, const double* poses,
                        const double* prev, size_t total, int max_cones) {
  if (int rc = ensure_inputs(c, in, n_frames > 0 ? n_frames : 1, total, prev != nullptr)) return rc;
  in.n_frames = n_frames;
  in.max_cones = max_cones;
  in.use_prev = prev != nullptr;
  if (n_frames == 0) return 0;
  CopySegs S;
  S.n = 0;
  // a slice of a larger batch (cone_offsets[0] = b > 0): the kernel subtracts b from the offsets on their way in and the cones
  // are read from row b — nothing is rebased or copied on the host, nothing pageable is handed to the runtime (round-5 advisor:
  // a pageable copy made the host wait for the stream's earlier work)
  S.rebase = off[0];
  S.seg[S.n++] = CopySeg{device_view(off), in.d_off, sizeof(int32_t) * ((unsigned long long)n_frames + 1)};
  if (total) S.seg[S.n++] = CopySeg{device_view(cones + 3 * (size_t)off[0]), in.d_cones, sizeof(double) * 3ull * total};
  S.seg[S.n++] = CopySeg{device_view(poses), in.d_poses, sizeof(double) * 4ull * (unsigned long long)n_frames};
  if (prev) S.seg[S.n++] = CopySeg{device_view(prev), in.d_prev, sizeof(double) * PATH_POINTS * 4ull * (unsigned long long)n_frames};
  hipLaunchKernelGGL(stage_in_kernel, dim3(256), dim3(256), 0, stream, S);
  return 0;
}

extern "C" {

const char* fsdp_version(void) { return "fsdp-hip 0.3 (gfx950)"; }
int fsdp_result_size(void) { return (int)sizeof(fsdp_frame_result); }
void fsdp_shapes(int32_t* out4) {
  out4[0] = FSDP_MAX_LEN;
  out4[1] = FSDP_MAX_NEIGHBORS;
  out4[2] = FSDP_MAX_MATCH;
  out4[3] = FSDP_PATH_POINTS;
}

int fsdp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* fsdp_last_error(const fsdp_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

void fsdp_default_params(fsdp_params* p) {
  if (!p) return;
  // fsd_path_planning/config.py:33-41 (sorting), :48 (fitting), :55-59 (path), :124-129 + full_pipeline.py:65 (matching)
  p->max_n_neighbors = 5;
  p->max_dist = 6.5;
  p->max_dist_to_first = 6.0;
  p->max_length = 12;
  p->threshold_directional_angle = 40 * FSDP_DEG;  // np.deg2rad(40)
  p->threshold_absolute_angle = 65 * FSDP_DEG;
  p->use_unknown_cones = 1;
  p->smoothing = 0.2;
  p->predict_every = 0.1;
  p->max_deg = 3;
  p->maximal_distance_for_valid_path = 5;
  p->mpc_path_length = 20;
  p->mpc_prediction_horizon = 40;
  p->min_track_width = 3;
  p->max_search_range = 5;
  p->max_search_angle = 50 * FSDP_DEG;
  p->matches_should_be_monotonic = 0;
}

// what the kernels can take: structural parameters within the compiled capacities, the fixed ones at their values
static const char* check_params(const fsdp_params& p) {
#define FSDP_STR2(x) #x
#define FSDP_STR(x) FSDP_STR2(x)
  if (p.max_n_neighbors < 1 || p.max_n_neighbors > KNN)
    return "max_n_neighbors must be in 1.." FSDP_STR(FSDP_MAX_NEIGHBORS) " (this build's shapes; the wide build takes 8: include/fsdp.h)";
  if (p.max_length < 3 || p.max_length > MAX_LEN)
    return "max_length must be in 3.." FSDP_STR(FSDP_MAX_LEN) " (this build's shapes; the wide build takes 16: include/fsdp.h)";
  if (!(p.max_dist > 0) || !(p.max_dist_to_first > 0)) return "max_dist / max_dist_to_first must be positive";
  if (!(p.smoothing > 0) || !(p.predict_every > 0)) return "smoothing / predict_every must be positive";
  if (!(p.mpc_path_length > 0) || !(p.maximal_distance_for_valid_path >= 0)) return "mpc_path_length must be positive";
  if (!(p.min_track_width > 0) || !(p.max_search_range > 0)) return "min_track_width / max_search_range must be positive";
  if (p.max_deg < 1 || p.max_deg > 3) return "max_deg must be in 1..3";
  if (p.mpc_prediction_horizon < 1 || p.mpc_prediction_horizon > FSDP_PATH_POINTS)
    return "mpc_prediction_horizon must be in 1.." FSDP_STR(FSDP_PATH_POINTS) " (rows of a result path in this build; the wide build holds 64: include/fsdp.h)";
  // the dense path update (fit #1 evaluated every predict_every over <= ~80 m) must fit the working polyline
  if (p.predict_every < 0.05) return "predict_every below 0.05 exceeds the working polyline capacity";
  // the refit is evaluated every predict_every up to 1.5 * mpc_path_length (core_calculate_path.py:248-251) into the same
  // polyline; the extension may add 50 points more (:301-331)
  if (std::ceil(p.mpc_path_length * 1.5 / p.predict_every) + 51 > PATH_CAP)
    return "mpc_path_length * 1.5 / predict_every exceeds the working polyline capacity (1408 points)";
  return nullptr;
}

int fsdp_create(int device, int mission, const fsdp_params* params, fsdp_ctx** out) {
  *out = nullptr;
  fsdp_params pp;
  fsdp_default_params(&pp);
  if (params) pp = *params;
  if (const char* why = check_params(pp)) {
    g_create_error = std::string("fsdp_create: ") + why;
    return 1;
  }
  int n = fsdp_device_count();
  if (n <= 0) {
    g_create_error = "no HIP device visible (libfsdp_hip.so has no CPU fallback)";
    return 1;
  }
  if (device < 0 || device >= n) {
    g_create_error = "device index out of range";
    return 1;
  }
  // utils/mission_types.py:11-25: acceleration = 1, skidpad = 2, ebs_test = 5 use a relocalizer (full_pipeline.py:46-50).
  // Skidpad state lives on the device (fsdp_skidpad_*).  The acceleration relocalizer is a one-off line fit per planner
  // that the host does (acceleration.py, explicit seed); the device side of those two missions is the ordinary path
  // stage with fsdp_set_global_path and empty cone lists.
  fsdp_ctx* c = new fsdp_ctx();
  c->device = device;
  c->mission = mission;
  for (int i = 0; i < FSDP_MAX_OVERLAP; i++) c->slot[i].index = i;
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->slot[0].stream, hipStreamNonBlocking);
  c->stream = c->slot[0].stream;
  for (int i = 0; i < 8 && e == hipSuccess; i++) e = hipEventCreate(&c->ev[i]);
  c->params.max_n_neighbors = pp.max_n_neighbors;
  c->params.max_length = pp.max_length;
  c->params.max_dist = pp.max_dist;
  c->params.max_dist_to_first = pp.max_dist_to_first;
  c->params.threshold_directional_angle = pp.threshold_directional_angle;
  c->params.threshold_absolute_angle = pp.threshold_absolute_angle;
  c->params.min_track_width = pp.min_track_width;
  c->params.max_search_range = pp.max_search_range;
  c->params.max_search_angle = pp.max_search_angle;
  c->params.smoothing = pp.smoothing;
  c->params.predict_every = pp.predict_every;
  c->params.maximal_distance_for_valid_path = pp.maximal_distance_for_valid_path;
  c->params.mpc_path_length = pp.mpc_path_length;
  c->params.max_deg = pp.max_deg;
  c->params.horizon = pp.mpc_prediction_horizon;
  c->params.matches_should_be_monotonic = pp.matches_should_be_monotonic ? 1 : 0;
  c->params.use_unknown_cones = pp.use_unknown_cones ? 1 : 0;
  c->params.retry_pack_min = 512;
  c->params.centers_cap = 0;
  c->params.centers = nullptr;
  c->params.n_centers = nullptr;
  if (e == hipSuccess) e = hipMalloc(&c->d_params, sizeof(Params));
  if (e == hipSuccess) e = hipMemcpyAsync(c->d_params, &c->params, sizeof(Params), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMalloc(&c->d_default_path, sizeof(double) * PATH_POINTS * 4);
  if (e != hipSuccess) {
    g_create_error = std::string("fsdp_create: ") + hipGetErrorString(e);
    delete c;
    return 2;
  }
  // constant initial previous path: almost-straight chord (path_calculator_helpers.py:26-68) fitted and
  // parameterized on the device (core_calculate_path.py:103-121)
  {
    double chord[CHORD_POINTS][2];
    default_chord_points(chord);
    double*& d_chord = c->d_chord;
    double* d_arena0 = nullptr;
    e = hipMalloc(&d_chord, sizeof(chord));
    if (e == hipSuccess) e = hipMalloc(&d_arena0, sizeof(double) * ARENA_DOUBLES);
    if (e == hipSuccess) e = hipMemcpyAsync(d_chord, chord, sizeof(chord), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(default_path_kernel, dim3(1), dim3(WAVE), 0, c->stream, d_chord, d_arena0, c->d_default_path, c->d_params);
      e = hipStreamSynchronize(c->stream);
    }
    if (d_arena0) (void)hipFree(d_arena0);
    if (e != hipSuccess) {
      g_create_error = std::string("fsdp_create(default path): ") + hipGetErrorString(e);
      delete c;
      return 2;
    }
  }
  *out = c;
  return 0;
}

void fsdp_destroy(fsdp_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  for (int i = 0; i < FSDP_MAX_OVERLAP; i++)
    if (c->slot[i].stream) (void)hipStreamSynchronize(c->slot[i].stream);
  (void)fsdp_comm_destroy(c);
  free_inputs(c->res);
  (void)hipFree(c->d_gpath);
  (void)hipFree(c->d_chord);
  (void)hipFree(c->d_table);
  (void)hipFree(c->d_noise);
  (void)hipFree(c->d_skid);
  for (hipEvent_t e : c->skid_group_ev) (void)hipEventDestroy(e);
  (void)hipFree(c->d_skid_backup);
  (void)hipFree(c->d_skid_sync);
  (void)hipFree(c->d_g_arena);
  (void)hipFree(c->d_g_mid);
  (void)hipFree(c->d_g_out);
  (void)hipFree(c->d_g_retry);
  (void)hipFree(c->d_g_sel);
  (void)hipFree(c->d_default_path);
  (void)hipFree(c->d_params);
  if (c->h_sort) (void)hipHostFree(c->h_sort);
  if (c->h_match) (void)hipHostFree(c->h_match);
  if (c->h_path) (void)hipHostFree(c->h_path);
  for (int i = 0; i < FSDP_MAX_OVERLAP; i++) {
    Work& w = c->slot[i];
    free_work(w);
    if (w.stream) (void)hipStreamDestroy(w.stream);
  }
  for (hipEvent_t e : c->tev) (void)hipEventDestroy(e);
  (void)hipFree(c->d_kclock);
  for (int i = 0; i < 8; i++)
    if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
  delete c;
}

// ---- page-locked host memory for the asynchronous entry points ----------------------------------------------------------
void* fsdp_host_alloc(size_t bytes) {
  void* p = nullptr;
  // portable + mapped: a buffer allocated while one GPU is current is read and written by the kernels of any context's GPU
  // (multi.py drives every GPU of a node from one process and stages all shards in buffers of this kind)
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}
void fsdp_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}
int fsdp_host_register(void* p, size_t bytes) {
  if (!p || !bytes) return 1;
  if (hipHostRegister(p, bytes, hipHostRegisterPortable | hipHostRegisterMapped) != hipSuccess) {
    (void)hipGetLastError();
    return 2;
  }
  return 0;
}
int fsdp_host_unregister(void* p) {
  if (!p) return 1;
  if (hipHostUnregister(p) != hipSuccess) {
    (void)hipGetLastError();
    return 2;
  }
  return 0;
}

int fsdp_host_is_pinned(const void* p, size_t bytes) { return (p && is_pinned(p, bytes)) ? 1 : 0; }

// ---- the resident batch ------------------------------------------------------------------------------------------------
int fsdp_upload(fsdp_ctx* c, int n_frames, const int32_t* off, const double* cones, const double* poses) {
  if (!c) return 1;
  if (c->outstanding) return busy_error(c, "fsdp_upload");
  HIP_TRY(c, hipSetDevice(c->device));
  size_t total;
  int max_cones;
  if (int rc = check_batch(c, n_frames, off, cones, poses, &total, &max_cones)) return rc;
  if (int rc = sync_all(c)) return rc;  // passes in flight still read the old inputs
  if (int rc = ensure_slots(c, n_frames > 0 ? n_frames : 1)) return rc;
  if (int rc = upload_inputs(c, c->res, c->stream, n_frames, off, cones, poses, nullptr, total, max_cones)) return rc;
  c->resident = true;
  c->res_checked = false;
  c->last_slot = 0;
  c->last_n = n_frames;
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // the other slots' streams read these inputs too; the caller's buffers are free again
  return 0;
}

int fsdp_set_overlap(fsdp_ctx* c, int depth) {
  if (!c || depth < 1 || depth > FSDP_MAX_OVERLAP) return 1;
  if (c->outstanding) return busy_error(c, "fsdp_set_overlap");
  HIP_TRY(c, hipSetDevice(c->device));
  if (int rc = sync_all(c)) return rc;
  // a smaller depth gives the slots beyond it back — their buffers AND their streams: every stream holds one of the GPU's hardware
  // queues, which all contexts and processes on the device share (a MultiPlanner next to a context that once ran twenty passes in
  // flight planned 3.9 instead of 5.1 M frames/s until those queues were released)
  for (int i = depth; i < FSDP_MAX_OVERLAP; i++) {
    Work& w = c->slot[i];
    if (!w.stream && !w.d_sort && !w.h_trailer) continue;
    free_work(w);
    if (w.stream) (void)hipStreamDestroy(w.stream);
    const int idx = w.index;
    w = Work();
    w.index = idx;
  }
  c->overlap = depth;
  c->turn = 0;
  c->last_slot = 0;
  c->last_ticket_slot = -1;
  for (bool& p : c->primed) p = false;  // the kernels of a pass depend on the frames in flight (launch_path)
  return ensure_slots(c, std::max(1, c->slot[0].cap_frames));
}

int fsdp_run(fsdp_ctx* c) {
  if (!c) return 1;
  if (c->outstanding) return busy_error(c, "fsdp_run");
  if (!c->resident) {
    c->err = "fsdp_run: no resident batch (fsdp_upload first)";
    return 1;
  }
  if (c->res.n_frames == 0) return 0;
  HIP_TRY(c, hipSetDevice(c->device));
  const int si = (c->overlap > 1) ? (int)(c->turn++ % (unsigned)c->overlap) : 0;
  Work& q = c->slot[si];
  // the slot's previous pass may still need its routes: settle it before its buffers are reused (passes over a checked
  // resident batch carry exactly the routes they need: nothing to settle, no host wait)
  if (q.unverified && !(c->res_checked && q.pass_in == &c->res)) {
    HIP_TRY(c, hipStreamSynchronize(q.stream));
    if (int rc = verify_pass(c, q)) return rc;
  }
  c->last_slot = si;
  if (int rc = launch_pass(c, q, c->res)) return rc;
  HIP_TRY(c, hipGetLastError());
  return 0;
}

int fsdp_resident_frames(const fsdp_ctx* c) { return c ? c->last_n : 0; }

int fsdp_sync(fsdp_ctx* c) {
  if (!c) return 1;
  HIP_TRY(c, hipSetDevice(c->device));
  return sync_all(c);
}

int fsdp_download(fsdp_ctx* c, fsdp_frame_result* results) {
  if (!c || (c->last_n > 0 && !results)) return 1;
  if (c->outstanding) return busy_error(c, "fsdp_download");
  const int n = c->last_n;
  if (n == 0) return 0;
  HIP_TRY(c, hipSetDevice(c->device));
  if (int rc = sync_all(c)) return rc;
  Work& q = c->slot[c->last_slot];  // the most recent pass
  HIP_TRY(c, hipMemcpyAsync(results, q.d_result, sizeof(fsdp_frame_result) * (size_t)n, hipMemcpyDeviceToHost, q.stream));
  HIP_TRY(c, hipStreamSynchronize(q.stream));
  return 0;
}

int fsdp_set_previous_paths(fsdp_ctx* c, const double* prev_paths) {
  if (!c) return 1;
  if (c->outstanding) return busy_error(c, "fsdp_set_previous_paths");
  HIP_TRY(c, hipSetDevice(c->device));
  if (!prev_paths) {
    c->res.use_prev = false;
    return 0;
  }
  if (!c->resident || c->res.n_frames <= 0) {
    c->err = "fsdp_set_previous_paths: upload a batch first";
    return 1;
  }
  if (int rc = sync_all(c)) return rc;
  if (int rc = ensure_inputs(c, c->res, c->res.n_frames, c->res.cap_cones, true)) return rc;
  HIP_TRY(c, copy_sync(c, c->res.d_prev, prev_paths, sizeof(double) * PATH_POINTS * 4 * (size_t)c->res.n_frames, hipMemcpyHostToDevice));
  c->res.use_prev = true;
  c->res_checked = false;
  return 0;
}

int fsdp_set_global_path(fsdp_ctx* c, const double* xy, int n) {
  if (!c || n < 0 || (n > 0 && !xy)) return 1;
  if (c->outstanding) return busy_error(c, "fsdp_set_global_path");
  HIP_TRY(c, hipSetDevice(c->device));
  if (int rc = sync_all(c)) return rc;
  if (c->d_gpath) (void)hipFree(c->d_gpath);
  c->d_gpath = nullptr;
  c->n_gpath = 0;
  c->res_checked = false;
  if (n == 0) return 0;
  HIP_TRY(c, hipMalloc(&c->d_gpath, sizeof(double) * 2 * (size_t)n));
  HIP_TRY(c, copy_sync(c, c->d_gpath, xy, sizeof(double) * 2 * (size_t)n, hipMemcpyHostToDevice));
  c->n_gpath = n;
  return 0;
}

// ---- streams of batches: submit / collect ------------------------------------------------------------------------------
// One batch per ticket.  Ticket t goes to slot t % depth and is queued on that slot's stream behind the slot's previous
// ticket (up to SLOT_QUEUE per slot): host -> device of the batch, the kernels of its pass and device -> host of its results
// are enqueued by fsdp_submit, which returns at once.  Page-locked buffers (fsdp_host_alloc / fsdp_host_register) are read
// and written by kernels of the stream itself (stage_in_kernel, assemble_kernel); pageable ones work, but their copies are
// staged by the runtime and block the caller.
static Work::Ticket* find_ticket(fsdp_ctx* c, long long ticket, Work** slot) {
  if (ticket < 0) return nullptr;
  for (int i = 0; i < FSDP_MAX_OVERLAP; i++)
    for (Work::Ticket& t : c->slot[i].tk)
      if (t.id == ticket) {
        if (slot) *slot = &c->slot[i];
        return &t;
      }
  return nullptr;
}

// enqueue ticket t's batch on slot q: inputs, the pass with its results' way back, the ticket's event
// in_flight: frames on the GPU next to this batch, its own included (< 0: counted from the outstanding tickets)
// A pageable batch of up to SMALL_BATCH_BYTES is packed into the ticket's own page-locked block by the host (a memcpy of a few KB)
// and then treated like any page-locked batch: the sorting kernel reads it over PCIe, no copy command is issued at all — three or
// four hipMemcpyAsync calls from pageable memory cost a single-frame call ~30 us of its ~860.
constexpr size_t SMALL_BATCH_BYTES = 256 * 1024;

static int enqueue_ticket(fsdp_ctx* c, Work& q, Work::Ticket& t, bool force_routes, long long in_flight = -1) {
  const int n = t.n;
  // a bigger batch than the slot has seen: its buffers are replaced — not under the feet of the passes queued on the stream
  if (n > q.cap_frames || n > q.in.cap_frames || t.total > q.in.cap_cones || (t.prev && n > q.in.cap_prev)) HIP_TRY(c, hipStreamSynchronize(q.stream));
  if (int rc = ensure_work(c, q, n > 0 ? n : 1)) return rc;
  const int32_t* off = t.off;
  const double* cones = t.cones;
  const double* poses = t.poses;
  const double* prev = t.prev;
  bool in_pinned = n > 0 && is_pinned(off, sizeof(int32_t) * ((size_t)n + 1)) && is_pinned(poses, sizeof(double) * 4 * (size_t)n) &&
                   (t.total == 0 || is_pinned(cones + 3 * (size_t)off[0], sizeof(double) * 3 * t.total)) &&
                   (!prev || is_pinned(prev, sizeof(double) * PATH_POINTS * 4 * (size_t)n));
  const size_t off_bytes = (sizeof(int32_t) * ((size_t)n + 1) + 15) & ~(size_t)15, cone_bytes = sizeof(double) * 3 * t.total,
               pose_bytes = sizeof(double) * 4 * (size_t)n, prev_bytes = prev ? sizeof(double) * PATH_POINTS * 4 * (size_t)n : 0;
  const size_t in_bytes = off_bytes + cone_bytes + pose_bytes + prev_bytes;
  if (!in_pinned && n > 0 && in_bytes <= SMALL_BATCH_BYTES) {
    if (in_bytes > t.cap_in) {
      // (the block's previous user — this ticket entry's previous batch — was collected before the entry was handed out again)
      if (t.h_in) (void)hipHostFree(t.h_in);
      t.h_in = nullptr;
      t.cap_in = 0;
      const size_t want = std::max(in_bytes, (size_t)16384);
      HIP_TRY(c, hipHostMalloc((void**)&t.h_in, want, hipHostMallocMapped));
      t.cap_in = want;
    }
    int32_t* so = (int32_t*)t.h_in;
    double* sc = (double*)(t.h_in + off_bytes);
    double* sp = (double*)(t.h_in + off_bytes + cone_bytes);
    double* sv = (double*)(t.h_in + off_bytes + cone_bytes + pose_bytes);
    const int32_t b = off[0];
    for (int i = 0; i <= n; i++) so[i] = off[i] - b;
    if (cone_bytes) memcpy(sc, cones + 3 * (size_t)b, cone_bytes);
    memcpy(sp, poses, pose_bytes);
    if (prev) memcpy(sv, prev, prev_bytes);
    off = so;
    cones = sc;
    poses = sp;
    prev = prev ? sv : nullptr;
    in_pinned = true;
  }
  const bool out_pinned = n > 0 && is_pinned(t.user_results, t.rec_bytes() * (size_t)n);
  q.in.h_off = nullptr;
  if (in_pinned && c->params.use_unknown_cones) {
    // the pass's sorting kernel reads the batch from the page-locked buffers and leaves the device copies (StageIn)
    if (int rc = ensure_inputs(c, q.in, n, t.total, prev != nullptr)) return rc;
    q.in.n_frames = n;
    q.in.max_cones = t.max_cones;
    q.in.use_prev = prev != nullptr;
    q.in.h_off = (const int32_t*)device_view(off);
    q.in.h_base = off[0];
    // (the view of the slice's first row, addressed by offsets relative to h_base; never read when total = 0)
    q.in.h_cones = t.total ? (const double*)device_view(cones + 3 * (size_t)off[0]) : (const double*)device_view(poses);
    q.in.h_poses = (const double*)device_view(poses);
    q.in.h_prev = prev ? (const double*)device_view(prev) : nullptr;
  } else if (in_pinned) {
    if (int rc = stage_inputs(c, q.in, q.stream, n, off, cones, poses, prev, t.total, t.max_cones)) return rc;
  } else if (int rc = upload_inputs(c, q.in, q.stream, n, off, cones, poses, prev, t.total, t.max_cones)) {
    return rc;
  }
  t.via_stage = false;
  if (n > 0) {
    // Results always leave the GPU inside the pass's last kernel, written over PCIe into page-locked memory: the caller's own buffer,
    // or — for a pageable one — the ticket's block, which fsdp_collect copies out (no copy command on the stream either way).
    fsdp_frame_result* dst = t.user_results;
    if (!out_pinned) {
      if (n > t.cap_stage) {
        if (t.h_stage) (void)hipHostFree(t.h_stage);
        t.h_stage = nullptr;
        t.cap_stage = 0;
        const int want = std::max(n, 64);
        HIP_TRY(c, hipHostMalloc((void**)&t.h_stage, sizeof(fsdp_frame_result) * (size_t)want, hipHostMallocMapped));
        t.cap_stage = want;
      }
      dst = t.h_stage;
      t.via_stage = true;
    }
    q.result_dst = (fsdp_frame_result*)device_view(dst);
    if (!q.result_dst) {
      c->err = "internal: result block is not mapped into the device's address space";
      return 2;
    }
    q.result_compact = t.compact;
    q.trailer_idx = (int)(&t - q.tk);  // the ticket's own trailer
    const int rc = launch_pass(c, q, q.in, nullptr, force_routes, in_flight >= 0 ? in_flight : frames_in_flight(c, n, true));
    q.result_dst = nullptr;
    q.result_compact = false;
    q.trailer_idx = SLOT_QUEUE;
    q.in.h_off = nullptr;  // (the views served that one sorting launch)
    q.unverified = false;  // settled by fsdp_collect through the ticket
    if (rc) return rc;
    t.seq = q.seq;
    t.ran_big = q.ran_big;
    t.ran_retry = q.ran_retry;
    HIP_TRY(c, hipGetLastError());
  }
  if (!t.done) HIP_TRY(c, hipEventCreateWithFlags(&t.done, hipEventDisableTiming));
  HIP_TRY(c, hipEventRecord(t.done, q.stream));
  return 0;
}

static int submit_impl(fsdp_ctx* c, int n_frames, const int32_t* off, const double* cones, const double* poses, const double* prev_paths,
                       fsdp_frame_result* results, long long* ticket, bool compact) {
  if (!c || !ticket) return 1;
  *ticket = -1;
  if (c->mission == 2) {
    c->err = "fsdp_submit: a skidpad context plans through fsdp_skidpad_submit";
    return 1;
  }
  if (n_frames > 0 && !results) {
    c->err = "fsdp_submit: results is NULL";
    return 1;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  size_t total;
  int max_cones;
  if (int rc = check_batch(c, n_frames, off, cones, poses, &total, &max_cones)) return rc;
  // the slot with the fewest tickets queued, starting from the one after the previous ticket's (in-order traffic: round robin)
  int si = -1, best = SLOT_QUEUE;
  long long oldest = -1;
  for (int k = 0; k < c->overlap; k++) {
    const int i = (c->last_ticket_slot + 1 + k) % c->overlap;
    int cnt = 0;
    for (const Work::Ticket& e : c->slot[i].tk) {
      if (e.id < 0) continue;
      cnt++;
      if (oldest < 0 || e.id < oldest) oldest = e.id;
    }
    if (cnt < best) {
      best = cnt;
      si = i;
    }
  }
  if (si < 0) {
    c->err = "fsdp_submit: " + std::to_string(c->outstanding) + " tickets outstanding (" + std::to_string(SLOT_QUEUE) + " per slot, " +
             std::to_string(c->overlap) + " slots): collect one first, e.g. ticket " + std::to_string(oldest);
    return 4;
  }
  c->last_ticket_slot = si;
  Work& q = c->slot[si];
  Work::Ticket* t = nullptr;
  for (Work::Ticket& e : q.tk)
    if (e.id < 0 && !t) t = &e;
  if (!q.stream) {
    if (int rc = ensure_work(c, q, n_frames > 0 ? n_frames : 1)) return rc;
  }
  if (q.unverified) {  // an fsdp_run pass nobody waited for
    HIP_TRY(c, hipStreamSynchronize(q.stream));
    if (int rc = verify_pass(c, q)) return rc;
  }
  t->n = n_frames;
  t->skid = false;
  t->off = off;
  t->cones = cones;
  t->poses = poses;
  t->prev = prev_paths;
  t->total = total;
  t->max_cones = max_cones;
  t->user_results = results;
  t->user_info = nullptr;
  t->compact = compact;
  if (int rc = enqueue_ticket(c, q, *t, false)) {
    // Part of the batch may already be queued on the slot's stream — kernels that read the caller's buffers or write his
    // page-locked results — and no ticket goes out that he could wait on: wait here, so that an error return means the
    // buffers are his again (round-3 advisor).
    (void)hipStreamSynchronize(q.stream);
    (void)hipGetLastError();
    t->user_results = nullptr;
    t->compact = false;
    return rc;
  }
  t->id = c->next_ticket++;
  c->outstanding++;
  c->last_slot = si;
  *ticket = t->id;
  return 0;
}
int fsdp_submit(fsdp_ctx* c, int n_frames, const int32_t* off, const double* cones, const double* poses, const double* prev_paths,
                fsdp_frame_result* results, long long* ticket) {
  return submit_impl(c, n_frames, off, cones, poses, prev_paths, results, ticket, false);
}
int fsdp_submit_compact(fsdp_ctx* c, int n_frames, const int32_t* off, const double* cones, const double* poses, const double* prev_paths,
                        fsdp_compact_result* results, long long* ticket) {
  return submit_impl(c, n_frames, off, cones, poses, prev_paths, (fsdp_frame_result*)results, ticket, true);
}

// 1: fsdp_collect will not block (unless the pass has to be repeated with a route kernel); 0: still running; < 0: unknown ticket
int fsdp_ticket_done(fsdp_ctx* c, long long ticket) {
  if (!c) return -1;
  Work::Ticket* t = find_ticket(c, ticket, nullptr);
  if (!t) return -1;
  (void)hipSetDevice(c->device);
  if (t->pending && flush_skid(c)) return -1;
  hipError_t e = hipEventQuery(t->done);
  if (e == hipSuccess) return 1;
  (void)hipGetLastError();
  return 0;
}

int fsdp_collect(fsdp_ctx* c, long long ticket) {
  if (!c) return 1;
  Work* qp = nullptr;
  Work::Ticket* tp = find_ticket(c, ticket, &qp);
  if (!tp) {
    c->err = "fsdp_collect: unknown ticket " + std::to_string(ticket);
    return 1;
  }
  Work& q = *qp;
  Work::Ticket& t = *tp;
  int rc = 0;
  hipError_t e = hipSetDevice(c->device);
  if (t.pending)
    if (int frc = flush_skid(c)) return frc;
  if (e == hipSuccess) e = hipEventSynchronize(t.done);
  if (e != hipSuccess) {
    c->err = std::string("fsdp_collect: ") + hipGetErrorString(e);
    rc = 2;
  }
  const int n = t.n;
  if (rc == 0 && n > 0 && !t.skid) {
    // did the pass get the route kernels it needed?  (its own trailer: later passes of the slot write other ones)
    const PassTrailer tr = read_trailer(q, (int)(&t - q.tk));
    if (tr.seq != t.seq) {
      c->err = "internal: trailer of ticket " + std::to_string(ticket) + " overwritten";
      rc = 2;
    } else {
      auto track = [](bool needed, bool& expect, int& clean) {
        if (needed) {
          expect = true;
          clean = 0;
        } else if (expect && ++clean >= ROUTE_DECAY) {
          expect = false;
          clean = 0;
        }
      };
      track(tr.n_big > 0, c->expect_big, c->clean_big);
      track(tr.n_retry > 0, c->expect_retry, c->clean_retry);
      c->retry_hint = std::max(tr.n_retry, c->retry_hint - (c->retry_hint + 7) / 8);
      if ((tr.n_big > 0 && !t.ran_big) || (tr.n_retry > 0 && !t.ran_retry)) {
        // the whole ticket once more, with both route kernels, behind whatever the slot's stream holds by now (the
        // caller's buffers are still his to leave alone: the batch is read again from them)
        c->reruns++;
        rc = enqueue_ticket(c, q, t, true);
        if (rc != 0) (void)hipStreamSynchronize(q.stream);  // (nothing of the repeated pass is left running over the caller's buffers)
        if (rc == 0 && (e = hipEventSynchronize(t.done)) != hipSuccess) {
          c->err = std::string("fsdp_collect: ") + hipGetErrorString(e);
          rc = 2;
        }
      }
    }
  }
  if (rc == 0 && n > 0) {
    if (t.via_stage) memcpy(t.user_results, t.h_stage, t.rec_bytes() * (size_t)n);
    if (t.user_info && t.h_info) {
      if (!c->skid_all_reloc) {
        bool all = true;
        for (int i = 0; i < n && all; i++) all = t.h_info[i].relocalized != 0;
        c->skid_all_reloc = all;
      }
      for (int i = 0; i < n; i++) {
        t.user_info[i].relocalized = t.h_info[i].relocalized;
        t.user_info[i].index_along_path = t.h_info[i].index_along_path;
        t.user_info[i].translation[0] = t.h_info[i].translation[0];
        t.user_info[i].translation[1] = t.h_info[i].translation[1];
        t.user_info[i].rotation = t.h_info[i].rotation;
      }
    }
  }
  t.id = -1;
  t.user_results = nullptr;
  t.user_info = nullptr;
  t.compact = false;
  c->outstanding--;
  return rc;
}

int fsdp_ticket_capacity(const fsdp_ctx* c) { return c ? c->overlap * (c->mission == 2 ? 1 : SLOT_QUEUE) : 0; }

int fsdp_route_stats(fsdp_ctx* c, int* expect_big, int* expect_retry, long long* reruns) {
  if (!c) return 1;
  if (expect_big) *expect_big = c->expect_big ? 1 : 0;
  if (expect_retry) *expect_retry = c->expect_retry ? 1 : 0;
  if (reruns) *reruns = c->reruns;
  return 0;
}

// ---- blocking calls on host buffers -------------------------------------------------------------------------------------
// A LARGE blocking call is cut into up to PLAN_CHUNKS contiguous chunks, each a ticket on a slot of its own: one chunk's transfers
// (in place for page-locked buffers, staged by the runtime for pageable ones — then the host copies chunk k + 1 while the kernels of
// chunk k run) go under the other chunks' kernels, and the first chunk's results are on their way back while the last one is still
// being planned.  Measured (tools/plan_probe.py, profiles/r06_plan_probe.txt): at 4096 frames chunks only add launches to a pass that is
// one dependent chain anyway (2.39 -> 2.73 ms page-locked, 3.11 -> 3.32 ms pageable); from 16 384 frames they pay for pageable buffers
// (10.2 -> 8.3 ms), at 65 536 for both (36.0 -> 26.6 ms pageable, 14.8 -> 13.4 ms page-locked).  Hence: four chunks from
// PLAN_CHUNK_FROM frames on, none below.  The chunks know they share the GPU (in_flight = the whole batch) and run the kernels the
// whole batch would.  Results do not depend on the chunking.  Option "plan_chunks" = k forces up to k chunks of >= PLAN_CHUNK_MIN frames
// (tests), 1 = never.
constexpr int PLAN_CHUNKS = 4, PLAN_CHUNK_FROM = 16384, PLAN_CHUNK_MIN = 512;

static int plan_blocking(fsdp_ctx* c, int n_frames, const int32_t* off, const double* cones, const double* poses, const double* prev,
                         fsdp_frame_result* results, bool compact) {
  if (!c) return 1;
  if (c->outstanding) return busy_error(c, "fsdp_plan_batch");
  if (n_frames > 0 && !results) return 1;
  if (c->mission == 2) {
    c->err = "fsdp_plan_batch: a skidpad context plans through fsdp_skidpad_step";
    return 1;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  size_t total;
  int max_cones;
  if (int rc = check_batch(c, n_frames, off, cones, poses, &total, &max_cones)) return rc;
  if (int rc = sync_all(c)) return rc;
  c->last_slot = 0;
  c->last_n = n_frames;
  if (n_frames == 0) return 0;
  const int chunks = c->plan_chunks > 0 ? std::max(1, std::min(c->plan_chunks, n_frames / PLAN_CHUNK_MIN)) : (n_frames >= PLAN_CHUNK_FROM ? PLAN_CHUNKS : 1);
  const size_t rec = compact ? sizeof(fsdp_compact_result) : sizeof(fsdp_frame_result);
  long long ids[PLAN_CHUNKS];
  int issued = 0, rc = 0;
  for (int k = 0; k < chunks && rc == 0; k++) {
    const int lo = (int)((long long)n_frames * k / chunks), hi = (int)((long long)n_frames * (k + 1) / chunks);
    Work& q = c->slot[k];  // (slots beyond the overlap depth get their stream here: a chunk is a pass in flight)
    Work::Ticket& t = q.tk[0];
    if ((rc = ensure_work(c, q, hi - lo))) break;
    t.n = hi - lo;
    t.skid = false;
    t.off = off + lo;
    t.cones = cones;
    t.poses = poses + 4 * (size_t)lo;
    t.prev = prev ? prev + (size_t)PATH_POINTS * 4 * (size_t)lo : nullptr;
    t.total = (size_t)(off[hi] - off[lo]);
    t.max_cones = max_cones;
    t.user_results = (fsdp_frame_result*)((char*)results + rec * (size_t)lo);
    t.user_info = nullptr;
    t.compact = compact;
    if ((rc = enqueue_ticket(c, q, t, false, n_frames))) {
      (void)hipStreamSynchronize(q.stream);
      (void)hipGetLastError();
      t.user_results = nullptr;
      t.compact = false;
      break;
    }
    t.id = ids[issued++] = c->next_ticket++;
    c->outstanding++;
    c->last_slot = k;
    c->last_n = hi - lo;
  }
  for (int k = 0; k < issued; k++) {
    const int rck = fsdp_collect(c, ids[k]);  // (every issued chunk is waited for, also after an error: the buffers are the caller's again)
    if (rc == 0) rc = rck;
  }
  c->next_ticket -= issued;  // (the chunks' numbers were internal: the caller's tickets keep counting up from where they were)
  return rc;
}

int fsdp_plan_batch_sequential(fsdp_ctx* c, int n_frames, const int32_t* off, const double* cones, const double* poses,
                               const double* prev_paths, fsdp_frame_result* results) {
  return plan_blocking(c, n_frames, off, cones, poses, prev_paths, results, false);
}

int fsdp_plan_batch(fsdp_ctx* c, int n_frames, const int32_t* off, const double* cones, const double* poses,
                    fsdp_frame_result* results) {
  return plan_blocking(c, n_frames, off, cones, poses, nullptr, results, false);
}

int fsdp_plan_batch_compact(fsdp_ctx* c, int n_frames, const int32_t* off, const double* cones, const double* poses, const double* prev_paths,
                            fsdp_compact_result* results) {
  return plan_blocking(c, n_frames, off, cones, poses, prev_paths, (fsdp_frame_result*)results, true);
}

// Options (include/fsdp.h fsdp_set_option): what tests and measurements pin.  Results never depend on them.
int fsdp_set_option(fsdp_ctx* c, const char* name, long long v) {
  if (!c || !name) return 1;
  if (c->outstanding) return busy_error(c, "fsdp_set_option");
  HIP_TRY(c, hipSetDevice(c->device));
  if (int rc = sync_all(c)) return rc;
  const std::string k(name);
  auto bad = [&]() {
    c->err = "fsdp_set_option: unknown option or value out of range: " + k + " = " + std::to_string(v);
    return 1;
  };
  if (k == "path_mode") {
    if (v < 0 || v > 2) return bad();
    c->force_path_mode = (int)v;
  } else if (k == "pack") {
    if (v < 0 || v > 2) return bad();
    c->force_pack = (int)v;
  } else if (k == "fit_g") {
    if (v != 0 && v != 4 && v != 8) return bad();
    c->fit_g = v == 8 ? 8 : 4;
  } else if (k == "always_route") {
    c->always_route = v != 0;
  } else if (k == "no_sort128") {
    c->no_sort128 = v != 0;
  } else if (k == "retry_pack_min") {
    if (v < 0 || v > 0x7fffffff) return bad();
    c->params.retry_pack_min = v == 0 ? 512 : (int)v;
    HIP_TRY(c, copy_sync(c, c->d_params, &c->params, sizeof(Params), hipMemcpyHostToDevice));
  } else if (k == "poison") {
    c->poison = v != 0;
  } else if (k == "plan_chunks") {
    if (v < 0 || v > PLAN_CHUNKS) return bad();
    c->plan_chunks = (int)v;
  } else if (k == "skid_group") {
    if (v < 0 || v > SKID_GROUP_MAX) return bad();
    c->skid_group_env = (int)v;
  } else if (k == "skid_pack_min") {
    if (v < 0 || v > 0x7fffffff) return bad();
    c->skid_pack_min = v == 0 ? 2048 : (int)v;
  } else {
    return bad();
  }
  for (bool& p : c->primed) p = false;  // (the kernels of a pass may have changed: fsdp_time_reserve warms the slots again)
  c->res_checked = false;
  return 0;
}

// What the link between this GPU and the host carries for page-locked buffers of `bytes` bytes: hipMemcpyAsync host -> device alone,
// device -> host alone, and both directions at once on two streams (GB/s each; measurement only — the ceiling a stream of batches
// is held against, bench.py "streaming.pcie_ceiling_GBps").
int fsdp_pcie_probe(fsdp_ctx* c, size_t bytes, int iters, double* h2d_GBps, double* d2h_GBps, double* both_each_GBps) {
  if (!c || bytes == 0 || iters <= 0) return 1;
  if (c->outstanding) return busy_error(c, "fsdp_pcie_probe");
  HIP_TRY(c, hipSetDevice(c->device));
  if (int rc = sync_all(c)) return rc;
  void *h_up = nullptr, *h_dn = nullptr, *d_up = nullptr, *d_dn = nullptr;
  hipStream_t s2 = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
  auto cleanup = [&]() {
    if (h_up) (void)hipHostFree(h_up);
    if (h_dn) (void)hipHostFree(h_dn);
    (void)hipFree(d_up);
    (void)hipFree(d_dn);
    if (s2) (void)hipStreamDestroy(s2);
    for (hipEvent_t e : {e0, e1, e2})
      if (e) (void)hipEventDestroy(e);
  };
  hipError_t e = hipHostMalloc(&h_up, bytes, hipHostMallocDefault);
  if (e == hipSuccess) e = hipHostMalloc(&h_dn, bytes, hipHostMallocDefault);
  if (e == hipSuccess) e = hipMalloc(&d_up, bytes);
  if (e == hipSuccess) e = hipMalloc(&d_dn, bytes);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  if (e == hipSuccess) e = hipEventCreate(&e2);
  if (e == hipSuccess) {
    memset(h_up, 1, bytes);
    e = hipMemsetAsync(d_dn, 2, bytes, c->stream);
  }
  auto run = [&](bool up, bool dn, double* each) -> hipError_t {
    hipError_t r = hipStreamSynchronize(c->stream);
    if (r == hipSuccess) r = hipStreamSynchronize(s2);
    if (r != hipSuccess) return r;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters && r == hipSuccess; i++) {
      if (up) r = hipMemcpyAsync(d_up, h_up, bytes, hipMemcpyHostToDevice, c->stream);
      if (dn && r == hipSuccess) r = hipMemcpyAsync(h_dn, d_dn, bytes, hipMemcpyDeviceToHost, s2);
    }
    if (r == hipSuccess) r = hipStreamSynchronize(c->stream);
    if (r == hipSuccess) r = hipStreamSynchronize(s2);
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (each) *each = (double)bytes * iters / sec / 1e9;
    return r;
  };
  if (e == hipSuccess) e = run(true, true, nullptr);  // warm
  if (e == hipSuccess) e = run(true, false, h2d_GBps);
  if (e == hipSuccess) e = run(false, true, d2h_GBps);
  if (e == hipSuccess) e = run(true, true, both_each_GBps);
  cleanup();
  HIP_TRY(c, e);
  return 0;
}

// ---- timing of the resident batch ---------------------------------------------------------------------------------------
// MAX_STAGES + 1 events per pass (before every kernel, after the last) + begin / end of the region
constexpr int TIMING_EPP = MAX_STAGES + 1;
static int reserve_timing(fsdp_ctx* c, int iters) {
  const size_t need = (size_t)TIMING_EPP * (size_t)iters + 2;
  while (c->tev.size() < need) {
    hipEvent_t e;
    HIP_TRY(c, hipEventCreate(&e));
    c->tev.push_back(e);
  }
  if (iters > c->kclock_cap) {
    (void)hipFree(c->d_kclock);
    c->d_kclock = nullptr;
    c->kclock_cap = 0;
    HIP_TRY(c, hipMalloc((void**)&c->d_kclock, sizeof(unsigned long long) * 2 * (size_t)iters));
    c->kclock_cap = iters;
  }
  return 0;
}

// one verified pass over the resident batch: afterwards the route expectations are exactly what this batch needs
static int check_resident(fsdp_ctx* c) {
  if (c->res_checked || !c->resident || c->res.n_frames == 0) return 0;
  if (int rc = sync_all(c)) return rc;
  Work& q = c->slot[0];
  if (int rc = launch_pass(c, q, c->res)) return rc;
  HIP_TRY(c, hipStreamSynchronize(q.stream));
  return verify_pass(c, q);  // sets res_checked and the exact expectations
}

int fsdp_time_reserve(fsdp_ctx* c, int iters) {
  if (!c || iters <= 0) return 1;
  if (c->outstanding) return busy_error(c, "fsdp_time_reserve");
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = reserve_timing(c, iters);
  if (rc) return rc;
  // The first launches on a stream pay for its hardware queue and scratch set-up: every slot of the current overlap depth
  // that has not run a pass yet runs one now (the resident batch, results overwritten by the timed passes).
  if (c->resident && c->res.n_frames > 0) {
    if ((rc = check_resident(c))) return rc;
    for (int i = 0; i < c->overlap; i++)
      if (!c->primed[i])
        if ((rc = launch_pass(c, c->slot[i], c->res))) return rc;
    rc = sync_all(c);
    if (rc) return rc;
    HIP_TRY(c, hipGetLastError());
  }
  return 0;
}

int fsdp_time_results(fsdp_ctx* c, float* ms_total, float* ms_stage) {
  if (!c) return 1;
  if (ms_stage)
    for (int k = 0; k < MAX_STAGES; k++) ms_stage[k] = 0;
  if (ms_total) *ms_total = 0;
  if (c->timed_iters <= 0) return 0;
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t need = (size_t)TIMING_EPP * (size_t)c->timed_iters + 2;
  float total = 0;
  HIP_TRY(c, hipEventElapsedTime(&total, c->tev[need - 2], c->tev[need - 1]));
  if (ms_stage)
    for (int it = 0; it < c->timed_iters; it++)
      for (int st = 0; st < c->timed_stages; st++) {
        float t;
        if ((c->tev_recorded[it] >> st & 3u) != 3u) continue;  // a kernel the region did not bracket: its time stays 0
        HIP_TRY(c, hipEventElapsedTime(&t, c->tev[(size_t)TIMING_EPP * (size_t)it + st], c->tev[(size_t)TIMING_EPP * (size_t)it + st + 1]));
        ms_stage[st] += t;
      }
  if (ms_total) *ms_total = total;
  return 0;
}

int fsdp_time_runs(fsdp_ctx* c, int iters, float* ms_total, float* ms_stage) {
  if (!c || iters <= 0) return 1;
  if (c->outstanding) return busy_error(c, "fsdp_time_runs");
  if (ms_stage)
    for (int k = 0; k < MAX_STAGES; k++) ms_stage[k] = 0;
  if (ms_total) *ms_total = 0;
  c->timed_iters = 0;
  if (!c->resident) {
    c->err = "fsdp_time_runs: no resident batch";
    return 1;
  }
  if (c->res.n_frames == 0) return 0;
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = sync_all(c);
  if (rc) return rc;
  if ((rc = check_resident(c))) return rc;  // the timed passes carry exactly the route kernels this batch needs
  // passes rotate through the slots when overlap is on and are NOT synchronised with the host in between
  constexpr int EPP = TIMING_EPP;
  const size_t need = (size_t)EPP * (size_t)iters + 2;
  rc = reserve_timing(c, iters);
  if (rc) return rc;
  hipEvent_t ev_begin = c->tev[need - 2], ev_end = c->tev[need - 1];
  // the refit kernel's own clock readings: atomicMin over "all ones", atomicMax over zero (set before the region begins)
  HIP_TRY(c, hipMemsetAsync(c->d_kclock, 0xff, sizeof(unsigned long long) * (size_t)c->kclock_cap, c->stream));
  HIP_TRY(c, hipMemsetAsync(c->d_kclock + c->kclock_cap, 0, sizeof(unsigned long long) * (size_t)c->kclock_cap, c->stream));
  HIP_TRY(c, hipEventRecord(ev_begin, c->stream));
  int last_of_slot[FSDP_MAX_OVERLAP];
  bool started[FSDP_MAX_OVERLAP];
  for (int i = 0; i < FSDP_MAX_OVERLAP; i++) {
    last_of_slot[i] = -1;
    started[i] = false;
  }
  int n_stages = 0;
  c->tev_recorded.assign((size_t)iters, 0u);
  for (int it = 0; it < iters; it++) {
    const int si = (c->overlap > 1) ? (int)(c->turn++ % (unsigned)c->overlap) : 0;
    Work& q = c->slot[si];
    c->last_slot = si;
    if (!started[si] && si != 0) HIP_TRY(c, hipStreamWaitEvent(q.stream, ev_begin, 0));
    started[si] = true;
    StageEvents t;
    t.ev = &c->tev[(size_t)EPP * (size_t)it];
    t.main_only = c->time_main_only;
    // (opt-in, fsdp_time_detail bit 1: the readings are two atomics per workgroup on one address — the launches of a region
    // timed without them are the production launches)
    t.clock_first = c->time_kernel_clock ? c->d_kclock + it : nullptr;
    t.clock_last = c->time_kernel_clock ? c->d_kclock + c->kclock_cap + it : nullptr;
    if ((rc = launch_pass(c, q, c->res, &t))) return rc;
    n_stages = t.n - 1;
    c->tev_recorded[it] = t.recorded;
    last_of_slot[si] = it;
  }
  // the end event follows the last pass of every slot
  for (int i = 1; i < FSDP_MAX_OVERLAP; i++)
    if (last_of_slot[i] >= 0) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->tev[(size_t)EPP * (size_t)last_of_slot[i] + n_stages], 0));
  HIP_TRY(c, hipEventRecord(ev_end, c->stream));
  HIP_TRY(c, hipEventSynchronize(ev_end));
  const long long reruns_before = c->reruns;
  rc = sync_all(c);
  if (rc) return rc;
  HIP_TRY(c, hipGetLastError());
  if (c->reruns != reruns_before) {
    c->err = "fsdp_time_runs: a timed pass lacked a route kernel it needed (internal: resident batch not checked)";
    return 2;
  }
  c->timed_iters = iters;
  c->timed_stages = n_stages;
  if (ms_total || ms_stage) return fsdp_time_results(c, ms_total, ms_stage);
  return 0;
}

int fsdp_time_kernel_clock(fsdp_ctx* c, double* ms_sum, int* launches) {
  if (!c || !ms_sum || !launches) return 1;
  *ms_sum = 0.0;
  *launches = 0;
  if (c->timed_iters <= 0 || !c->d_kclock) return 0;
  HIP_TRY(c, hipSetDevice(c->device));
  int khz = 0;
  HIP_TRY(c, hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device));
  if (khz <= 0) return 0;
  std::vector<unsigned long long> h(2 * (size_t)c->kclock_cap);
  HIP_TRY(c, hipMemcpy(h.data(), c->d_kclock, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
  for (int it = 0; it < c->timed_iters && it < c->kclock_cap; it++) {
    const unsigned long long a = h[(size_t)it], b = h[(size_t)c->kclock_cap + (size_t)it];
    if (a == ~0ull || b == 0ull || b < a) continue;  // (a pass whose path stage was not the three-kernel form)
    *ms_sum += (double)(b - a) / (double)khz;
    (*launches)++;
  }
  return 0;
}

int fsdp_time_detail(fsdp_ctx* c, int every_kernel) {
  if (!c) return 1;
  c->time_main_only = (every_kernel & 1) == 0;
  c->time_kernel_clock = (every_kernel & 2) != 0;
  return 0;
}

int fsdp_stage_names(fsdp_ctx* c, char* out, int cap) {
  if (!c || !out || cap < 64) return 1;
  snprintf(out, (size_t)cap, "%s", c->stage_names.c_str());
  return 0;
}

// ---- stage-level entry points (host buffers, blocking, slot 0; the route kernels always run) ---------------------------
int fsdp_sort_batch(fsdp_ctx* c, int n_frames, const int32_t* off, const double* cones, const double* poses,
                    fsdp_frame_result* results) {
  if (!c) return 1;
  if (c->outstanding) return busy_error(c, "fsdp_sort_batch");
  HIP_TRY(c, hipSetDevice(c->device));
  size_t total;
  int max_cones;
  if (int rc = check_batch(c, n_frames, off, cones, poses, &total, &max_cones)) return rc;
  if (n_frames == 0) return 0;
  if (int rc = sync_all(c)) return rc;
  Work& q = c->slot[0];
  if (int rc = ensure_work(c, q, n_frames)) return rc;
  if (int rc = upload_inputs(c, q.in, q.stream, n_frames, off, cones, poses, nullptr, total, max_cones)) return rc;
  Inputs fin;
  const bool filtered = !c->params.use_unknown_cones;
  if (filtered)
    if (int rc = launch_filter(c, q, q.in, &fin)) return rc;
  const Inputs& in = filtered ? fin : q.in;
  launch_sort(c, q, in);
  if (int rc = launch_sort_big(c, q, in)) return rc;
  HIP_TRY(c, hipMemsetAsync(q.d_big, 0, sizeof(int), q.stream));  // (no assemble_kernel follows to reset the list)
  if (int rcs = ensure_staging(c, n_frames)) return rcs;
  HIP_TRY(c, hipMemcpyAsync(c->h_sort, q.d_sort, sizeof(SortOut) * n_frames, hipMemcpyDeviceToHost, q.stream));
  std::vector<int32_t> map, moff;
  if (filtered) {  // indices back into the caller's index space (what assemble_kernel does for a full pass)
    map.resize(total ? total : 1);
    moff.resize((size_t)n_frames + 1);
    HIP_TRY(c, hipMemcpyAsync(map.data(), q.f_map, sizeof(int32_t) * total, hipMemcpyDeviceToHost, q.stream));
    HIP_TRY(c, hipMemcpyAsync(moff.data(), q.f_off, sizeof(int32_t) * ((size_t)n_frames + 1), hipMemcpyDeviceToHost, q.stream));
  }
  HIP_TRY(c, hipStreamSynchronize(q.stream));
  for (int i = 0; i < n_frames; i++) {
    memset(&results[i], 0, sizeof(fsdp_frame_result));
    assemble(&c->h_sort[i], nullptr, nullptr, &results[i]);
    if (filtered) {
      auto back = [&](int32_t& v) {
        if (v >= 0) v = map[(size_t)moff[i] + v];
      };
      for (int k = 0; k < MAX_LEN; k++) {
        back(results[i].left_idx[k]);
        back(results[i].right_idx[k]);
      }
      for (int k = 0; k < 2; k++) {
        back(results[i].first_k_left[k]);
        back(results[i].first_k_right[k]);
      }
    }
  }
  return 0;
}

int fsdp_match_batch(fsdp_ctx* c, int n_frames, const double* sorted_left, const int32_t* n_left, const double* sorted_right,
                     const int32_t* n_right, const double* poses, fsdp_frame_result* results) {
  if (!c || n_frames < 0) return 1;
  if (c->outstanding) return busy_error(c, "fsdp_match_batch");
  if (n_frames == 0) return 0;
  HIP_TRY(c, hipSetDevice(c->device));
  // express the already sorted cones as a tiny frame each: cones = [left..., right...], indices 0..nl-1 / nl..nl+nr-1
  std::vector<int32_t> off(n_frames + 1, 0);
  std::vector<double> cones;
  std::vector<SortOut> so(n_frames);
  for (int f = 0; f < n_frames; f++) {
    int nl = n_left[f], nr = n_right[f];
    if (nl < 0 || nl > MAX_LEN || nr < 0 || nr > MAX_LEN) {
      c->err = "fsdp_match_batch: side length out of range";
      return 1;
    }
    memset(&so[f], 0, sizeof(SortOut));
    for (int i = 0; i < MAX_LEN; i++) so[f].left_idx[i] = so[f].right_idx[i] = -1;
    so[f].n_left = nl;
    so[f].n_right = nr;
    for (int i = 0; i < nl; i++) {
      so[f].left_idx[i] = i;
      cones.push_back(sorted_left[((size_t)f * MAX_LEN + i) * 2]);
      cones.push_back(sorted_left[((size_t)f * MAX_LEN + i) * 2 + 1]);
      cones.push_back((double)T_LEFT);
    }
    for (int i = 0; i < nr; i++) {
      so[f].right_idx[i] = nl + i;
      cones.push_back(sorted_right[((size_t)f * MAX_LEN + i) * 2]);
      cones.push_back(sorted_right[((size_t)f * MAX_LEN + i) * 2 + 1]);
      cones.push_back((double)T_RIGHT);
    }
    off[f + 1] = off[f] + nl + nr;
  }
  if (int rc = sync_all(c)) return rc;
  Work& q = c->slot[0];
  if (int rc = ensure_work(c, q, n_frames)) return rc;
  if (int rc = upload_inputs(c, q.in, q.stream, n_frames, off.data(), cones.data(), poses, nullptr, cones.size() / 3, 2 * MAX_LEN)) return rc;
  HIP_TRY(c, hipMemcpyAsync(q.d_sort, so.data(), sizeof(SortOut) * n_frames, hipMemcpyHostToDevice, q.stream));
  launch_match(c, q, q.in);
  if (int rcs = ensure_staging(c, n_frames)) return rcs;
  HIP_TRY(c, hipMemcpyAsync(c->h_match, q.d_match, sizeof(MatchOut) * n_frames, hipMemcpyDeviceToHost, q.stream));
  HIP_TRY(c, hipStreamSynchronize(q.stream));
  for (int i = 0; i < n_frames; i++) {
    memset(&results[i], 0, sizeof(fsdp_frame_result));
    assemble(nullptr, &c->h_match[i], nullptr, &results[i]);
  }
  return 0;
}

// centers / n_centers / centers_cap: optional side output (fsdp_path_batch_centers)
static int path_batch_impl(fsdp_ctx* c, int n_frames, const double* poses, const double* prev_paths, fsdp_frame_result* results,
                           double* centers, int32_t* n_centers, int centers_cap) {
  if (!c || n_frames < 0) return 1;
  if (c->outstanding) return busy_error(c, "fsdp_path_batch");
  if (n_frames == 0) return 0;
  HIP_TRY(c, hipSetDevice(c->device));
  if (int rc = sync_all(c)) return rc;  // passes in flight still use the buffers ensure_work may replace
  // the centre points leave the kernels through a buffer the device copy of the parameters points to for this one call
  struct CentersScope {
    fsdp_ctx* c;
    double* d_xy = nullptr;
    int32_t* d_n = nullptr;
    ~CentersScope() {
      if (!d_xy && !d_n) return;
      c->params.centers = nullptr;
      c->params.n_centers = nullptr;
      c->params.centers_cap = 0;
      (void)hipMemcpy(c->d_params, &c->params, sizeof(Params), hipMemcpyHostToDevice);
      (void)hipFree(d_xy);
      (void)hipFree(d_n);
    }
  } cs{c};
  if (centers) {
    HIP_TRY(c, hipMalloc(&cs.d_xy, sizeof(double) * 2 * (size_t)centers_cap * (size_t)n_frames));
    HIP_TRY(c, hipMalloc(&cs.d_n, sizeof(int32_t) * (size_t)n_frames));
    HIP_TRY(c, hipMemset(cs.d_n, 0, sizeof(int32_t) * (size_t)n_frames));
    c->params.centers = cs.d_xy;
    c->params.n_centers = cs.d_n;
    c->params.centers_cap = centers_cap;
    HIP_TRY(c, hipMemcpy(c->d_params, &c->params, sizeof(Params), hipMemcpyHostToDevice));
  }
  Work& q = c->slot[0];
  if (int rc = ensure_work(c, q, n_frames)) return rc;
  if (int rc = ensure_inputs(c, q.in, n_frames, 1, prev_paths != nullptr)) return rc;
  std::vector<MatchOut> mo(n_frames);
  for (int f = 0; f < n_frames; f++) {
    memset(&mo[f], 0, sizeof(MatchOut));
    const fsdp_frame_result& r = results[f];
    if (r.n_left_v < 0 || r.n_left_v > MAX_MATCH || r.n_right_v < 0 || r.n_right_v > MAX_MATCH) {
      c->err = "fsdp_path_batch: cone count out of range";
      return 1;
    }
    mo[f].n_left_v = r.n_left_v;
    mo[f].n_right_v = r.n_right_v;
    memcpy(mo[f].left_v, r.left_v, sizeof(r.left_v));
    memcpy(mo[f].right_v, r.right_v, sizeof(r.right_v));
    memcpy(mo[f].l2r, r.l2r, sizeof(r.l2r));
    memcpy(mo[f].r2l, r.r2l, sizeof(r.r2l));
  }
  q.in.n_frames = n_frames;
  q.in.max_cones = 0;
  q.in.use_prev = prev_paths != nullptr;
  HIP_TRY(c, hipMemcpyAsync(q.d_match, mo.data(), sizeof(MatchOut) * n_frames, hipMemcpyHostToDevice, q.stream));
  HIP_TRY(c, hipMemcpyAsync(q.in.d_poses, poses, sizeof(double) * 4 * (size_t)n_frames, hipMemcpyHostToDevice, q.stream));
  if (prev_paths)
    HIP_TRY(c, hipMemcpyAsync(q.in.d_prev, prev_paths, sizeof(double) * PATH_POINTS * 4 * (size_t)n_frames, hipMemcpyHostToDevice, q.stream));
  std::string names;
  launch_path(c, q, q.in, nullptr, names, n_frames);
  launch_path_retry(c, q, q.in);
  HIP_TRY(c, hipMemsetAsync(q.d_retry, 0, sizeof(int), q.stream));  // (no assemble_kernel follows to reset the list)
  c->stage_names = names + "path_retry_kernel";
  if (int rcs = ensure_staging(c, n_frames)) return rcs;
  HIP_TRY(c, hipMemcpyAsync(c->h_path, q.d_path, sizeof(PathOut) * n_frames, hipMemcpyDeviceToHost, q.stream));
  HIP_TRY(c, hipStreamSynchronize(q.stream));
  for (int i = 0; i < n_frames; i++) {
    results[i].status = 0;
    assemble(nullptr, nullptr, &c->h_path[i], &results[i]);
  }
  if (centers) {
    HIP_TRY(c, hipMemcpy(centers, cs.d_xy, sizeof(double) * 2 * (size_t)centers_cap * (size_t)n_frames, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(n_centers, cs.d_n, sizeof(int32_t) * (size_t)n_frames, hipMemcpyDeviceToHost));
  }
  return 0;
}

int fsdp_path_batch(fsdp_ctx* c, int n_frames, const double* poses, const double* prev_paths, fsdp_frame_result* results) {
  return path_batch_impl(c, n_frames, poses, prev_paths, results, nullptr, nullptr, 0);
}

int fsdp_path_batch_centers(fsdp_ctx* c, int n_frames, const double* poses, const double* prev_paths, fsdp_frame_result* results,
                            double* centers, int32_t* n_centers, int centers_cap) {
  if (!c) return 1;
  if (!centers || !n_centers || centers_cap <= 0) {
    c->err = "fsdp_path_batch_centers: centers, n_centers and a positive centers_cap are required";
    return 1;
  }
  return path_batch_impl(c, n_frames, poses, prev_paths, results, centers, n_centers, centers_cap);
}

#ifdef FSDP_PROFILE
int fsdp_profile_select(fsdp_ctx* c, int sort_kernel_instead_of_path) {
  if (!c) return 1;
  c->profile_sort = sort_kernel_instead_of_path != 0;
  return 0;
}
// profiling build only (tools/section_profile.py): per-frame per-section cycle sums of the path kernel, resident batch
int fsdp_profile_path(fsdp_ctx* c, long long* out32_per_frame) {
  if (!c || !c->resident || c->res.n_frames == 0) return 1;
  long long* d = nullptr;
  size_t bytes = sizeof(long long) * 32 * (size_t)c->res.n_frames;
  HIP_TRY(c, hipMalloc(&d, bytes));
  HIP_TRY(c, hipMemsetAsync(d, 0, bytes, c->stream));
  HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(fsdp::g_prof), &d, sizeof(d)));
  Work& q = c->slot[0];
  std::string names;
  if (c->profile_sort)
    launch_sort(c, q, c->res);
  else
    launch_path(c, q, c->res, nullptr, names, frames_in_flight(c, c->res.n_frames, false));
  HIP_TRY(c, hipMemsetAsync(q.d_big, 0, sizeof(int), q.stream));
  HIP_TRY(c, hipMemsetAsync(q.d_retry, 0, sizeof(int), q.stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, copy_sync(c, out32_per_frame, d, bytes, hipMemcpyDeviceToHost));
  long long* z = nullptr;
  HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(fsdp::g_prof), &z, sizeof(z)));
  (void)hipFree(d);
  return 0;
}
#endif

int fsdp_skidpad_set_tables(fsdp_ctx* c, const double* table_xy, int n_table, const double* noise, int n_noise) {
  if (!c || !table_xy || n_table < 20 || n_table > 8192 || !noise || n_noise < 6) return 1;
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = sync_all(c);
  if (rc) return rc;
  // known global path = table[::2] (skidpad_relocalizer.py:242-243)
  std::vector<double> half;
  for (int i = 0; i < n_table; i += 2) {
    half.push_back(table_xy[2 * i]);
    half.push_back(table_xy[2 * i + 1]);
  }
  if (c->d_table) (void)hipFree(c->d_table);
  if (c->d_noise) (void)hipFree(c->d_noise);
  c->d_table = c->d_noise = nullptr;
  c->have_tables = false;
  HIP_TRY(c, hipMalloc(&c->d_table, sizeof(double) * half.size()));
  HIP_TRY(c, hipMalloc(&c->d_noise, sizeof(double) * (size_t)n_noise));
  HIP_TRY(c, copy_sync(c, c->d_table, half.data(), sizeof(double) * half.size(), hipMemcpyHostToDevice));
  HIP_TRY(c, copy_sync(c, c->d_noise, noise, sizeof(double) * (size_t)n_noise, hipMemcpyHostToDevice));
  // the two reference centres and the table spacing are derived from the table on the device (skid_centers_kernel)
  double *d_full = nullptr, *d_scratch = nullptr, *d_out = nullptr;
  HIP_TRY(c, hipMalloc(&d_full, sizeof(double) * 2 * (size_t)n_table));
  HIP_TRY(c, hipMalloc(&d_scratch, sizeof(double) * 3 * (size_t)n_table));
  HIP_TRY(c, hipMalloc(&d_out, sizeof(double) * 5));
  HIP_TRY(c, hipMemcpyAsync(d_full, table_xy, sizeof(double) * 2 * (size_t)n_table, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(skid_centers_kernel, dim3(1), dim3(WAVE), 0, c->stream, d_full, n_table, d_scratch, d_out);
  hipError_t e = hipMemcpyAsync(c->skid_consts, d_out, sizeof(double) * 5, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d_full);
  (void)hipFree(d_scratch);
  (void)hipFree(d_out);
  HIP_TRY(c, e);
  c->tables.path = c->d_table;
  c->tables.n_path = (int)(half.size() / 2);
  c->tables.noise = c->d_noise;
  c->tables.n_noise = n_noise;
  c->tables.ref_right[0] = c->skid_consts[0];
  c->tables.ref_right[1] = c->skid_consts[1];
  c->tables.ref_left[0] = c->skid_consts[2];
  c->tables.ref_left[1] = c->skid_consts[3];
  c->tables.mean_distance = c->skid_consts[4];
  c->tables.prm = c->d_params;
  c->have_tables = true;
  return 0;
}

int fsdp_skidpad_constants(fsdp_ctx* c, double* out5) {
  if (!c || !out5 || !c->have_tables) return 1;
  memcpy(out5, c->skid_consts, sizeof(double) * 5);
  return 0;
}


int fsdp_skidpad_reset(fsdp_ctx* c, int n_instances) {
  if (!c || n_instances <= 0) return 1;
  if (c->outstanding) return busy_error(c, "fsdp_skidpad_reset");
  HIP_TRY(c, hipSetDevice(c->device));
  if (int rc = sync_all(c)) return rc;
  if (n_instances != c->n_instances) {
    HIP_TRY(c, regrow(c->d_skid, (size_t)n_instances));
    HIP_TRY(c, regrow(c->d_skid_backup, (size_t)n_instances));
    HIP_TRY(c, regrow(c->d_skid_sync, (size_t)n_instances + 1));
    c->n_instances = n_instances;
  }
  // fresh planners: nothing latched, previous path = the constant initial path
  std::vector<SkidState> init(n_instances);
  double def[PATH_POINTS][4];
  HIP_TRY(c, copy_sync(c, def, c->d_default_path, sizeof(def), hipMemcpyDeviceToHost));
  for (auto& s : init) {
    memset(&s, 0, sizeof(s));
    memcpy(s.prev, def, sizeof(def));
  }
  HIP_TRY(c, copy_sync(c, c->d_skid, init.data(), sizeof(SkidState) * (size_t)n_instances, hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemsetAsync(c->d_skid_sync, 0, sizeof(uint32_t) * ((size_t)n_instances + 1), c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->skid_ticket_base = 0;
  c->skid_step_no = 0;
  c->skid_all_reloc = false;
  return 0;
}

// Steps per launch for a caller that submits ahead (half the slots at most, so that one group runs while the next one is
// being submitted).  Thousands of (instance, step) pairs go through the packed kernels of the autocross path stage: the
// group is made as large as gives them 16384 frames (the more steps share the launches the better: 1024 instances plan
// 3.0 / 3.8 / 4.1 / 4.6 M frames/s in groups of 4 / 8 / 12 / 16); below that the steps get a wavefront each (skid_path_kernel) and the
// group is what puts at least three wavefronts on every SIMD (256 CUs x 4 SIMDs; a partly filled second round costs less
// than an unfilled first one: profiles/r03_skidpad_groups.txt).
static int skid_group_size(const fsdp_ctx* c) {
  const int n = c->n_instances;
  const int half = c->overlap / 2 > 1 ? c->overlap / 2 : 1;
  auto clamp = [&](int g) { return g < 1 ? 1 : g > SKID_GROUP_MAX ? SKID_GROUP_MAX : g > half ? half : g; };
  if (c->skid_group_env > 0) return c->skid_group_env > SKID_GROUP_MAX ? SKID_GROUP_MAX : c->skid_group_env > c->overlap ? c->overlap : c->skid_group_env;
  const int packed = clamp((16384 + n - 1) / n);
  if ((long long)packed * n >= c->skid_pack_min && c->params.max_deg == 3) return packed;
  return clamp((3072 + n - 1) / n);
}

static SkidGroup skid_group_of(fsdp_ctx* c, const int* slots, int n_steps, int step0) {
  SkidGroup g;
  memset(&g, 0, sizeof(g));
  for (int k = 0; k < n_steps; k++) {
    Work& q = c->slot[slots[k]];
    g.step[k] = SkidStep{q.in.d_poses, q.skid_attempted ? q.d_skid_status : nullptr, q.d_arena, q.d_path, q.d_skid_info};
  }
  g.n_steps = n_steps;
  g.step0 = step0;
  g.ticket_base = c->skid_ticket_base;
  return g;
}

// skid_path_kernel for the steps whose inputs, relocalization status and output records sit in slots[0 .. n_steps): one
// wavefront per (instance, step), csrc/skidpad_kernel.h "steps in flight, a wavefront per (instance, step)"
static void launch_skid_path(fsdp_ctx* c, const int* slots, int n_steps, int step0) {
  const int n = c->n_instances;
  const SkidGroup g = skid_group_of(c, slots, n_steps, step0);
  c->skid_ticket_base += (uint32_t)n * (uint32_t)n_steps;
  hipLaunchKernelGGL(skid_path_kernel, dim3((unsigned)n * (unsigned)n_steps), dim3(WAVE), 0, c->stream, n, g, c->d_skid, c->tables, c->d_chord,
                     c->d_skid_sync);
}

static int launch_skid_packed(fsdp_ctx* c, const int* slots, int n_steps, int step0) {
  const int n = c->n_instances;
  const int frames = n * n_steps;
  if ((size_t)frames > c->g_cap) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // (room for the groups this context forms: 16 384 frames with 1024 planners ~ 1.4 GB, most of it fit workspace)
    const size_t m = std::max((size_t)frames, (size_t)n * (size_t)skid_group_size(c));
    HIP_TRY(c, regrow(c->d_g_arena, (size_t)ARENA_DOUBLES * m));
    HIP_TRY(c, regrow(c->d_g_mid, m));
    HIP_TRY(c, regrow(c->d_g_out, m));
    HIP_TRY(c, regrow(c->d_g_retry, m + 1));
    HIP_TRY(c, regrow(c->d_g_sel, m));
    c->g_cap = m;
  }
  const SkidGroup g = skid_group_of(c, slots, n_steps, step0);
  hipStream_t xs = c->stream;
  HIP_TRY(c, hipMemsetAsync(c->d_g_retry, 0, sizeof(int), xs));  // (the packed kernels' list of frames they hand on; the commit kernel goes by the frames' records)
  skid_group_mark(c);
  hipLaunchKernelGGL(skid_select_kernel, dim3((unsigned)n), dim3(WAVE), 0, xs, n, g, c->d_skid, c->tables, c->d_g_sel);
  if (frames >= PACK_FRAMES) {
    if (c->fit_g == 4)
      launch_skid_packed_kernels<8, 4>(c, frames);
    else
      launch_skid_packed_kernels<8, 8>(c, frames);
  } else {
    launch_skid_packed_kernels<16, 16>(c, frames);
  }
  hipLaunchKernelGGL(skid_commit_kernel, dim3((unsigned)n), dim3(WAVE), 0, xs, n, g, c->d_skid, c->tables, c->d_chord, c->d_g_sel, c->d_g_mid, c->d_g_out,
                     c->d_g_arena, c->d_skid_sync);
  skid_group_mark(c);
  if (c->skid_time_groups) c->skid_group_frames.push_back(frames);
  return 0;
}

// The path kernel, result assembly and completion event of the steps submitted so far whose launch was put off
// (fsdp_skidpad_submit): one skid_path_kernel over all of them.
static int flush_skid(fsdp_ctx* c) {
  const int n_steps = c->n_skid_pending;
  if (n_steps == 0) return 0;
  c->n_skid_pending = 0;
  HIP_TRY(c, hipSetDevice(c->device));
  const int n = c->n_instances;
  hipStream_t xs = c->stream;
  if ((long long)n * n_steps >= c->skid_pack_min && c->params.max_deg == 3) {
    if (int rc = launch_skid_packed(c, c->skid_pending, n_steps, c->skid_step_no - n_steps)) return rc;
  } else {
    launch_skid_path(c, c->skid_pending, n_steps, c->skid_step_no - n_steps);
  }
  for (int k = 0; k < n_steps; k++) {
    Work& q = c->slot[c->skid_pending[k]];
    Work::Ticket& t = q.tk[0];
    t.pending = false;
    // page-locked results: assemble_kernel writes them into the caller's buffer (over PCIe); the planners' information
    // records ride along into the ticket's pinned block
    // (only a buffer that is page-locked over its WHOLE extent — decided at submit time, `via_stage` otherwise: a view
    // that merely starts inside a registered range must not be written from the device)
    fsdp_frame_result* direct = (t.user_results && !t.via_stage) ? (fsdp_frame_result*)device_view(t.user_results) : nullptr;
    if (t.compact) {
      // compact results = the path stage's own records: one plain copy (and the information records) instead of the assembly of
      // 2.4 KB results whose sorting / matching fields a skidpad step leaves empty anyway
      CopySegs segs;
      segs.n = 0;
      segs.rebase = 0;
      if (direct) segs.seg[segs.n++] = CopySeg{q.d_path, direct, sizeof(PathOut) * (unsigned long long)n};
      if (t.user_info) segs.seg[segs.n++] = CopySeg{q.d_skid_info, device_view(t.h_info), sizeof(SkidInfo) * (unsigned long long)n};
      if (segs.n) hipLaunchKernelGGL(stage_in_kernel, dim3(128), dim3(256), 0, xs, segs);
      HIP_TRY(c, hipGetLastError());
      if (t.via_stage) HIP_TRY(c, hipMemcpyAsync(t.h_stage, q.d_path, sizeof(PathOut) * (size_t)n, hipMemcpyDeviceToHost, xs));
      HIP_TRY(c, hipEventRecord(t.done, xs));
      continue;
    }
    if (t.user_results || t.user_info)
      launch_assemble(c, q, t.user_results ? n : 0, true, direct, xs, t.user_info ? q.d_skid_info : nullptr,
                      t.user_info ? (SkidInfo*)device_view(t.h_info) : nullptr);
    HIP_TRY(c, hipGetLastError());
    if (t.via_stage) HIP_TRY(c, hipMemcpyAsync(t.h_stage, q.d_result, sizeof(fsdp_frame_result) * (size_t)n, hipMemcpyDeviceToHost, xs));
    HIP_TRY(c, hipEventRecord(t.done, xs));
  }
  return 0;
}

// One frame for every planner instance, asynchronously.  Every command goes to the context's main stream in submit order;
// a step's transfers are kernels of that same stream when the caller's buffers are page-locked (stage_in_kernel reads the
// inputs from host memory, assemble_kernel writes results and planner information into it), so nothing ever waits for
// the host: a replay that knows its frames ahead submits ahead (up to `depth` steps, each with its own buffers on the
// device) and collects behind.  The inputs and the relocalization attempt of a step are enqueued at once; its path
// kernel is put off until `skid_group` steps have been submitted — they share one launch, with one wavefront per
// (instance, step) — or until somebody asks for the step (fsdp_collect, fsdp_ticket_done, any blocking call), so a live
// car that submits and collects one step at a time (fsdp_skidpad_step) gets one launch per step.
static int skidpad_submit_impl(fsdp_ctx* c, int n_instances, const int32_t* off, const double* cones, const double* poses,
                               fsdp_frame_result* results, fsdp_skidpad_info* info, long long* ticket, bool compact) {
  if (!c || !ticket) return 1;
  *ticket = -1;
  if (!c->have_tables || n_instances != c->n_instances || !c->d_skid) {
    c->err = "fsdp_skidpad_submit: call fsdp_skidpad_set_tables and fsdp_skidpad_reset(n_instances) first";
    return 1;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  size_t total;
  int max_cones;
  if (int rc = check_batch(c, n_instances, off, cones, poses, &total, &max_cones)) return rc;
  // one step per slot (the slots only hold the steps' buffers, every command goes to the main stream): the next free one
  int si = (int)(c->next_ticket % c->overlap);
  for (int k = 0; k < c->overlap && c->slot[si].tk[0].id >= 0; k++) si = (si + 1) % c->overlap;
  Work& q = c->slot[si];
  Work::Ticket& t = q.tk[0];
  if (t.id >= 0) {
    c->err = "fsdp_skidpad_submit: all " + std::to_string(c->overlap) + " slots hold a ticket; collect one first, e.g. ticket " + std::to_string(t.id);
    return 4;
  }
  // (the slot's ticket is free, i.e. collected: nothing queued uses its buffers any more, they may be replaced)
  if (int rc = ensure_work(c, q, n_instances)) return rc;
  hipStream_t xs = c->stream;
  // Once every planner is relocalized nobody reads cones any more (Relocalizer.attempt_relocalization_calculation returns
  // at once, relocalization_base_class.py:56-57; the path comes from the known map): they stay on the host, and the
  // relocalization kernel is not launched.  (Known from the planner information of a collected step.)
  const bool attempt = !c->skid_all_reloc;
  if (!attempt) total = 0;
  const bool in_pinned = is_pinned(off, sizeof(int32_t) * ((size_t)n_instances + 1)) && is_pinned(poses, sizeof(double) * 4 * (size_t)n_instances) &&
                         (total == 0 || is_pinned(cones, sizeof(double) * 3 * total));
  if (in_pinned) {
    if (int rc = stage_inputs(c, q.in, xs, n_instances, off, cones, poses, nullptr, total, max_cones)) return rc;
  } else if (int rc = upload_inputs(c, q.in, xs, n_instances, off, cones, poses, nullptr, total, max_cones)) {
    return rc;
  }
  q.skid_attempted = attempt;
  if (info && n_instances > t.cap_info) {
    if (t.h_info) (void)hipHostFree(t.h_info);
    t.h_info = nullptr;
    t.cap_info = 0;
    HIP_TRY(c, hipHostMalloc((void**)&t.h_info, sizeof(SkidInfo) * (size_t)n_instances, hipHostMallocDefault));
    t.cap_info = n_instances;
  }
  const bool direct = results && is_pinned(results, (compact ? sizeof(PathOut) : sizeof(fsdp_frame_result)) * (size_t)n_instances);
  if (results && !direct && n_instances > t.cap_stage) {
    if (t.h_stage) (void)hipHostFree(t.h_stage);
    t.h_stage = nullptr;
    t.cap_stage = 0;
    HIP_TRY(c, hipHostMalloc((void**)&t.h_stage, sizeof(fsdp_frame_result) * (size_t)n_instances, hipHostMallocDefault));
    t.cap_stage = n_instances;
  }
  if (!t.done) HIP_TRY(c, hipEventCreateWithFlags(&t.done, hipEventDisableTiming));
  if (attempt)
    hipLaunchKernelGGL(skid_reloc_kernel, dim3((unsigned)n_instances), dim3(WAVE), 0, c->stream, n_instances, q.in.d_off, q.in.d_cones, q.in.d_poses,
                       c->d_skid, c->tables, q.d_arena, q.d_skid_status, c->skid_step_no);
  HIP_TRY(c, hipGetLastError());
  c->skid_step_no++;
  q.pass_in = &q.in;
  q.pass_skid = true;
  q.unverified = false;
  t.n = n_instances;
  t.skid = true;
  t.pending = true;
  t.user_results = results;
  t.user_info = info;
  t.via_stage = results && !direct;
  t.compact = compact;
  c->skid_pending[c->n_skid_pending++] = si;
  c->last_slot = si;
  c->last_n = n_instances;
  t.id = c->next_ticket++;
  c->outstanding++;
  *ticket = t.id;
  if (c->n_skid_pending >= skid_group_size(c)) return flush_skid(c);
  return 0;
}

int fsdp_skidpad_submit(fsdp_ctx* c, int n_instances, const int32_t* off, const double* cones, const double* poses,
                        fsdp_frame_result* results, fsdp_skidpad_info* info, long long* ticket) {
  return skidpad_submit_impl(c, n_instances, off, cones, poses, results, info, ticket, false);
}
static_assert(sizeof(fsdp_path_result) == sizeof(PathOut) && offsetof(fsdp_path_result, status) == offsetof(PathOut, status) &&
                  offsetof(fsdp_path_result, path_fallback) == offsetof(PathOut, fallback) && offsetof(fsdp_path_result, n_dense) == offsetof(PathOut, n_dense),
              "fsdp_path_result is the path stage's record");
int fsdp_skidpad_submit_compact(fsdp_ctx* c, int n_instances, const int32_t* off, const double* cones, const double* poses,
                                fsdp_path_result* results, fsdp_skidpad_info* info, long long* ticket) {
  return skidpad_submit_impl(c, n_instances, off, cones, poses, (fsdp_frame_result*)results, info, ticket, true);
}

int fsdp_skidpad_step(fsdp_ctx* c, int n_instances, const int32_t* off, const double* cones, const double* poses,
                      fsdp_frame_result* results, fsdp_skidpad_info* info) {
  if (!c) return 1;
  if (c->outstanding) return busy_error(c, "fsdp_skidpad_step");
  long long t;
  if (int rc = fsdp_skidpad_submit(c, n_instances, off, cones, poses, results, info, &t)) return rc;
  return fsdp_collect(c, t);
}

// Timing of the packed path-stage kernels of the groups of steps a replay forms (fsdp_skidpad_submit): enable, replay, read.
int fsdp_skidpad_time_groups(fsdp_ctx* c, int enable) {
  if (!c) return 1;
  for (hipEvent_t e : c->skid_group_ev) (void)hipEventDestroy(e);
  c->skid_group_ev.clear();
  c->skid_group_frames.clear();
  c->skid_group_names.clear();
  c->skid_time_groups = enable != 0;
  return 0;
}
int fsdp_skidpad_group_times(fsdp_ctx* c, float* ms5, int* n_groups, long long* n_frames, char* names, int names_cap) {
  if (!c || !ms5 || !n_groups || !n_frames) return 1;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (int k = 0; k < 5; k++) ms5[k] = 0.f;
  const size_t groups = c->skid_group_frames.size();
  if (c->skid_group_ev.size() != 6 * groups) {
    c->err = "fsdp_skidpad_group_times: incomplete event set";
    return 1;
  }
  long long frames = 0;
  for (size_t gidx = 0; gidx < groups; gidx++) {
    for (int k = 0; k < 5; k++) {
      float t = 0.f;
      HIP_TRY(c, hipEventElapsedTime(&t, c->skid_group_ev[6 * gidx + k], c->skid_group_ev[6 * gidx + k + 1]));
      ms5[k] += t;
    }
    frames += c->skid_group_frames[gidx];
  }
  *n_groups = (int)groups;
  *n_frames = frames;
  if (names && names_cap > 0) snprintf(names, (size_t)names_cap, "%s", c->skid_group_names.c_str());
  return 0;
}

int fsdp_skidpad_time_path(fsdp_ctx* c, int iters, float* ms_total) {
  if (!c || !c->d_skid || c->n_instances <= 0 || iters <= 0) return 1;
  if (c->outstanding) return busy_error(c, "fsdp_skidpad_time_path");
  HIP_TRY(c, hipSetDevice(c->device));
  if (int rc = sync_all(c)) return rc;
  Work& q = c->slot[c->last_slot];
  if (!q.pass_skid || q.in.n_frames != c->n_instances) {
    c->err = "fsdp_skidpad_time_path: run a step first";
    return 1;
  }
  size_t bytes = sizeof(SkidState) * (size_t)c->n_instances;
  HIP_TRY(c, hipMemcpyAsync(c->d_skid_backup, c->d_skid, bytes, hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(c, hipEventRecord(c->ev[4], c->stream));
  for (int i = 0; i < iters; i++) launch_skid_path(c, &c->last_slot, 1, c->skid_step_no);  // (a step number of its own: nothing to wait for)
  HIP_TRY(c, hipEventRecord(c->ev[5], c->stream));
  HIP_TRY(c, hipEventSynchronize(c->ev[5]));
  float t = 0;
  HIP_TRY(c, hipEventElapsedTime(&t, c->ev[4], c->ev[5]));
  HIP_TRY(c, hipMemcpyAsync(c->d_skid, c->d_skid_backup, bytes, hipMemcpyDeviceToDevice, c->stream));
  // the timed launches published step skid_step_no + 1 for every planner while the restored state is that of step
  // skid_step_no: put the publish counters back too, or the second step of the next group launch would not wait for the first
  HIP_TRY(c, hipMemsetD32Async((hipDeviceptr_t)(c->d_skid_sync + 1), c->skid_step_no, (size_t)c->n_instances, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (ms_total) *ms_total = t;
  return 0;
}

// ---- per-stage intermediate of the path stage: the refit's spline ------------------------------------------------------
extern "C" int fsdp_debug_refit(fsdp_ctx* c, int32_t* n_knots, double* knots34, double* coeffs68) {
  if (!c || !n_knots || !knots34 || !coeffs68 || c->last_n <= 0) return 1;
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = sync_all(c);
  if (rc) return rc;
  Work& q = c->slot[c->last_slot];
  const int n = c->last_n;
  std::vector<FitRec> recs((size_t)n);
  std::vector<PathMid> mids((size_t)n);
  const size_t fit_off = (size_t)ARENA_FIT * sizeof(double);  // frame_arena(): A.fit
  HIP_TRY(c, hipMemcpy2DAsync(recs.data(), sizeof(FitRec), (const char*)q.d_arena + fit_off, sizeof(double) * ARENA_DOUBLES,
                              sizeof(FitRec), (size_t)n, hipMemcpyDeviceToHost, q.stream));
  HIP_TRY(c, hipMemcpyAsync(mids.data(), q.d_mid, sizeof(PathMid) * (size_t)n, hipMemcpyDeviceToHost, q.stream));
  HIP_TRY(c, hipStreamSynchronize(q.stream));
  for (int i = 0; i < n; i++) {
    // only frames that went prep -> fit -> finish hold a record (the others took the exact route or ended earlier)
    const bool fast = c->stage_names.find("fit_kernel") != std::string::npos && mids[i].status == ST_OK;
    n_knots[i] = fast ? recs[i].n : -1;
    memcpy(knots34 + (size_t)i * 34, recs[i].t, sizeof(double) * 34);
    memcpy(coeffs68 + (size_t)i * 68, recs[i].c, sizeof(double) * 68);
  }
  return 0;
}

// raw doubles of a frame's scratch arena of the most recent pass (tests / debug builds)
extern "C" int fsdp_debug_arena(fsdp_ctx* c, int frame, int offset, int count, double* out) {
  if (!c || !out || frame < 0 || frame >= c->last_n || offset < 0 || count < 0 || offset + count > ARENA_DOUBLES) return 1;
  HIP_TRY(c, hipSetDevice(c->device));
  if (int rc = sync_all(c)) return rc;
  Work& q = c->slot[c->last_slot];
  HIP_TRY(c, copy_sync(c, out, q.d_arena + (size_t)frame * ARENA_DOUBLES + offset, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost));
  return 0;
}

// ---- device arithmetic self-test ------------------------------------------------------------------------------------------
// The hand-rolled sequences of spline_device.h against the compiler's IEEE operations, element-wise on the device:
// out[0][i] = sqrt_1_2(x[i]), out[1][i] = sqrt(x[i]) (x in [1, 2]); out[2][i] = div_rcp(a[i], b[i], rcp_refined(b[i])),
// out[3][i] = a[i] / b[i]; out[4][i] = in_div_band(a[i]) && in_div_band(b[i]).
__global__ void math_selftest_kernel(int n, const double* __restrict__ x, const double* __restrict__ a, const double* __restrict__ b,
                                     double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = sqrt_1_2(x[i]);
  out[(size_t)n + i] = sqrt(x[i]);
  out[2 * (size_t)n + i] = div_rcp(a[i], b[i], rcp_refined(b[i]));
  out[3 * (size_t)n + i] = a[i] / b[i];
  out[4 * (size_t)n + i] = (in_div_band(a[i]) && in_div_band(b[i])) ? 1.0 : 0.0;
}
// max_abs_nn / min_abs_nn (the one-instruction max(|a|, b) / min(|a|, b) of the Givens step) element-wise: out = [max | min]
__global__ void absminmax_selftest_kernel(int n, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = max_abs_nn(a[i], b[i]);
  out[(size_t)n + i] = min_abs_nn(a[i], b[i]);
}
extern "C" int fsdp_selftest_absminmax(fsdp_ctx* c, int n, const double* a, const double* b, double* out2n) {
  if (!c || n <= 0 || !a || !b || !out2n) return 1;
  HIP_TRY(c, hipSetDevice(c->device));
  double *da = nullptr, *db = nullptr, *dout = nullptr;
  const size_t bytes = sizeof(double) * (size_t)n;
  HIP_TRY(c, hipMalloc(&da, bytes));
  HIP_TRY(c, hipMalloc(&db, bytes));
  HIP_TRY(c, hipMalloc(&dout, 2 * bytes));
  hipError_t e = hipMemcpyAsync(da, a, bytes, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(db, b, bytes, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(absminmax_selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, da, db, dout);
    e = hipMemcpyAsync(out2n, dout, 2 * bytes, hipMemcpyDeviceToHost, c->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(da);
  (void)hipFree(db);
  (void)hipFree(dout);
  HIP_TRY(c, e);
  return 0;
}

extern "C" int fsdp_selftest_math(fsdp_ctx* c, int n, const double* x, const double* a, const double* b, double* out5n) {
  if (!c || n <= 0 || !x || !a || !b || !out5n) return 1;
  HIP_TRY(c, hipSetDevice(c->device));
  double *dx = nullptr, *da = nullptr, *db = nullptr, *dout = nullptr;
  const size_t bytes = sizeof(double) * (size_t)n;
  HIP_TRY(c, hipMalloc(&dx, bytes));
  HIP_TRY(c, hipMalloc(&da, bytes));
  HIP_TRY(c, hipMalloc(&db, bytes));
  HIP_TRY(c, hipMalloc(&dout, 5 * bytes));
  hipError_t e = hipMemcpyAsync(dx, x, bytes, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(da, a, bytes, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(db, b, bytes, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(math_selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, dx, da, db, dout);
    e = hipMemcpyAsync(out5n, dout, 5 * bytes, hipMemcpyDeviceToHost, c->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(dx);
  (void)hipFree(da);
  (void)hipFree(db);
  (void)hipFree(dout);
  HIP_TRY(c, e);
  return 0;
}

// The Givens step's arithmetic (spline_device.h fpgivs_guarded<true>: max / min, the first quotient, sqrt on [1, 2], the reciprocal of dd
// seeded from the square root's own iterate, the two quotients) next to FITPACK's fpgivs with the compiler's IEEE operations:
// out = [cs | sn | dd] fast, [cs | sn | dd] IEEE, [guard: 1 = operands inside the fast sequence's band]
__global__ void givens_selftest_kernel(int n, const double* __restrict__ piv, const double* __restrict__ ww, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double w = ww[i], cs, sn;
  int bad = 0;
  fpgivs_guarded<true>(piv[i], w, cs, sn, bad);
  out[i] = cs;
  out[(size_t)n + i] = sn;
  out[2 * (size_t)n + i] = w;
  double w2 = ww[i], cs2, sn2;
  fpgivs(piv[i], w2, cs2, sn2);
  out[3 * (size_t)n + i] = cs2;
  out[4 * (size_t)n + i] = sn2;
  out[5 * (size_t)n + i] = w2;
  out[6 * (size_t)n + i] = bad ? 0.0 : 1.0;
}
extern "C" int fsdp_selftest_givens(fsdp_ctx* c, int n, const double* piv, const double* ww, double* out7n) {
  if (!c || n <= 0 || !piv || !ww || !out7n) return 1;
  HIP_TRY(c, hipSetDevice(c->device));
  double *da = nullptr, *db = nullptr, *dout = nullptr;
  const size_t bytes = sizeof(double) * (size_t)n;
  HIP_TRY(c, hipMalloc(&da, bytes));
  HIP_TRY(c, hipMalloc(&db, bytes));
  HIP_TRY(c, hipMalloc(&dout, 7 * bytes));
  hipError_t e = hipMemcpyAsync(da, piv, bytes, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(db, ww, bytes, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(givens_selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, da, db, dout);
    e = hipMemcpyAsync(out7n, dout, 7 * bytes, hipMemcpyDeviceToHost, c->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(da);
  (void)hipFree(db);
  (void)hipFree(dout);
  HIP_TRY(c, e);
  return 0;
}

// det3_lu (path_kernel.h: the sign of numpy.linalg.det of three homogeneous points) element-wise on the device
__global__ void det3_selftest_kernel(int n, const double* __restrict__ xy6, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* p = xy6 + 6 * (size_t)i;
  out[i] = det3_lu(p[0], p[1], p[2], p[3], p[4], p[5]);
}
extern "C" int fsdp_selftest_det3(fsdp_ctx* c, int n, const double* xy6, double* out) {
  if (!c || n <= 0 || !xy6 || !out) return 1;
  HIP_TRY(c, hipSetDevice(c->device));
  double *dx = nullptr, *dout = nullptr;
  HIP_TRY(c, hipMalloc(&dx, sizeof(double) * 6 * (size_t)n));
  HIP_TRY(c, hipMalloc(&dout, sizeof(double) * (size_t)n));
  hipError_t e = hipMemcpyAsync(dx, xy6, sizeof(double) * 6 * (size_t)n, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(det3_selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, dx, dout);
    e = hipMemcpyAsync(out, dout, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, c->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(dx);
  (void)hipFree(dout);
  HIP_TRY(c, e);
  return 0;
}

// The device libm values the sorting stage's discrete decisions hang on (atan2 of the search predicates, start-cone bearings and
// cost terms; acos where a cosine sits within 1e-9 of a threshold) next to the correctly rounded det_atan2 (det_math.h)
__global__ void libm_selftest_kernel(int n, const double* __restrict__ y, const double* __restrict__ x, const double* __restrict__ cs,
                                     double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = atan2(y[i], x[i]);
  out[(size_t)n + i] = detm::det_atan2(y[i], x[i]);
  out[2 * (size_t)n + i] = acos(cs[i]);
}
extern "C" int fsdp_selftest_libm(fsdp_ctx* c, int n, const double* y, const double* x, const double* cs, double* out3n) {
  if (!c || n <= 0 || !y || !x || !cs || !out3n) return 1;
  HIP_TRY(c, hipSetDevice(c->device));
  double *din = nullptr, *dout = nullptr;
  const size_t bytes = sizeof(double) * (size_t)n;
  HIP_TRY(c, hipMalloc(&din, 3 * bytes));
  HIP_TRY(c, hipMalloc(&dout, 3 * bytes));
  hipError_t e = hipMemcpyAsync(din, y, bytes, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(din + n, x, bytes, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(din + 2 * (size_t)n, cs, bytes, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(libm_selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, din, din + n, din + 2 * (size_t)n, dout);
    e = hipMemcpyAsync(out3n, dout, 3 * bytes, hipMemcpyDeviceToHost, c->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(din);
  (void)hipFree(dout);
  HIP_TRY(c, e);
  return 0;
}

// ---- multi-GPU: RCCL over xGMI (see fsdp_comm.h) ----------------------------------------------------------------------
#define NCCL_TRY(ctx, call)                                                                                    \
  do {                                                                                                         \
    ncclResult_t r_ = (call);                                                                                  \
    if (r_ != ncclSuccess) {                                                                                   \
      (ctx)->err = std::string(#call) + ": " + fsdp_comm::api().GetErrorString(r_);                            \
      return 3;                                                                                                \
    }                                                                                                          \
  } while (0)

int fsdp_comm_unique_id(void* out128) {
  if (!out128) return 1;
  if (!fsdp_comm::load()) {
    g_create_error = fsdp_comm::api().error;
    return 3;
  }
  static_assert(sizeof(ncclUniqueId) == FSDP_COMM_ID_BYTES, "unique id size");
  ncclUniqueId id;
  ncclResult_t r = fsdp_comm::api().GetUniqueId(&id);
  if (r != ncclSuccess) {
    g_create_error = std::string("ncclGetUniqueId: ") + fsdp_comm::api().GetErrorString(r);
    return 3;
  }
  memcpy(out128, &id, sizeof(id));
  return 0;
}

int fsdp_comm_init(fsdp_ctx* c, int rank, int world, const void* id128) {
  if (!c || !id128 || world < 1 || rank < 0 || rank >= world) return 1;
  if (c->comm.comm) {
    c->err = "fsdp_comm_init: communicator already initialised";
    return 1;
  }
  if (!fsdp_comm::load()) {
    c->err = fsdp_comm::api().error;
    return 3;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  NCCL_TRY(c, fsdp_comm::api().CommInitRank(&c->comm.comm, world, id, rank));
  fflush(stdout);  // RCCL prints its version banner through C stdio: out now, not after the caller's own last line
  c->comm.rank = rank;
  c->comm.world = world;
  return 0;
}

int fsdp_comm_size(fsdp_ctx* c) {
  if (!c || !c->comm.comm) return 0;
  int n = 0;
  if (fsdp_comm::api().CommCount(c->comm.comm, &n) != ncclSuccess) return 0;
  return n;
}

int fsdp_comm_rank(fsdp_ctx* c) {
  if (!c || !c->comm.comm) return -1;
  int r = -1;
  if (fsdp_comm::api().CommUserRank(c->comm.comm, &r) != ncclSuccess) return -1;
  return r;
}

static int comm_staging(fsdp_ctx* c, size_t bytes) {
  if (bytes <= c->comm.cap) return 0;
  if (c->comm.d_buf) (void)hipFree(c->comm.d_buf);
  c->comm.d_buf = nullptr;
  c->comm.cap = 0;
  HIP_TRY(c, hipMalloc(&c->comm.d_buf, bytes));
  c->comm.cap = bytes;
  return 0;
}

int fsdp_comm_broadcast(fsdp_ctx* c, void* host_buf, size_t bytes, int root) {
  if (!c || !c->comm.comm || (bytes > 0 && !host_buf) || root < 0 || root >= c->comm.world) return 1;
  if (bytes == 0) return 0;
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = comm_staging(c, bytes);
  if (rc) return rc;
  if (c->comm.rank == root) HIP_TRY(c, hipMemcpyAsync(c->comm.d_buf, host_buf, bytes, hipMemcpyHostToDevice, c->stream));
  NCCL_TRY(c, fsdp_comm::api().Broadcast(c->comm.d_buf, c->comm.d_buf, bytes, ncclUint8, root, c->comm.comm, c->stream));
  if (c->comm.rank != root) HIP_TRY(c, hipMemcpyAsync(host_buf, c->comm.d_buf, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return 0;
}

int fsdp_comm_allreduce(fsdp_ctx* c, double* values, int n, int op) {
  if (!c || !c->comm.comm || n < 0 || (n > 0 && !values) || op < 0 || op > 2) return 1;
  if (n == 0) return 0;
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t bytes = sizeof(double) * (size_t)n;
  int rc = comm_staging(c, bytes);
  if (rc) return rc;
  const ncclRedOp_t ops[3] = {ncclSum, ncclMax, ncclMin};
  HIP_TRY(c, hipMemcpyAsync(c->comm.d_buf, values, bytes, hipMemcpyHostToDevice, c->stream));
  NCCL_TRY(c, fsdp_comm::api().AllReduce(c->comm.d_buf, c->comm.d_buf, (size_t)n, ncclFloat64, ops[op], c->comm.comm, c->stream));
  HIP_TRY(c, hipMemcpyAsync(values, c->comm.d_buf, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return 0;
}

int fsdp_comm_barrier(fsdp_ctx* c) {
  if (!c || !c->comm.comm) return 1;
  int rc = sync_all(c);  // everything this rank has enqueued is done before it reports in
  if (rc) return rc;
  double one = 1.0;
  rc = fsdp_comm_allreduce(c, &one, 1, 0);
  if (rc) return rc;
  if ((int)one != c->comm.world) {
    c->err = "fsdp_comm_barrier: rank count mismatch";
    return 3;
  }
  return 0;
}

int fsdp_comm_destroy(fsdp_ctx* c) {
  if (!c) return 1;
  if (c->comm.comm) {
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)fsdp_comm::api().CommDestroy(c->comm.comm);
    c->comm.comm = nullptr;
  }
  if (c->comm.d_buf) (void)hipFree(c->comm.d_buf);
  c->comm.d_buf = nullptr;
  c->comm.cap = 0;
  c->comm.rank = 0;
  c->comm.world = 1;
  return 0;
}

int fsdp_default_path(fsdp_ctx* c, double* out) {
  if (!c || !out) return 1;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, copy_sync(c, out, c->d_default_path, sizeof(double) * PATH_POINTS * 4, hipMemcpyDeviceToHost));
  return 0;
}
}
