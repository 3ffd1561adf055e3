/* fsdp.h — C ABI of the MI355X-native batched PathPlanner hot path (libfsdp_hip.so).
 *
 * Drop-in boundary.  The reference (papalotis/ft-fsd-path-planning) is pure Python and has no
 * FFI layer; the boundary a replacement can sit behind is the method
 *   PathPlanner.calculate_path_in_global_frame(cones, vehicle_position, vehicle_direction,
 *                                              return_intermediate_results)
 *   fsd_path_planning/full_pipeline/full_pipeline.py:84-207
 * and the three public stage classes it wires together
 *   ConeSorting.run_cone_sorting      sorting_cones/core_cone_sorting.py:117-136
 *   ConeMatching.run_cone_matching    cone_matching/core_cone_matching.py:87-124
 *   CalculatePath.run_path_calculation calculate_path/core_calculate_path.py:514-575
 * Each entry point below names the reference interface it replaces.  The Python host
 * (ft-fsd-path-planning_amd/) binds these with ctypes and re-creates the reference's class
 * API on top; INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions: plain C types, caller-allocated buffers, no exceptions across the ABI.
 * Return value 0 = success; non-zero = API misuse or HIP error (fsdp_last_error()).  Per-frame
 * conditions are reported in fsdp_frame_result.status, never through the return code.
 * A context is bound to one GPU and its own HIP streams; contexts are independent and may be interleaved from one host
 * thread (every entry point selects its context's device first).  All floating point is IEEE float64.
 */
#ifndef FSDP_H
#define FSDP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The shapes of a result record.  The library exists in two builds of the same sources: the STANDARD one (libfsdp_hip.so: the
 * reference's default structural parameters, config.py:34-37,58) and the WIDE one (libfsdp_hip_wide.so, compiled — like a
 * binding that talks to it — with -DFSDP_WIDE_SHAPES) for contexts whose max_n_neighbors, max_length or
 * mpc_prediction_horizon exceed the standard shapes.  fsdp_shapes() reports which one a loaded library is; the Python host
 * picks the build from a context's parameters (ft-fsd-path-planning_amd/_capi.py). */
#ifdef FSDP_WIDE_SHAPES
#define FSDP_MAX_LEN 16       /* config.py:36 max_length                         */
#define FSDP_MAX_NEIGHBORS 8  /* config.py:34 max_n_neighbors                    */
#define FSDP_MAX_MATCH 32     /* cones incl. virtual per side after matching (2 x FSDP_MAX_LEN) */
#define FSDP_PATH_POINTS 64   /* config.py:58 mpc_prediction_horizon             */
#else
#define FSDP_MAX_LEN 12
#define FSDP_MAX_NEIGHBORS 5
#define FSDP_MAX_MATCH 24
#define FSDP_PATH_POINTS 40
#endif
#define FSDP_MAX_CONES 8192  /* cones per frame (frames beyond 255 are sorted with their state in global memory) */

/* ConeTypes — utils/cone_types.py:10-19 (values are part of the input format) */
enum { FSDP_CONE_UNKNOWN = 0, FSDP_CONE_RIGHT = 1, FSDP_CONE_LEFT = 2, FSDP_CONE_ORANGE_SMALL = 3, FSDP_CONE_ORANGE_BIG = 4 };

/* per-frame status */
enum {
  FSDP_OK = 0,
  /* the reference raises out of calculate_path_in_global_frame on this input */
  FSDP_REF_UNDEFINED_SET_DIFF = 101,  /* nearby_cone_search.py:88-94                         */
  FSDP_REF_UNDEFINED_DFS_OOB = 102,   /* end_configurations.py:369                           */
  FSDP_REF_UNDEFINED_PATH = 103,      /* core_calculate_path.py:482-483 / second fallback    */
  FSDP_REF_UNDEFINED_MATCH_IDX = 104, /* core_calculate_path.py:544 index into empty array   */
  /* fixed device capacities exceeded (the reference's buffers grow without bound) */
  FSDP_OVERFLOW_CONES = 201, /* more than 8192 cones in a frame */
  FSDP_OVERFLOW_ENDS = 202,  /* more than 4096 raw end configurations on one side */
  FSDP_OVERFLOW_PATH = 203,
  FSDP_OVERFLOW_KNOTS = 204, /* more than 256 knots in a spline (fits beyond the packed kernels' 16 / 32 / 64 are re-planned by the
                                one-frame-per-wavefront kernel, which keeps 256; a skidpad step: 64) */
  FSDP_OVERFLOW_CLUSTERS = 205, /* skidpad relocalization: more than 64 centre clusters */
  FSDP_SYNC_LOST = 206 /* skidpad steps sharing a launch: a step never saw its predecessor's state (internal error) */
};

/* Defined behaviour where the reference's is not (stated here because it is part of the contract, not an implementation detail):
 *
 * Exact ties.  The reference orders start-cone candidates and a cone's nearest neighbours with np.argsort, NumPy's unstable
 * default sort (trace_sorter/adjacency_matrix.py:55, core_trace_sorter.py:344-377): on an EXACT distance tie — mirror-symmetric
 * tracks, exactly duplicated cones — which cone it takes depends on the NumPy build and CPU (AVX-512 / AVX2 / scalar sort
 * kernels).  This library orders ties by the cones' position in the input: the LOWEST INDEX first (a stable sort), the same on
 * every route and batch composition.  On all tied demo scenarios the final sorted configurations equal the reference's
 * (tests/golden/scenarios.npz); where only intermediates are compared the tied frames are flagged in the fixtures.
 *
 * Capacities refused at fsdp_create (the reference takes any value; its buffers grow by doubling,
 * trace_sorter/end_configurations.py:74-105): max_n_neighbors > 5, max_length > 12 (register / result-struct shapes),
 * mpc_prediction_horizon > 40 (rows of a result path) in the standard build; > 8, > 16, > 64 in the wide build
 * (FSDP_WIDE_SHAPES above).  Refused per frame, with a status and never truncated:
 * more than 8192 cones (201), more than 4096 raw end configurations on a side (202), a working polyline beyond 1408 points
 * (203), more than 256 knots in a spline (204; FITPACK's own bound is nest = m + 2k, utils/spline_fit.py:117 — the noisiest
 * frames of the fuzz sets end with 171), more than 64 skidpad centre clusters (205). */

/* path_fallback bits */
enum {
  FSDP_FB_PREVIOUS_CENTER = 1, /* < 2 centre points / < 3 cones both sides: previous path (core_calculate_path.py:202-203,531-536) */
  FSDP_FB_SPLINE_ERROR = 2,    /* :218-221 */
  FSDP_FB_TOO_FAR = 4,         /* :235-236 */
  FSDP_FB_MPC_RETRY = 8,       /* :564-570 */
  FSDP_FB_ARC_EXTENSION = 16,  /* :301-324 */
  FSDP_FB_LINE_EXTENSION = 32  /* :326-331 */
};

/* What calculate_path_in_global_frame(..., return_intermediate_results=True) returns
 * (full_pipeline.py:196-205) plus the index form of the sorted cones. */
typedef struct {
  int32_t status;
  int32_t n_left, n_right;
  int32_t left_idx[FSDP_MAX_LEN];  /* left_config: indices into the frame's flattened cones, -1 padded  */
  int32_t right_idx[FSDP_MAX_LEN]; /* (trace_sorter/core_trace_sorter.py:197-214)                       */
  int32_t n_left_v, n_right_v;
  double left_v[FSDP_MAX_MATCH][2];  /* left_cones_with_virtual  */
  double right_v[FSDP_MAX_MATCH][2]; /* right_cones_with_virtual */
  int32_t l2r[FSDP_MAX_MATCH];       /* left_to_right_match, -1 = none */
  int32_t r2l[FSDP_MAX_MATCH];
  double path[FSDP_PATH_POINTS][4];  /* [spline parameter, x, y, curvature] */
  int32_t n_configs_left, n_configs_right;
  int32_t first_k_left[2], first_k_right[2];
  double best_cost_left, best_cost_right;
  int32_t path_fallback;
  int32_t n_dense; /* number of dense spline samples the outputs (one per row of the horizon) were drawn from */
} fsdp_frame_result;

/* The COMPACT result: what a caller that only drives the car reads of calculate_path_in_global_frame — the path
 * (full_pipeline.py:207, the non-intermediate return value), the sorted cone indices and the status: 1384 bytes per frame
 * instead of 2408 (SURVEY.md 8d counts exactly these bytes as a frame's output).  A result block is what a batch moves back
 * over PCIe; fsdp_submit_compact / fsdp_plan_batch_compact return these records, bit-equal to the same fields of the
 * full record (tests/test_streaming_gpu.py). */
typedef struct {
  double path[FSDP_PATH_POINTS][4];  /* [spline parameter, x, y, curvature], rows beyond the horizon NaN */
  int32_t left_idx[FSDP_MAX_LEN];    /* -1 padded, as in fsdp_frame_result */
  int32_t right_idx[FSDP_MAX_LEN];
  int32_t status;
  uint8_t n_left, n_right;
  uint8_t path_fallback; /* FSDP_FB_* bits */
  uint8_t n_dense;
} fsdp_compact_result;

typedef struct fsdp_ctx fsdp_ctx;

/* The reference's configuration constants — the kwargs of ConeSorting (sorting_cones/core_cone_sorting.py:49-100), ConeMatching
 * (cone_matching/core_cone_matching.py:50-71) and CalculatePath (calculate_path/core_calculate_path.py:69-110), whose defaults
 * are the factories of fsd_path_planning/config.py:28-163.  fsdp_default_params fills those defaults.  A context keeps its own
 * copy on the device.  Every value the reference accepts is accepted, within the compiled capacities of the structural
 * ones: max_n_neighbors <= 5, max_length <= 12, mpc_prediction_horizon <= 40 (8 / 16 / 64 in the wide build; a path of the
 * result holds FSDP_PATH_POINTS rows: with a
 * horizon h below FSDP_PATH_POINTS the rows [h, FSDP_PATH_POINTS) are NaN, and previous paths handed in are read up to row h), max_deg in 1..3 (fits of degree
 * < 3 take the one-frame-per-wavefront path kernel).  use_unknown_cones = 0 drops the cones of type UNKNOWN before sorting
 * (core_cone_sorting.py:113-115); the sorted indices still refer to the caller's array.  matches_should_be_monotonic is the
 * branch of functional_cone_matching.py:164-171 (the pipeline passes False, full_pipeline.py:65; ConeMatching's own default
 * factory passes True, config.py:131-146).  fsdp_create fails on anything outside (no silent substitution). */
typedef struct {
  /* config.py:33-41 get_cone_sorting_config */
  int32_t max_n_neighbors;
  double max_dist;
  double max_dist_to_first;
  int32_t max_length;
  double threshold_directional_angle; /* radians */
  double threshold_absolute_angle;    /* radians */
  int32_t use_unknown_cones;
  /* config.py:48 get_cone_fitting_config */
  double smoothing;
  double predict_every;
  int32_t max_deg;
  /* config.py:55-59 get_path_calculation_config */
  double maximal_distance_for_valid_path;
  double mpc_path_length;
  int32_t mpc_prediction_horizon;
  /* config.py:124-129 get_default_matching_kwargs */
  double min_track_width;
  double max_search_range;
  double max_search_angle; /* radians */
  int32_t matches_should_be_monotonic;
} fsdp_params;
void fsdp_default_params(fsdp_params* out);

const char* fsdp_version(void);
int fsdp_result_size(void);                 /* sizeof(fsdp_frame_result), for binding sanity checks */
/* out4 = [FSDP_MAX_LEN, FSDP_MAX_NEIGHBORS, FSDP_MAX_MATCH, FSDP_PATH_POINTS] of this build: the largest max_length,
 * max_n_neighbors and mpc_prediction_horizon its contexts accept (config.py:34-37,58) and the array shapes of its records. */
void fsdp_shapes(int32_t* out4);
int fsdp_device_count(void);                /* number of visible HIP devices (0 if none)            */

/* PathPlanner(mission) — full_pipeline.py:54-69.  Creates the per-GPU context (device buffers,
 * stream, the constant initial previous path of core_calculate_path.py:103-107 computed on device).
 * mission = a value of utils/mission_types.py; 2 (skidpad) makes the context a set of stateful planner instances
 * (fsdp_skidpad_*), every other mission runs the batch entry points below. */
int fsdp_create(int device, int mission, const fsdp_params* params /* NULL = the reference's defaults */, fsdp_ctx** out);
void fsdp_destroy(fsdp_ctx* ctx);
const char* fsdp_last_error(const fsdp_ctx* ctx); /* ctx may be NULL: last creation error */

/* calculate_path_in_global_frame for a batch of independent frames (fresh-planner semantics).
 * cone_offsets: (n_frames+1) CSR offsets; cones_xyt: (total,3) rows [x,y,ConeTypes] — the reference's own
 * flattened layout (core_trace_sorter.py:37-54); poses: (n_frames,4) rows [px,py,dir_x,dir_y].
 * Frame i owns the rows [cone_offsets[i], cone_offsets[i+1]) of cones_xyt.  cone_offsets[0] need not be 0 (every batch
 * entry point): a SLICE [lo, hi) of a larger batch is handed over as (hi - lo, cone_offsets + lo, cones_xyt, poses + 4 lo,
 * results + lo) — the larger batch's own offsets and cone array, nothing rebased or copied (how multi.py shards one
 * page-locked batch over the GPUs of a node).  Sorted indices of a result stay frame-relative.
 * Host buffers; does H2D, the kernels of a pass and D2H on the context's stream, then synchronises. */
int fsdp_plan_batch(fsdp_ctx* ctx, int n_frames, const int32_t* cone_offsets, const double* cones_xyt,
                    const double* poses, fsdp_frame_result* results);
/* A batch of 16 384 frames or more is pipelined inside the call: it is cut into four contiguous chunks, each on a pass slot
 * (HIP stream) of its own, so that one chunk's transfers run under the other chunks' kernels (pageable buffers: -18 % of the
 * call at 16 384 frames, -26 % at 65 536; below that size chunks only add launches: profiles/r06_plan_probe.txt); with
 * page-locked buffers (fsdp_host_alloc / fsdp_host_register) the chunks are read and written in place.  Results do not
 * depend on the chunking (frames are independent).  The same call with compact records: */
int fsdp_plan_batch_compact(fsdp_ctx* ctx, int n_frames, const int32_t* cone_offsets, const double* cones_xyt,
                            const double* poses, const double* prev_paths /* or NULL */, fsdp_compact_result* results);

/* Sequential-replay form: frame i additionally gets prev_paths[i] (FSDP_PATH_POINTS,4) = the path this planner returned for its previous
 * frame, i.e. CalculatePath.previous_paths[-1] (core_calculate_path.py:203,219-221,236,531-536,568-573).  prev_paths NULL =
 * fresh planners (identical to fsdp_plan_batch).  Typical use: n_frames = number of cars advanced in lock-step. */
int fsdp_plan_batch_sequential(fsdp_ctx* ctx, int n_frames, const int32_t* cone_offsets, const double* cones_xyt,
                               const double* poses, const double* prev_paths, fsdp_frame_result* results);
/* the resident form of the above: applies to the next fsdp_run calls until reset with NULL */
int fsdp_set_previous_paths(fsdp_ctx* ctx, const double* prev_paths);

/* PathPlanner.set_global_path (full_pipeline.py:81-82) / the known path of a relocalized planner (:118-136): with a
 * global path (n,2) the path of every following frame is drawn from it (core_calculate_path.py:514-529: the part within
 * 30 m of the car, rolled to start a third of the table before the closest point) instead of from the matched cones.
 * n = 0 / xy = NULL switches back.  The acceleration / ebs_test missions run on this: their relocalizer (one line fit per
 * planner) is host code, the device gets the transformed pose, empty cone lists and the relocalizer's known path. */
int fsdp_set_global_path(fsdp_ctx* ctx, const double* xy, int n);

/* ---- streams of batches: several DIFFERENT batches in flight -------------------------------------------------------------
 * The reference's only harness feeds the planner a stream of frames, one call after the other (demo/json_demo.py:103-131);
 * the batched form of that stream is a sequence of batches.  A context keeps up to `depth` of them in flight
 * (fsdp_set_overlap): every pass slot owns a HIP stream, device copies of its batch's inputs, the intermediates of a pass
 * and a result block (assembled on the device in the layout of the fsdp_frame_result struct), so
 *   fsdp_submit  = host -> device of the batch, the kernels of its pass, device -> host of its results, all enqueued on
 *                  the slot's stream; returns at once with a ticket;
 *   fsdp_collect = waits for that ticket only; afterwards `results` (the pointer given to fsdp_submit) holds the batch's
 *                  results, exactly what fsdp_plan_batch[_sequential] returns for the same inputs.
 * Tickets count up from 0; a ticket is queued on the slot that holds the fewest (round robin for in-order traffic), behind
 * that slot's previous ticket, two tickets per slot (so a slot's next batch is already waiting on its stream when the
 * current one ends, and the caller's collect / submit round trip costs the GPU nothing): at most fsdp_ticket_capacity =
 * 2 x depth tickets are outstanding (fsdp_submit returns 4 beyond that: collect one first); they may be collected in any
 * order.  The
 * caller's buffers must stay valid and untouched from submit to collect.  For the transfers to be asynchronous they must
 * be page-locked: allocate them with fsdp_host_alloc or pin existing memory with fsdp_host_register — the batch then
 * crosses PCIe inside kernels of the slot's own stream (one reads the inputs from host memory, the last one of the pass
 * writes the results into it; no copy-engine command at all); pageable buffers are accepted (inputs of up to 256 KB — a
 * single frame, a car's handful — are packed into the ticket's own page-locked block by the host and then read in place,
 * larger ones are copied before fsdp_submit returns; results are written by the pass's last kernel into the ticket's
 * page-locked block and copied out by fsdp_collect).  prev_paths: (n_frames,FSDP_PATH_POINTS,4) as for fsdp_plan_batch_sequential, or NULL.
 * While tickets are outstanding the blocking / resident entry points of the context return an error. */
void* fsdp_host_alloc(size_t bytes);            /* page-locked host memory (hipHostMalloc), NULL on failure */
void fsdp_host_free(void* p);
int fsdp_host_register(void* p, size_t bytes);  /* pin memory the caller owns (hipHostRegister) */
int fsdp_host_unregister(void* p);
/* 1 if [p, p + bytes) is page-locked over its whole extent as ONE mapping (what fsdp_submit requires of a buffer before it
 * lets kernels read / write it in place), else 0 — a caller that shards one batch over several contexts (multi.py) asks
 * once and then hands out slices without copying them. */
int fsdp_host_is_pinned(const void* p, size_t bytes);
int fsdp_submit(fsdp_ctx* ctx, int n_frames, const int32_t* cone_offsets, const double* cones_xyt, const double* poses,
                const double* prev_paths, fsdp_frame_result* results, long long* ticket);
/* fsdp_submit with compact records (same tickets, same fsdp_collect) */
int fsdp_submit_compact(fsdp_ctx* ctx, int n_frames, const int32_t* cone_offsets, const double* cones_xyt, const double* poses,
                        const double* prev_paths, fsdp_compact_result* results, long long* ticket);
int fsdp_collect(fsdp_ctx* ctx, long long ticket);
int fsdp_ticket_done(fsdp_ctx* ctx, long long ticket); /* 1: fsdp_collect will not block; 0: still running; -1: unknown ticket */
int fsdp_ticket_capacity(const fsdp_ctx* ctx);         /* tickets that may be outstanding at the current depth */

/* The resident form: one batch stays in HBM and is planned again and again (the benchmark's step; a caller that plans the
 * same frames under several global paths / previous paths). */
int fsdp_upload(fsdp_ctx* ctx, int n_frames, const int32_t* cone_offsets, const double* cones_xyt, const double* poses);
int fsdp_run(fsdp_ctx* ctx);      /* enqueue one pass over the resident batch on the next slot's stream (async) */
int fsdp_sync(fsdp_ctx* ctx);     /* wait for all passes in flight */
int fsdp_resident_frames(const fsdp_ctx* ctx); /* frames of the most recent pass (what fsdp_download writes) */
int fsdp_download(fsdp_ctx* ctx, fsdp_frame_result* results); /* the results of the most recent fsdp_run pass (tickets and blocking calls
                                                                  hand their results to the caller's buffer, not to the slot's block) */

/* Pass overlap: depth d (<= FSDP_MAX_OVERLAP) gives the context d pass slots (HIP stream + buffers each); fsdp_submit
 * tickets and consecutive fsdp_run passes rotate through them, so the next passes fill the compute units that the slowest
 * frames of the previous ones no longer occupy, and one batch's transfers run under the other batches' kernels.
 * fsdp_sync waits for all passes in flight, fsdp_download returns the most recent one.  depth 1 (default) = strictly one
 * pass after the other.  Every stream takes one of the HIP runtime's hardware queues (environment variable
 * GPU_MAX_HW_QUEUES, default 4): with more streams than queues two passes share a queue and serialize — raise
 * GPU_MAX_HW_QUEUES above the depth (bench.py sets 16 for its depth of 10; a skidpad context uses one stream whatever
 * the depth: its slots only hold the steps' buffers).  Every extra depth costs one more set of
 * buffers (~0.1 MB per frame).  Measured at 4096 frames x 128 cones: the steady rate saturates at depth 8; a run of 20
 * passes is fastest with 10 in flight (two full rounds instead of 8 + 8 + 4). */
#define FSDP_MAX_OVERLAP 32
int fsdp_set_overlap(fsdp_ctx* ctx, int depth);

/* The kernels that only serve frames the fast kernels hand on (sort_big_kernel: frames beyond the sorting kernel's LDS
 * capacities; path_retry_kernel: the exact one-frame-per-wavefront path stage) are launched with a pass only when the
 * context expects them to be needed; the last kernel of every pass reports the hand-off lists' lengths, and a pass that
 * needed a kernel it was not given is run again with it before its results are handed out (then the kernel stays part of
 * every pass until 4096 passes in a row did not need it).  Results never depend on this.  Diagnostics: */
int fsdp_route_stats(fsdp_ctx* ctx, int* expect_big, int* expect_retry, long long* reruns);

/* Options of a context: what a test or a measurement may pin instead of leaving it to the library.  Results NEVER depend on
 * them (every route and packing returns the same bits: tests/test_gpu_parity.py
 * test_every_path_kernel_instantiation_equals_oracle).  0 restores the library's own choice.  The library reads no
 * environment variable for any of this (the one it reads: FSDP_RCCL_LIB, the path of librccl.so, csrc/fsdp_comm.h).
 *   "path_mode"       1: the one-kernel path stage (one frame per wavefront) whatever the batch; 2: the three-kernel form
 *   "pack"            1: four frames per wavefront in the three-kernel form; 2: the packed kernels (8 / 16 frames per wavefront)
 *   "fit_g"           4 | 8: lanes per frame of the packed refit kernel (default 4)
 *   "always_route"    1: sort_big_kernel and path_retry_kernel with every pass (default: only when expected, see below)
 *   "no_sort128"      1: the sorting kernel's 255-cone state also for frames of up to 128 cones
 *   "retry_pack_min"  retry lists longer than this run four frames per wavefront in path_retry_kernel (default 512)
 *   "plan_chunks"     k > 1: every blocking call is cut into up to k chunks of >= 512 frames; 1: never (default: four from 16 384 frames)
 *   "poison"          1: every pass first fills its intermediates and scratch with 0xFF bytes (tests: no result depends on what a
 *                     buffer held before — tests/test_streaming_gpu.py)
 *   "skid_group"      steps per launch of a skidpad replay that submits ahead (default: from the instance count)
 *   "skid_pack_min"   (instance, step) pairs from which a group of skidpad steps takes the packed kernels (default 2048)
 * Returns 1 on an unknown name or a value outside its range, or while tickets are outstanding. */
int fsdp_set_option(fsdp_ctx* ctx, const char* name, long long value);

/* Enqueue `iters` back-to-back passes over the resident batch (rotating through the slots when passes overlap; no host
 * synchronisation in between) and time them with HIP events recorded on the streams the kernels run on.
 * ms_total: whole region; ms_stage[FSDP_MAX_STAGES]: summed durations of every kernel of a pass, in launch order (events
 * around each launch; with overlap these include the time a launch shares the chip with the other slots' kernels);
 * unused entries are 0 and fsdp_stage_names names the used ones.  Either pointer may be NULL. */
#define FSDP_MAX_STAGES 8
int fsdp_time_runs(fsdp_ctx* ctx, int iters, float* ms_total, float* ms_stage);
/* The same in three steps, for a caller that puts its own wall clock around the passes: fsdp_time_reserve creates the
 * events of an `iters`-pass region ahead of time (and runs one pass on every slot of the overlap depth that has not run
 * one yet: the first launches on a stream pay for its queue and scratch set-up), fsdp_time_runs(ctx, iters, NULL, NULL)
 * enqueues the passes and returns
 * when the last one has finished (no event is read), fsdp_time_results reads the times of that most recent region. */
int fsdp_time_reserve(fsdp_ctx* ctx, int iters);
/* Which launches the next fsdp_time_runs brackets with events: every kernel (bit 0 set; 1 is the default), or only the main
 * kernel of the path stage — fit_kernel, or path_kernel<64> for small batches — of every pass (bit 0 clear): an event record is
 * a packet in the stream's queue, and seven of them per pass cost about 3 % of the throughput of overlapped passes.  ms_stage
 * entries of kernels that were not bracketed are 0.  Bit 1 (value 2) additionally lets every launch of the refit kernel note
 * its own start / end clock for fsdp_time_kernel_clock (two atomics per workgroup: off by default, so that a region timed
 * without it runs the production launches unchanged).  (Reference: the Timer context managers of full_pipeline.py:113-176 time the
 * stages on the host; they are switched off in the reference's own benchmark runs.) */
int fsdp_time_detail(fsdp_ctx* ctx, int every_kernel);
int fsdp_time_results(fsdp_ctx* ctx, float* ms_total, float* ms_stage);
/* The refit kernel (fit_kernel, the dominant kernel of a large batch) by its own clock: every launch of the most recent timed
 * region notes when its first wavefront started and its last one ended (the device's constant-rate counter, s_memrealtime),
 * i.e. the duration a kernel trace reports — without the time the launch waited in its hardware queue, which an event bracket
 * on the stream includes when twenty streams share the command processor.  *ms_sum = summed duration, *launches = how many
 * launches it covers (0 when the region ran the one-kernel path stage, or was timed without bit 1 of fsdp_time_detail).
 * No counterpart in the reference (measurement only). */
int fsdp_time_kernel_clock(fsdp_ctx* ctx, double* ms_sum, int* launches);
/* What the link between this GPU and the host carries for page-locked buffers of `bytes` bytes, hipMemcpyAsync x iters: host ->
 * device alone, device -> host alone, and each direction with both running at once (GB/s).  Measurement only: the ceiling the
 * rate of a stream of batches (fsdp_submit / fsdp_collect) is held against.  No counterpart in the reference. */
int fsdp_pcie_probe(fsdp_ctx* ctx, size_t bytes, int iters, double* h2d_GBps, double* d2h_GBps, double* both_each_GBps);
/* comma-separated kernel names behind ms_stage of the most recent pass, e.g.
 * "sort_kernel_128,match_kernel<32>,path_prep_kernel<8>,fit_kernel<4>,path_finish_kernel<8>,assemble_kernel" */
int fsdp_stage_names(fsdp_ctx* ctx, char* out, int cap);

/* Stage-level entry points (README "parts of the pipeline are also available as individual classes"). */
/* ConeSorting.run_cone_sorting — fills status, n_left/right, left/right_idx and the sorting diagnostics. */
int fsdp_sort_batch(fsdp_ctx* ctx, int n_frames, const int32_t* cone_offsets, const double* cones_xyt,
                    const double* poses, fsdp_frame_result* results);
/* ConeMatching.run_cone_matching — sorted_left/right: (n_frames,FSDP_MAX_LEN,2) padded, counts (n_frames,). */
int fsdp_match_batch(fsdp_ctx* ctx, int n_frames, const double* sorted_left, const int32_t* n_left,
                     const double* sorted_right, const int32_t* n_right, const double* poses,
                     fsdp_frame_result* results);
/* CalculatePath.run_path_calculation — inputs: the matching fields of `results` (left_v, right_v, l2r, r2l,
 * counts) and poses; prev_paths (n_frames,FSDP_PATH_POINTS,4) = CalculatePath.previous_paths[-1] of every frame's planner, or NULL for
 * fresh planners; fills path, path_fallback, n_dense, status. */
int fsdp_path_batch(fsdp_ctx* ctx, int n_frames, const double* poses, const double* prev_paths, fsdp_frame_result* results);
/* The same call with the SECOND value run_path_calculation returns (calculate_path/core_calculate_path.py:575,
 * `center_along_match_connection`): the points the first spline fit is given — centres of the matched cone pairs
 * (:185-205), the previous path's xy when fewer than two matches exist (:203) or both sides hold fewer than three cones
 * (:531-536), or the slice of the global path within 30 m of the car (:514-529).  centers: (n_frames, centers_cap, 2);
 * n_centers[f] = the number of points of frame f's array (0 when the frame's status is not 0 before the first fit); when
 * it exceeds centers_cap only the first centers_cap points were stored (a global path slice holds up to 1408). */
int fsdp_path_batch_centers(fsdp_ctx* ctx, int n_frames, const double* poses, const double* prev_paths,
                            fsdp_frame_result* results, double* centers, int32_t* n_centers, int centers_cap);

/* ---- skidpad mission (BASELINE config 5): stateful planner instances ------------------------------------------------
 * PathPlanner(MissionTypes.skidpad) keeps state across calls (relocalizer transform, SkidpadCalculatePath.index_along_path,
 * previous path: full_pipeline.py:118-194, relocalization/, calculate_path/skidpad_calculate_path.py).  A context created
 * with mission = 2 holds n_instances independent planners; one fsdp_skidpad_step call = one
 * calculate_path_in_global_frame call of every instance (instance i gets frame i of the batch). */
typedef struct {
  int32_t relocalized;       /* Relocalizer.is_relocalized after this call                                  */
  int32_t index_along_path;  /* SkidpadCalculatePath.index_along_path                                       */
  double translation[2];     /* PathPlanner.relocalization_info.translation (relocalization_information.py) */
  double rotation;           /*                              .rotation                                        */
} fsdp_skidpad_info;

/* Constant inputs (data, not code): table_xy = the known skidpad path BASE_SKIDPAD_PATH (n_table,2)
 * (relocalization/skidpad/skidpad_path_data.py:10-5799); noise_randn = numpy.random.RandomState(42).randn(1140,3,2) flattened
 * (circle_fit_powerset re-seeds on every call, skidpad_relocalizer.py:38,52).  What the reference derives from the table
 * is derived here on the device: the two reference circle centres (calculate_reference_centers_for_skidpad_path,
 * skidpad_relocalizer.py:172-183 -> utils/math_utils.py:579-646 hyper circle fit) and the table spacing
 * (skidpad_calculate_path.py:58). */
int fsdp_skidpad_set_tables(fsdp_ctx* ctx, const double* table_xy, int n_table, const double* noise_randn, int n_noise);
/* out5 = [right centre xy, left centre xy, mean spacing] as the device computed them */
int fsdp_skidpad_constants(fsdp_ctx* ctx, double* out5);
/* (re)create n_instances fresh planners (PathPlanner.__init__ / "reset = construct a new object", README.md:153-154) */
int fsdp_skidpad_reset(fsdp_ctx* ctx, int n_instances);
/* one frame for every instance; results[i].path is in the caller's (original) frame like the reference's return value */
int fsdp_skidpad_step(fsdp_ctx* ctx, int n_instances, const int32_t* cone_offsets, const double* cones_xyt,
                      const double* poses, fsdp_frame_result* results, fsdp_skidpad_info* info);
/* The same as a ticket (fsdp_collect / fsdp_ticket_done as above; one ticket per pass slot, fsdp_set_overlap).  A replay
 * knows the frames of the next steps ahead of the planner, which is what makes submitting ahead meaningful: consecutive
 * steps of a planner then share their launches.  A step's inputs and its relocalization attempt are enqueued at once, its
 * path stage when enough steps have been submitted (half the slots at most) or somebody asks for it (fsdp_collect,
 * fsdp_ticket_done, any blocking call):
 *   - from 2048 (instance, step) pairs (option "skid_pack_min") the pairs are frames of the packed kernels of the autocross
 *     path stage — a planner's window index depends on the poses alone (skidpad_calculate_path.py:60-67), so every step's
 *     window is known up front — and one wavefront per planner then takes its steps in order, keeps the packed result or,
 *     where the step needs the planner's previous path (too far from the car, the ValueError retry) or left the packed
 *     kernels' envelope, plans it itself, and moves the state on;
 *   - below that every (instance, step) pair gets a wavefront of its own in one launch: it plans from the window index
 *     its predecessors' poses lead to, waits for its predecessor's published state and keeps its result unless it read
 *     the previous path.
 * Once a collected step's planner information (info != NULL) has shown every planner relocalized, later steps leave their
 * cones on the host and skip the relocalization attempt: it would return at once (relocalization_base_class.py:56-57), and
 * with a relocalizer the reference neither sorts nor matches (full_pipeline.py:122-140).
 * Either way results, planner information and states are those of one launch per step, bit for bit
 * (tests/test_skidpad_gpu.py, tests/test_skidpad_cpu.py).  A live car submits and collects one step at a time:
 * fsdp_skidpad_step = submit + collect = one launch per step. */
int fsdp_skidpad_submit(fsdp_ctx* ctx, int n_instances, const int32_t* cone_offsets, const double* cones_xyt,
                        const double* poses, fsdp_frame_result* results, fsdp_skidpad_info* info, long long* ticket);
/* The same step with COMPACT results: what a skidpad step produces is its path and status — sorting and matching are skipped
 * (full_pipeline.py:138-140: the intermediates are empty arrays) — so a replay may ask for fsdp_path_result records (1296 bytes
 * per planner and step instead of the 2408 of fsdp_frame_result: the result block is what a step moves over PCIe).  Same
 * tickets, same fsdp_collect; results may be page-locked (written by a kernel of the context's stream) or pageable. */
typedef struct {
  double path[FSDP_PATH_POINTS][4]; /* [spline parameter, x, y, curvature], rows beyond the horizon NaN */
  int32_t status;
  int32_t path_fallback;
  int32_t n_dense;
  int32_t pad;
} fsdp_path_result;
int fsdp_skidpad_submit_compact(fsdp_ctx* ctx, int n_instances, const int32_t* cone_offsets, const double* cones_xyt,
                                const double* poses, fsdp_path_result* results, fsdp_skidpad_info* info, long long* ticket);
/* time `iters` repetitions of the one-wavefront-per-planner path kernel on the last step's inputs with HIP events (the
 * states are restored afterwards) */
int fsdp_skidpad_time_path(fsdp_ctx* ctx, int iters, float* ms_total);
/* Measurement of a replay submitted ahead: with enable != 0 every group of steps that goes through the packed path-stage
 * kernels is bracketed by HIP events on the context's stream (six per group); fsdp_skidpad_group_times waits for the stream
 * and returns the summed durations ms5 = [select | prep | fit | finish | commit], the number of groups, the (instance, step)
 * pairs they held and the kernels' names (comma separated).  enable = 0 / a new enable drops the events. */
int fsdp_skidpad_time_groups(fsdp_ctx* ctx, int enable);
int fsdp_skidpad_group_times(fsdp_ctx* ctx, float* ms5, int* n_groups, long long* n_frames, char* names, int names_cap);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI ---------------------------------------------------------------------
 * The reference is single-process Python with no distributed layer ("distributed communication backend: none");
 * frames are independent, so ranks shard them with NO data-path collective (contiguous frame ranges).  RCCL
 * carries only the start-up broadcast of constant tables — the skidpad track table of
 * relocalization/skidpad/skidpad_path_data.py:10-5799 (root rank loads it, the others receive it), the consistency check
 * of the constant previous path (core_calculate_path.py:103-107) — and the timing barrier / reductions of a benchmark.
 * Usage: rank 0 calls fsdp_comm_unique_id and hands the 128 bytes to every rank out of band (the Python host: one TCP
 * exchange at MASTER_ADDR:MASTER_PORT+1, dist.py); every rank then calls fsdp_comm_init on its context.  Collectives run
 * on the context's own stream and return when complete on this rank.  librccl is dlopen'ed on first use. */
#define FSDP_COMM_ID_BYTES 128
int fsdp_comm_unique_id(void* out128);                                          /* ncclGetUniqueId                      */
int fsdp_comm_init(fsdp_ctx* ctx, int rank, int world, const void* id128);      /* ncclCommInitRank on the ctx's GPU    */
int fsdp_comm_size(fsdp_ctx* ctx);                                              /* ncclCommCount (0 = no communicator)  */
int fsdp_comm_rank(fsdp_ctx* ctx);                                              /* ncclCommUserRank (-1 = none)         */
int fsdp_comm_broadcast(fsdp_ctx* ctx, void* host_buf, size_t bytes, int root); /* ncclBroadcast of a host buffer       */
int fsdp_comm_allreduce(fsdp_ctx* ctx, double* values, int n, int op);          /* in place; op 0 sum, 1 max, 2 min     */
int fsdp_comm_barrier(fsdp_ctx* ctx);        /* waits for this rank's passes in flight, then an all-reduce rendezvous */
int fsdp_comm_destroy(fsdp_ctx* ctx);        /* also done by fsdp_destroy */

/* Per-stage intermediate of the path stage (tests): the smoothing spline of the refit (fit #2,
 * core_calculate_path.py:239-259 -> utils/spline_fit.py:117 splprep) of every frame of the most recent pass that went
 * through the three-kernel path stage: n_knots (n_frames) (-1: the frame took another route), knots (n_frames,34),
 * coefficients (n_frames,68) = x coefficients [0,n) then y coefficients [n,2n). */
int fsdp_debug_refit(fsdp_ctx* ctx, int32_t* n_knots, double* knots34, double* coeffs68);

/* Raw doubles [offset, offset + count) of frame `frame`'s scratch arena after the most recent pass (tests, debug builds). */
int fsdp_debug_arena(fsdp_ctx* ctx, int frame, int offset, int count, double* out);

/* Self-test of the device's hand-rolled FP64 sequences against the compiler's IEEE operations (n elements each; out5n =
 * [sqrt_1_2(x) | sqrt(x) | fast quotient a/b | IEEE a/b | operands inside the fast division's exponent band]): the spline
 * kernels replace sqrt on [1, 2] and divisions of safe-band operands by shorter sequences that must return the same bits. */
int fsdp_selftest_math(fsdp_ctx* ctx, int n, const double* x, const double* a, const double* b, double* out5n);

/* max(|a|, b) and min(|a|, b) as the Givens step of the spline kernels takes them (one v_max_f64 / v_min_f64 with an
 * absolute-value source modifier: spline_device.h max_abs_nn / min_abs_nn): out2n = [max (n) | min (n)]. */
int fsdp_selftest_absminmax(fsdp_ctx* ctx, int n, const double* a, const double* b, double* out2n);

/* FITPACK's fpgivs (the plane rotation every data row of a spline fit goes through four times: utils/spline_fit.py:117 -> splprep ->
 * fppara / fpgivs) as the spline kernels compute it — max / min instead of the branch, scaling-free quotients, the reciprocal of the
 * new diagonal seeded from the square root's own iterate (spline_device.h givens_dd_rd) — next to the same routine with the
 * compiler's IEEE division and square root: out7n = [cs | sn | dd] of the kernels' sequence, [cs | sn | dd] IEEE, [1 where the operands
 * lie inside the sequence's exponent band (outside it the kernels re-plan the frame with the IEEE operations)].  piv: pivots, ww >= 0:
 * diagonals. */
int fsdp_selftest_givens(fsdp_ctx* ctx, int n, const double* piv, const double* ww, double* out7n);

/* The device's restatement of numpy.linalg.det for three homogeneous points (calculate_path/path_parameterization.py:86-92
 * takes the curvature's sign from it): xy6 = (n,6) rows x0,y0,x1,y1,x2,y2 -> out (n) determinants whose SIGN is NumPy's. */
int fsdp_selftest_det3(fsdp_ctx* ctx, int n, const double* xy6, double* out);

/* The device's libm where discrete decisions of the sorting stage take its values (the reference compares angles from
 * np.arctan2 / np.arccos: sorting_cones/trace_sorter/end_configurations.py:108-223, cost_function.py:40-120,
 * core_trace_sorter.py:344-377): out3n = [atan2(y, x) of the ROCm device library | det_atan2(y, x), the correctly rounded
 * value (csrc/det_math.h) | acos(cs)].  A test holds the first to within 2 ulp of the second on 10^6 arguments (fewer than 1 % of
 * them at 2 ulp) and the third to within 2 ulp of the host's, so that a ROCm release that moves them cannot move a sorted index silently. */
int fsdp_selftest_libm(fsdp_ctx* ctx, int n, const double* y, const double* x, const double* cs, double* out3n);

/* The constant initial previous path (core_calculate_path.py:103-107), (FSDP_PATH_POINTS,4) — rows beyond the horizon NaN —, as computed on the device. */
int fsdp_default_path(fsdp_ctx* ctx, double* out40x4);

#ifdef __cplusplus
}
#endif
#endif /* FSDP_H */
