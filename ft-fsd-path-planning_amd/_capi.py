"""ctypes binding of libfsdp_hip.so (C ABI: include/fsdp.h).

There is deliberately NO fallback: if the HIP library is missing, or no GPU is visible, the
calls below raise.  (The CPU oracle under oracle/ is test infrastructure and is never imported
from here.)
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

# multi-process GPU work (RCCL): the host driver only supports dmabuf IPC; must be in the environment before the HIP
# runtime starts in this process
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# every pass in flight (Context.set_overlap) runs on its own HIP stream; streams beyond the runtime's hardware queues
# (default 4) share a queue and serialize (20 passes in flight on 16 queues: 5.0 M frames/s, on 20-32: 5.8 M; the queues of all processes
# on one GPU add up: two processes with 24 each crawl, profiles/r04_ab_variants.txt 7) — also read when the HIP runtime starts
os.environ.setdefault("GPU_MAX_HW_QUEUES", "22")

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = PKG_DIR / "lib" / "libfsdp_hip.so"

MAX_CONES = 8192
MAX_STAGES = 8  # FSDP_MAX_STAGES


class Shapes:
    """The array shapes of one build of the library (include/fsdp.h: FSDP_MAX_LEN, FSDP_MAX_NEIGHBORS, FSDP_MAX_MATCH,
    FSDP_PATH_POINTS) with the NumPy mirrors of its records.  Two builds of the same sources exist: the standard one (the
    reference's default structural parameters, config.py:34-37,58) and the wide one for contexts whose max_n_neighbors,
    max_length or mpc_prediction_horizon exceed them."""

    def __init__(self, name, lib_name, max_len, max_neighbors, max_match, path_points):
        self.name, self.lib_path = name, PKG_DIR / "lib" / lib_name
        self.max_len, self.max_neighbors, self.max_match, self.path_points = max_len, max_neighbors, max_match, path_points
        # numpy mirror of fsdp_frame_result (include/fsdp.h)
        self.result_dtype = np.dtype(
            [
                ("status", "<i4"),
                ("n_left", "<i4"),
                ("n_right", "<i4"),
                ("left_idx", "<i4", (max_len,)),
                ("right_idx", "<i4", (max_len,)),
                ("n_left_v", "<i4"),
                ("n_right_v", "<i4"),
                ("left_v", "<f8", (max_match, 2)),
                ("right_v", "<f8", (max_match, 2)),
                ("l2r", "<i4", (max_match,)),
                ("r2l", "<i4", (max_match,)),
                ("path", "<f8", (path_points, 4)),
                ("n_configs_left", "<i4"),
                ("n_configs_right", "<i4"),
                ("first_k_left", "<i4", (2,)),
                ("first_k_right", "<i4", (2,)),
                ("best_cost_left", "<f8"),
                ("best_cost_right", "<f8"),
                ("path_fallback", "<i4"),
                ("n_dense", "<i4"),
            ],
            align=True,
        )
        # fsdp_path_result (include/fsdp.h): a skidpad step's compact result
        self.path_result_dtype = np.dtype(
            [("path", "<f8", (path_points, 4)), ("status", "<i4"), ("path_fallback", "<i4"), ("n_dense", "<i4"), ("pad", "<i4")], align=True)
        # fsdp_compact_result (include/fsdp.h): path + sorted indices + status of a trackdrive frame (1384 bytes in the standard build)
        self.compact_dtype = np.dtype(
            [("path", "<f8", (path_points, 4)), ("left_idx", "<i4", (max_len,)), ("right_idx", "<i4", (max_len,)), ("status", "<i4"),
             ("n_left", "u1"), ("n_right", "u1"), ("path_fallback", "u1"), ("n_dense", "u1")], align=True)

    def holds(self, params) -> bool:
        """Do this build's shapes take these structural parameters (an fsdp_params)?"""
        return (params.max_n_neighbors <= self.max_neighbors and params.max_length <= self.max_len
                and params.mpc_prediction_horizon <= self.path_points)


STANDARD = Shapes("standard", "libfsdp_hip.so", 12, 5, 24, 40)
WIDE = Shapes("wide", "libfsdp_hip_wide.so", 16, 8, 32, 64)  # libfsdp_hip_wide.so = the same sources with -DFSDP_WIDE_SHAPES
LIB_PATH = STANDARD.lib_path
# the standard build's shapes under their historic names (what every context with the reference's default structural
# parameters uses; a context knows its own: Context.shapes / Context.result_dtype)
MAX_LEN, MAX_MATCH, PATH_POINTS = STANDARD.max_len, STANDARD.max_match, STANDARD.path_points
RESULT_DTYPE = STANDARD.result_dtype
PATH_RESULT_DTYPE = STANDARD.path_result_dtype
COMPACT_DTYPE = STANDARD.compact_dtype

# Options every new Context applies on top of the library's defaults (fsdp_set_option, include/fsdp.h): the hook tests and
# measurement tools use to pin a route or a packing (monkeypatch.setitem(_capi.DEFAULT_OPTIONS, "pack", 2)).  Results never
# depend on them.  The library itself reads no environment variable for this; tools that are driven from a shell translate
# theirs with options_from_env().
DEFAULT_OPTIONS: dict = {}
OPTION_NAMES = ("path_mode", "pack", "fit_g", "always_route", "no_sort128", "retry_pack_min", "plan_chunks", "poison", "skid_group", "skid_pack_min")


def options_from_env(env=None) -> dict:
    """Measurement tools only (tools/*.py, tests/fuzz_*.py): FSDP_PATH_MODE=mono|split, FSDP_PACK=0|1, FSDP_FIT_G=4|8,
    FSDP_ALWAYS_ROUTE, FSDP_NO_SORT128, FSDP_RETRY_PACK_MIN, FSDP_PLAN_CHUNKS, FSDP_SKID_GROUP, FSDP_SKID_PACK_MIN -> the options of
    fsdp_set_option.  The product never calls this."""
    env = os.environ if env is None else env
    out = {}
    if env.get("FSDP_PATH_MODE") in ("mono", "split"):
        out["path_mode"] = 1 if env["FSDP_PATH_MODE"] == "mono" else 2
    if "FSDP_PACK" in env:
        out["pack"] = 2 if int(env["FSDP_PACK"]) else 1
    for name in ("fit_g", "retry_pack_min", "plan_chunks", "poison", "skid_group", "skid_pack_min"):
        if f"FSDP_{name.upper()}" in env:
            out[name] = int(env[f"FSDP_{name.upper()}"])
    for name in ("always_route", "no_sort128"):
        if f"FSDP_{name.upper()}" in env:
            out[name] = 1
    return out


class FsdpError(RuntimeError):
    pass


class Params(ctypes.Structure):
    """fsdp_params (include/fsdp.h): the kwargs of the reference's ConeSorting / ConeMatching / CalculatePath
    (defaults: fsd_path_planning/config.py:28-163, filled by fsdp_default_params)."""

    _fields_ = [
        ("max_n_neighbors", ctypes.c_int32), ("max_dist", ctypes.c_double), ("max_dist_to_first", ctypes.c_double),
        ("max_length", ctypes.c_int32), ("threshold_directional_angle", ctypes.c_double), ("threshold_absolute_angle", ctypes.c_double),
        ("use_unknown_cones", ctypes.c_int32),
        ("smoothing", ctypes.c_double), ("predict_every", ctypes.c_double), ("max_deg", ctypes.c_int32),
        ("maximal_distance_for_valid_path", ctypes.c_double), ("mpc_path_length", ctypes.c_double), ("mpc_prediction_horizon", ctypes.c_int32),
        ("min_track_width", ctypes.c_double), ("max_search_range", ctypes.c_double), ("max_search_angle", ctypes.c_double),
        ("matches_should_be_monotonic", ctypes.c_int32),
    ]


PARAM_NAMES = [f[0] for f in Params._fields_]


def make_params(overrides=None) -> Params:
    """Defaults from the library, then `overrides` (a dict of the reference's kwarg names); unknown names raise."""
    p = Params()
    load().fsdp_default_params(ctypes.byref(p))
    for k, v in (overrides or {}).items():
        if k == "experimental_performance_improvements":
            if v:
                raise NotImplementedError("the experimental sorting cache is out of scope (SURVEY.md section 2 row 15)")
            continue
        if k not in PARAM_NAMES:
            raise TypeError(f"unknown parameter {k!r}")
        setattr(p, k, type(getattr(p, k))(v))
    return p


_libs = {}


def load(shapes: Shapes = STANDARD) -> ctypes.CDLL:
    """Load libfsdp_hip.so (or, for shapes = WIDE, libfsdp_hip_wide.so) or raise FsdpError (no CPU fallback exists)."""
    if shapes.name in _libs:
        return _libs[shapes.name]
    path = LIB_PATH if shapes is STANDARD else shapes.lib_path  # (LIB_PATH: what the A/B tools point at an experiment build)
    if not path.exists():
        raise FsdpError(
            f"{path} not found: build it with `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950); this package has no CPU fallback"
        )
    lib = ctypes.CDLL(str(path))
    lib.fsdp_version.restype = ctypes.c_char_p
    lib.fsdp_last_error.restype = ctypes.c_char_p
    lib.fsdp_last_error.argtypes = [ctypes.c_void_p]
    lib.fsdp_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
    lib.fsdp_destroy.argtypes = [ctypes.c_void_p]
    lib.fsdp_resident_frames.argtypes = [ctypes.c_void_p]
    lib.fsdp_stage_names.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
    lib.fsdp_time_runs.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.fsdp_time_reserve.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.fsdp_time_detail.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.fsdp_time_results.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.fsdp_time_kernel_clock.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.fsdp_comm_unique_id.argtypes = [ctypes.c_void_p]
    lib.fsdp_comm_init.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]
    lib.fsdp_comm_broadcast.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    lib.fsdp_comm_allreduce.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int]
    for name in ("fsdp_comm_size", "fsdp_comm_rank", "fsdp_comm_barrier", "fsdp_comm_destroy"):
        getattr(lib, name).argtypes = [ctypes.c_void_p]
    lib.fsdp_host_alloc.restype = ctypes.c_void_p
    lib.fsdp_host_alloc.argtypes = [ctypes.c_size_t]
    lib.fsdp_host_free.argtypes = [ctypes.c_void_p]
    lib.fsdp_host_register.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    lib.fsdp_host_unregister.argtypes = [ctypes.c_void_p]
    lib.fsdp_host_is_pinned.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    lib.fsdp_submit.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong)]
    lib.fsdp_submit_compact.argtypes = lib.fsdp_submit.argtypes
    lib.fsdp_plan_batch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.fsdp_plan_batch_sequential.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.fsdp_plan_batch_compact.argtypes = lib.fsdp_plan_batch_sequential.argtypes
    lib.fsdp_set_option.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_longlong]
    lib.fsdp_pcie_probe.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                    ctypes.POINTER(ctypes.c_double)]
    lib.fsdp_collect.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
    lib.fsdp_ticket_done.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
    lib.fsdp_ticket_capacity.argtypes = [ctypes.c_void_p]
    lib.fsdp_skidpad_submit_compact.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong)]
    lib.fsdp_skidpad_submit.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong)]
    lib.fsdp_route_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_longlong)]
    got = (ctypes.c_int32 * 4)()
    lib.fsdp_shapes(got)
    want = [shapes.max_len, shapes.max_neighbors, shapes.max_match, shapes.path_points]
    assert shapes.compact_dtype.itemsize == 32 * shapes.path_points + 8 * shapes.max_len + 8
    if list(got) != want or lib.fsdp_result_size() != shapes.result_dtype.itemsize:
        raise FsdpError(f"{path.name}: shapes {list(got)} / result size {lib.fsdp_result_size()}, this binding expects {want} / {shapes.result_dtype.itemsize}")
    _libs[shapes.name] = lib
    return lib


EXPORTED_SYMBOLS = [
    "fsdp_version", "fsdp_result_size", "fsdp_shapes", "fsdp_device_count", "fsdp_default_params", "fsdp_create", "fsdp_destroy", "fsdp_last_error",
    "fsdp_plan_batch", "fsdp_upload", "fsdp_run", "fsdp_sync", "fsdp_download", "fsdp_time_runs", "fsdp_time_reserve", "fsdp_time_results", "fsdp_time_kernel_clock", "fsdp_time_detail", "fsdp_stage_names", "fsdp_resident_frames",
    "fsdp_sort_batch", "fsdp_match_batch", "fsdp_path_batch", "fsdp_path_batch_centers", "fsdp_default_path",
    "fsdp_plan_batch_sequential", "fsdp_set_previous_paths", "fsdp_set_overlap", "fsdp_set_global_path",
    "fsdp_comm_unique_id", "fsdp_comm_init", "fsdp_comm_size", "fsdp_comm_rank", "fsdp_comm_broadcast", "fsdp_comm_allreduce",
    "fsdp_comm_barrier", "fsdp_comm_destroy", "fsdp_selftest_math", "fsdp_debug_refit",
    "fsdp_skidpad_set_tables", "fsdp_skidpad_constants", "fsdp_skidpad_reset", "fsdp_skidpad_step", "fsdp_skidpad_time_path", "fsdp_skidpad_time_groups", "fsdp_skidpad_group_times", "fsdp_skidpad_submit_compact",
    "fsdp_host_alloc", "fsdp_host_free", "fsdp_host_register", "fsdp_host_unregister", "fsdp_host_is_pinned", "fsdp_submit", "fsdp_collect", "fsdp_ticket_done",
    "fsdp_submit_compact", "fsdp_plan_batch_compact", "fsdp_set_option", "fsdp_pcie_probe",
    "fsdp_skidpad_submit", "fsdp_route_stats", "fsdp_ticket_capacity", "fsdp_selftest_det3", "fsdp_debug_arena", "fsdp_selftest_absminmax", "fsdp_selftest_libm", "fsdp_selftest_givens",
]


def host_lib() -> ctypes.CDLL:
    """The library the host-memory helpers go through: any build that is already loaded (a process whose contexts all run on
    the wide build never needs libfsdp_hip.so for them), else the standard one.  Page-locked memory is a property of the HIP
    runtime, which the builds share."""
    for lib in _libs.values():
        return lib
    return load()


def pinned_empty(shape, dtype=np.float64) -> np.ndarray:
    """A NumPy array in page-locked host memory (fsdp_host_alloc): what Context.submit can copy to / from asynchronously.
    Freed when the array (and every view of it) is gone."""
    import weakref

    lib = host_lib()
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) if np.ndim(shape) else int(shape)
    nbytes = max(1, n * dtype.itemsize)
    ptr = lib.fsdp_host_alloc(nbytes)
    if not ptr:
        raise FsdpError(f"fsdp_host_alloc({nbytes}) failed")
    buf = (ctypes.c_byte * nbytes).from_address(ptr)
    weakref.finalize(buf, lib.fsdp_host_free, ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)


def is_pinned(a: np.ndarray) -> bool:
    """Is the whole extent of this (contiguous) array page-locked as one mapping, i.e. will fsdp_submit let its kernels read /
    write it in place?"""
    return bool(a.flags.c_contiguous and (a.nbytes == 0 or host_lib().fsdp_host_is_pinned(ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(a.nbytes))))


class PinnedPool:
    """Page-locked arrays that go back to a free list — not to hipHostFree — when the last reference to them is dropped.
    hipHostMalloc costs about a millisecond for a 10 MB result block, a batch of 4096 frames takes 0.9 ms on the GPU: a stream of
    batches must not allocate.  get() hands out an ordinary NumPy array (no copy is ever made of it); whoever holds it, or a view
    of it, keeps the block."""

    def __init__(self):
        import threading

        self._free = {}  # nbytes -> [address]
        self._lock = threading.Lock()
        self._lib = host_lib()
        self._closed = False

    def _put(self, nbytes, ptr):
        with self._lock:
            if not self._closed:
                self._free.setdefault(nbytes, []).append(ptr)
                return
        self._lib.fsdp_host_free(ptr)  # (an array that outlived its pool)

    def get(self, n: int, dtype) -> np.ndarray:
        import weakref

        dtype = np.dtype(dtype)
        want = max(1, int(n) * dtype.itemsize)
        nbytes = 1 << (want - 1).bit_length()  # size classes: a stream of ragged batches reuses its blocks
        with self._lock:
            lst = self._free.get(nbytes)
            ptr = lst.pop() if lst else None
        if ptr is None:
            ptr = self._lib.fsdp_host_alloc(nbytes)
            if not ptr:
                raise FsdpError(f"fsdp_host_alloc({nbytes}) failed")
        buf = (ctypes.c_byte * nbytes).from_address(ptr)
        weakref.finalize(buf, self._put, nbytes, ptr)
        return np.frombuffer(buf, dtype=dtype, count=int(n))

    def close(self):
        with self._lock:
            blocks, self._free = self._free, {}
            self._closed = True
        for lst in blocks.values():
            for ptr in lst:
                self._lib.fsdp_host_free(ptr)


def pinned_copy(a, dtype=None) -> np.ndarray:
    a = np.asarray(a, dtype=dtype)
    out = pinned_empty(a.shape, a.dtype)
    out[...] = a
    return out


class Ticket:
    """One batch in flight (Context.submit): keeps the buffers alive until it is collected."""

    __slots__ = ("id", "out", "info", "_keep")

    def __init__(self, id_, out, info, keep):
        self.id, self.out, self.info, self._keep = id_, out, info, keep



def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


class Context:
    """One GPU context (= fsdp_ctx): device buffers + one HIP stream."""

    def __init__(self, device: int | None = None, mission: int = 4, params: dict | None = None, shapes: Shapes | None = None,
                 options: dict | None = None):
        """params: overrides of the reference's configuration constants by their kwarg names (None = defaults).
        options: fsdp_set_option knobs for tests / measurements (on top of _capi.DEFAULT_OPTIONS); results never depend on them.
        shapes: which build of the library carries the context (None = the standard build unless max_n_neighbors, max_length or
        mpc_prediction_horizon exceed its shapes — then the wide build, whose records are larger: Context.result_dtype)."""
        self.params = make_params(params)
        if shapes is None:
            shapes = STANDARD if STANDARD.holds(self.params) else WIDE
        lib = load(shapes)
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0")) % max(lib.fsdp_device_count(), 1)
        h = ctypes.c_void_p()
        rc = lib.fsdp_create(int(device), int(mission), ctypes.byref(self.params), ctypes.byref(h))
        if rc != 0:
            raise FsdpError(f"fsdp_create failed ({rc}): {lib.fsdp_last_error(None).decode()}")
        self._lib, self._h, self.device, self.n_frames = lib, h, device, 0
        self.shapes, self.result_dtype, self.path_result_dtype = shapes, shapes.result_dtype, shapes.path_result_dtype
        self.compact_dtype = shapes.compact_dtype
        for k, v in {**DEFAULT_OPTIONS, **(options or {})}.items():
            self.set_option(k, v)

    def set_option(self, name: str, value: int):
        """fsdp_set_option (include/fsdp.h): pin a route / packing for a test or a measurement; 0 = the library's choice."""
        self._check(self._lib.fsdp_set_option(self._h, name.encode(), int(value)), f"fsdp_set_option({name})")

    @property
    def horizon(self) -> int:
        """mpc_prediction_horizon of this context: rows of a path (the result struct holds shapes.path_points, the rest NaN)."""
        return int(self.params.mpc_prediction_horizon)

    def pad_paths(self, paths) -> np.ndarray:
        """(n, horizon, 4) previous paths as the reference keeps them -> the (n, shapes.path_points, 4) block the C ABI reads."""
        rows = self.shapes.path_points
        p = np.asarray(paths, dtype=np.float64)
        p = p.reshape(-1, p.shape[-2], 4)
        if p.shape[1] == rows:
            return np.ascontiguousarray(p)
        out = np.full((len(p), rows, 4), np.nan)
        out[:, : p.shape[1]] = p
        return out

    def _check(self, rc, what):
        if rc != 0:
            raise FsdpError(f"{what} failed ({rc}): {self._lib.fsdp_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.fsdp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _prep(offsets, cones, poses):
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        cones = np.ascontiguousarray(cones, dtype=np.float64).reshape(-1, 3)
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 4)
        n = len(offsets) - 1
        if n < 0 or len(poses) != n or (n > 0 and (int(offsets[0]) != 0 or int(offsets[-1]) != len(cones))):
            raise ValueError("inconsistent batch: offsets / cones / poses")
        if n > 0 and np.any(np.diff(offsets) < 0):
            raise ValueError("cone_offsets must be non-decreasing")
        return offsets, cones, poses, n

    def plan_batch(self, offsets, cones, poses, prev_paths=None, out=None, compact: bool = False) -> np.ndarray:
        """One blocking call: the batch in, its results out (a batch of 16 384 frames or more is pipelined in chunks inside the
        library).  prev_paths: per-frame previous paths (plan_batch_sequential).  out: the array the results go to — a page-locked
        one (pinned_empty) is written in place by the GPU.  compact: fsdp_compact_result records (path, sorted indices, status:
        self.compact_dtype) instead of the full ones."""
        offsets, cones, poses, n = self._prep(offsets, cones, poses)
        prev = None if prev_paths is None else self.pad_paths(prev_paths)
        assert prev is None or len(prev) == n
        dt = self.compact_dtype if compact else self.result_dtype
        if out is None:
            out = np.zeros(n, dtype=dt)
        assert out.dtype == dt and len(out) == n and out.flags.c_contiguous
        pp = None if prev is None else prev.ctypes.data
        cp = cones.ctypes.data if len(cones) else None
        if compact:
            self._check(self._lib.fsdp_plan_batch_compact(self._h, n, offsets.ctypes.data, cp, poses.ctypes.data, pp, out.ctypes.data), "fsdp_plan_batch_compact")
        elif prev is not None:
            self._check(self._lib.fsdp_plan_batch_sequential(self._h, n, offsets.ctypes.data, cp, poses.ctypes.data, pp, out.ctypes.data), "fsdp_plan_batch_sequential")
        else:
            self._check(self._lib.fsdp_plan_batch(self._h, n, offsets.ctypes.data, cp, poses.ctypes.data, out.ctypes.data), "fsdp_plan_batch")
        self.n_frames = n
        return out

    def plan_batch_sequential(self, offsets, cones, poses, prev_paths) -> np.ndarray:
        """plan_batch with a per-frame previous path (n_frames,40,4): the stateful fallbacks of the reference."""
        return self.plan_batch(offsets, cones, poses, prev_paths=prev_paths)

    def sort_batch(self, offsets, cones, poses) -> np.ndarray:
        offsets, cones, poses, n = self._prep(offsets, cones, poses)
        out = np.zeros(n, dtype=self.result_dtype)
        self._check(self._lib.fsdp_sort_batch(self._h, n, _ip(offsets), _dp(cones), _dp(poses), ctypes.c_void_p(out.ctypes.data)), "fsdp_sort_batch")
        return out

    def match_batch(self, sorted_left, n_left, sorted_right, n_right, poses) -> np.ndarray:
        sorted_left = np.ascontiguousarray(sorted_left, np.float64).reshape(-1, self.shapes.max_len, 2)
        sorted_right = np.ascontiguousarray(sorted_right, np.float64).reshape(-1, self.shapes.max_len, 2)
        n_left = np.ascontiguousarray(n_left, np.int32)
        n_right = np.ascontiguousarray(n_right, np.int32)
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 4)
        n = len(poses)
        out = np.zeros(n, dtype=self.result_dtype)
        self._check(self._lib.fsdp_match_batch(self._h, n, _dp(sorted_left), _ip(n_left), _dp(sorted_right), _ip(n_right), _dp(poses), ctypes.c_void_p(out.ctypes.data)), "fsdp_match_batch")
        return out

    def path_batch(self, poses, results: np.ndarray, prev_paths=None) -> np.ndarray:
        """prev_paths (n,40,4): CalculatePath.previous_paths[-1] of every frame's planner (None = fresh planners)."""
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 4)
        results = np.ascontiguousarray(results)
        assert results.dtype == self.result_dtype and len(results) == len(poses)
        prev = None if prev_paths is None else _dp(self.pad_paths(prev_paths))
        self._check(self._lib.fsdp_path_batch(self._h, len(poses), _dp(poses), prev, ctypes.c_void_p(results.ctypes.data)), "fsdp_path_batch")
        return results

    def path_batch_centers(self, poses, results: np.ndarray, prev_paths=None, cap: int = 1408):
        """path_batch + the second value of CalculatePath.run_path_calculation (core_calculate_path.py:575): a list of
        (n_i, 2) arrays, the points every frame's first spline fit was given."""
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 4)
        results = np.ascontiguousarray(results)
        assert results.dtype == self.result_dtype and len(results) == len(poses)
        prev = None if prev_paths is None else _dp(self.pad_paths(prev_paths))
        n = len(poses)
        centers = np.zeros((n, cap, 2))
        counts = np.zeros(n, np.int32)
        self._check(self._lib.fsdp_path_batch_centers(self._h, n, _dp(poses), prev, ctypes.c_void_p(results.ctypes.data), _dp(centers), _ip(counts),
                                                      ctypes.c_int(cap)), "fsdp_path_batch_centers")
        if (counts > cap).any():
            raise FsdpError(f"centre points beyond the buffer ({int(counts.max())} > {cap})")
        return results, [centers[i, : counts[i]].copy() for i in range(n)]

    # streams of batches: several different batches in flight (fsdp_submit / fsdp_collect)
    def submit(self, offsets, cones, poses, prev_paths=None, out=None, compact: bool = False) -> Ticket:
        """Enqueue one batch (H2D, the kernels of a pass, D2H) on the next pass slot and return at once.  Arrays made by
        ``pinned_empty`` / ``pinned_copy`` are transferred asynchronously; others are accepted but staged.  ``out``: the
        self.result_dtype (compact: self.compact_dtype) array the results go to (default: a new pinned array).  Raises when
        every slot holds a ticket.  The arrays must stay untouched until collect()."""
        offsets, cones, poses, n = self._prep(offsets, cones, poses)
        prev = None if prev_paths is None else self.pad_paths(prev_paths)
        dt = self.compact_dtype if compact else self.result_dtype
        if out is None:
            out = pinned_empty(n, dt)
        assert out.dtype == dt and len(out) == n and out.flags.c_contiguous
        t = ctypes.c_longlong(-1)
        fn = self._lib.fsdp_submit_compact if compact else self._lib.fsdp_submit
        self._check(fn(self._h, n, offsets.ctypes.data, cones.ctypes.data if len(cones) else None, poses.ctypes.data,
                       None if prev is None else prev.ctypes.data, out.ctypes.data, ctypes.byref(t)), "fsdp_submit")
        return Ticket(int(t.value), out, None, (offsets, cones, poses, prev))

    def submit_slice(self, lo: int, hi: int, offsets, cones, poses, prev, out, compact: bool = False) -> Ticket:
        """Frames [lo, hi) of a batch that is already in the ABI's layout (int32 offsets of the WHOLE batch, (N, 3) cones,
        (F, 4) poses, optional (F, 40, 4) previous paths, self.result_dtype out of the whole batch): the slice goes to fsdp_submit as
        pointers into those arrays (include/fsdp.h: cone_offsets[0] need not be 0) — nothing is copied or rebased on the host."""
        n = hi - lo
        t = ctypes.c_longlong(-1)
        fn = self._lib.fsdp_submit_compact if compact else self._lib.fsdp_submit
        self._check(fn(self._h, n, offsets.ctypes.data + 4 * lo, cones.ctypes.data if len(cones) else None,
                                          poses.ctypes.data + 32 * lo, None if prev is None else prev.ctypes.data + prev.strides[0] * lo,
                                          out.ctypes.data + out.strides[0] * lo, ctypes.byref(t)), "fsdp_submit")
        return Ticket(int(t.value), out, None, (offsets, cones, poses, prev))

    def collect(self, ticket: Ticket) -> np.ndarray:
        """Wait for this ticket only; returns its result array."""
        self._check(self._lib.fsdp_collect(self._h, ctypes.c_longlong(ticket.id)), "fsdp_collect")
        ticket._keep = None
        return ticket.out

    @property
    def ticket_capacity(self) -> int:
        """Tickets that may be outstanding at the current overlap depth (two per pass slot)."""
        return int(self._lib.fsdp_ticket_capacity(self._h))

    def ticket_done(self, ticket: Ticket) -> bool:
        return int(self._lib.fsdp_ticket_done(self._h, ctypes.c_longlong(ticket.id))) == 1

    def route_stats(self):
        """(sort_big_kernel expected, path_retry_kernel expected, passes re-run because a route kernel was missing)."""
        a, b, r = ctypes.c_int(), ctypes.c_int(), ctypes.c_longlong()
        self._check(self._lib.fsdp_route_stats(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(r)), "fsdp_route_stats")
        return bool(a.value), bool(b.value), int(r.value)

    # resident API
    def upload(self, offsets, cones, poses):
        offsets, cones, poses, n = self._prep(offsets, cones, poses)
        self._check(self._lib.fsdp_upload(self._h, n, _ip(offsets), _dp(cones), _dp(poses)), "fsdp_upload")
        self._check(self._lib.fsdp_sync(self._h), "fsdp_sync")
        self.n_frames = n

    def set_global_path(self, xy):
        """PathPlanner.set_global_path: (n,2) array, or None to plan from the matched cones again."""
        if xy is None:
            self._check(self._lib.fsdp_set_global_path(self._h, None, 0), "fsdp_set_global_path")
            return
        xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
        self._check(self._lib.fsdp_set_global_path(self._h, _dp(xy), ctypes.c_int(len(xy))), "fsdp_set_global_path")

    def set_overlap(self, depth: int):
        """depth d (1..32): consecutive run() passes rotate through d HIP streams / buffer sets and overlap (a replay is a
        stream of batches); depth 1 (default): one pass after the other."""
        self._check(self._lib.fsdp_set_overlap(self._h, int(depth)), "fsdp_set_overlap")

    def run(self):
        self._check(self._lib.fsdp_run(self._h), "fsdp_run")

    def sync(self):
        self._check(self._lib.fsdp_sync(self._h), "fsdp_sync")

    def download(self) -> np.ndarray:
        # sized from the library's own count: stage-level calls change the resident batch behind this object's back
        out = np.zeros(int(self._lib.fsdp_resident_frames(self._h)), dtype=self.result_dtype)
        self._check(self._lib.fsdp_download(self._h, ctypes.c_void_p(out.ctypes.data)), "fsdp_download")
        return out

    def time_runs(self, iters: int, collect: bool = True):
        """`iters` back-to-back passes over the resident batch with HIP events around every kernel.  collect=False: only
        enqueue and wait (for a caller's own wall clock); time_results() reads the events afterwards."""
        if not collect:
            self._check(self._lib.fsdp_time_runs(self._h, int(iters), None, None), "fsdp_time_runs")
            return None
        tot = ctypes.c_float()
        st = (ctypes.c_float * MAX_STAGES)()
        self._check(self._lib.fsdp_time_runs(self._h, int(iters), ctypes.byref(tot), st), "fsdp_time_runs")
        n = len(self.stage_names())
        return float(tot.value), [float(x) for x in st][:n]

    def time_reserve(self, iters: int):
        """Create the events of an `iters`-pass timed region ahead of time."""
        self._check(self._lib.fsdp_time_reserve(self._h, int(iters)), "fsdp_time_reserve")

    def time_detail(self, every_kernel: bool, kernel_clock: bool = False):
        """Events around every kernel launch of the next time_runs (True, the default) or only around the path stage's
        main kernel of every pass (False: the other kernels' times come back as 0).  kernel_clock: the refit kernel's launches
        also note their own start / end clock (time_kernel_clock) — two atomics per workgroup, so off unless asked for."""
        self._check(self._lib.fsdp_time_detail(self._h, (1 if every_kernel else 0) | (2 if kernel_clock else 0)), "fsdp_time_detail")

    def time_results(self):
        """(ms of the whole region, summed ms per kernel) of the most recent time_runs."""
        tot = ctypes.c_float()
        st = (ctypes.c_float * MAX_STAGES)()
        self._check(self._lib.fsdp_time_results(self._h, ctypes.byref(tot), st), "fsdp_time_results")
        n = len(self.stage_names())
        return float(tot.value), [float(x) for x in st][:n]

    def time_kernel_clock(self):
        """(summed ms, launches) of the refit kernel in the most recent time_runs by the kernel's own clock readings: first
        wavefront's start to last wavefront's end, what a kernel trace reports (fsdp_time_kernel_clock)."""
        ms = ctypes.c_double()
        n = ctypes.c_int()
        self._check(self._lib.fsdp_time_kernel_clock(self._h, ctypes.byref(ms), ctypes.byref(n)), "fsdp_time_kernel_clock")
        return float(ms.value), int(n.value)

    def stage_names(self):
        """Kernel names behind the per-stage times of the most recent time_runs (the path kernel's lane-group
        instantiation depends on how the batch was launched)."""
        buf = ctypes.create_string_buffer(256)
        self._check(self._lib.fsdp_stage_names(self._h, buf, 256), "fsdp_stage_names")
        return buf.value.decode().split(",")

    def debug_refit(self):
        """(n_knots (n,), knots (n,34), coefficients (n,68)) of the refit of every frame of the most recent pass."""
        n = int(self._lib.fsdp_resident_frames(self._h))
        nk, t, c = np.zeros(n, np.int32), np.zeros((n, 34)), np.zeros((n, 68))
        self._check(self._lib.fsdp_debug_refit(self._h, _ip(nk), _dp(t), _dp(c)), "fsdp_debug_refit")
        return nk, t, c

    def selftest_math(self, x, a, b) -> np.ndarray:
        """(5, n): sqrt_1_2(x), sqrt(x), fast a/b, IEEE a/b, operands in the safe band — all computed on the device."""
        x, a, b = (np.ascontiguousarray(v, np.float64) for v in (x, a, b))
        out = np.zeros((5, len(x)))
        self._check(self._lib.fsdp_selftest_math(self._h, len(x), _dp(x), _dp(a), _dp(b), _dp(out)), "fsdp_selftest_math")
        return out

    def selftest_givens(self, piv, ww) -> np.ndarray:
        """(7, n): cs, sn, dd of the kernels' fpgivs sequence | cs, sn, dd with the IEEE operations | operands inside the band."""
        piv, ww = (np.ascontiguousarray(v, np.float64) for v in (piv, ww))
        out = np.zeros((7, len(piv)))
        self._check(self._lib.fsdp_selftest_givens(self._h, len(piv), _dp(piv), _dp(ww), _dp(out)), "fsdp_selftest_givens")
        return out

    def selftest_absminmax(self, a, b) -> np.ndarray:
        """(2, n): max(|a|, b), min(|a|, b) as the device's Givens step computes them."""
        a, b = (np.ascontiguousarray(v, np.float64) for v in (a, b))
        out = np.zeros((2, len(a)))
        self._check(self._lib.fsdp_selftest_absminmax(self._h, len(a), _dp(a), _dp(b), _dp(out)), "fsdp_selftest_absminmax")
        return out

    def selftest_libm(self, y, x, cs) -> np.ndarray:
        """(3, n): the device library's atan2(y, x), det_atan2(y, x) (correctly rounded), the device library's acos(cs)."""
        y, x, cs = (np.ascontiguousarray(v, np.float64) for v in (y, x, cs))
        out = np.zeros((3, len(y)))
        self._check(self._lib.fsdp_selftest_libm(self._h, len(y), _dp(y), _dp(x), _dp(cs), _dp(out)), "fsdp_selftest_libm")
        return out

    def selftest_det3(self, xy6) -> np.ndarray:
        """det3_lu of (n,6) rows x0,y0,x1,y1,x2,y2 on the device."""
        xy6 = np.ascontiguousarray(xy6, np.float64).reshape(-1, 6)
        out = np.zeros(len(xy6))
        self._check(self._lib.fsdp_selftest_det3(self._h, len(xy6), _dp(xy6), _dp(out)), "fsdp_selftest_det3")
        return out

    def pcie_probe(self, nbytes: int, iters: int = 30) -> dict:
        """GB/s of page-locked hipMemcpyAsync over this GPU's host link: {'h2d', 'd2h', 'both_each'} (fsdp_pcie_probe)."""
        a, b, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        self._check(self._lib.fsdp_pcie_probe(self._h, ctypes.c_size_t(int(nbytes)), int(iters), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)), "fsdp_pcie_probe")
        return {"h2d": a.value, "d2h": b.value, "both_each": c.value}

    def default_path(self) -> np.ndarray:
        out = np.zeros((self.shapes.path_points, 4))
        self._check(self._lib.fsdp_default_path(self._h, _dp(out)), "fsdp_default_path")
        return out
