"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Imported by tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke(); never by the
product package.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"
# tests/oracle_lib_wide.py runs this file a second time as its own module with WIDE_SHAPES preset: the same restatement built
# with the record shapes of the library's wide build (oracle/Makefile liboracle_wide.so)
WIDE_SHAPES = bool(globals().get("WIDE_SHAPES", False))
LIB_PATH = ORACLE_DIR / ("liboracle_wide.so" if WIDE_SHAPES else "liboracle.so")

MAX_LEN, MAX_MATCH, PATH_POINTS = (16, 32, 64) if WIDE_SHAPES else (12, 24, 40)


class FrameResult(ctypes.Structure):
    _fields_ = [
        ("status", ctypes.c_int32),
        ("n_left", ctypes.c_int32),
        ("n_right", ctypes.c_int32),
        ("left_idx", ctypes.c_int32 * MAX_LEN),
        ("right_idx", ctypes.c_int32 * MAX_LEN),
        ("n_left_v", ctypes.c_int32),
        ("n_right_v", ctypes.c_int32),
        ("left_v", (ctypes.c_double * 2) * MAX_MATCH),
        ("right_v", (ctypes.c_double * 2) * MAX_MATCH),
        ("l2r", ctypes.c_int32 * MAX_MATCH),
        ("r2l", ctypes.c_int32 * MAX_MATCH),
        ("path", (ctypes.c_double * 4) * PATH_POINTS),
        ("n_configs_left", ctypes.c_int32),
        ("n_configs_right", ctypes.c_int32),
        ("first_k_left", ctypes.c_int32 * 2),
        ("first_k_right", ctypes.c_int32 * 2),
        ("best_cost_left", ctypes.c_double),
        ("best_cost_right", ctypes.c_double),
        ("path_fallback", ctypes.c_int32),
    ]


RESULT_DTYPE = np.dtype(
    [
        ("status", "<i4"),
        ("n_left", "<i4"),
        ("n_right", "<i4"),
        ("left_idx", "<i4", (MAX_LEN,)),
        ("right_idx", "<i4", (MAX_LEN,)),
        ("n_left_v", "<i4"),
        ("n_right_v", "<i4"),
        ("left_v", "<f8", (MAX_MATCH, 2)),
        ("right_v", "<f8", (MAX_MATCH, 2)),
        ("l2r", "<i4", (MAX_MATCH,)),
        ("r2l", "<i4", (MAX_MATCH,)),
        ("path", "<f8", (PATH_POINTS, 4)),
        ("n_configs_left", "<i4"),
        ("n_configs_right", "<i4"),
        ("first_k_left", "<i4", (2,)),
        ("first_k_right", "<i4", (2,)),
        ("best_cost_left", "<f8"),
        ("best_cost_right", "<f8"),
        ("path_fallback", "<i4"),
    ],
    align=True,
)

_lib = None


def build() -> None:
    subprocess.run(["make", "-s", "-C", str(ORACLE_DIR)], check=True)


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            build()
        _lib = ctypes.CDLL(str(LIB_PATH))
        assert _lib.fsdo_result_size() == RESULT_DTYPE.itemsize == ctypes.sizeof(FrameResult), (
            _lib.fsdo_result_size(),
            RESULT_DTYPE.itemsize,
            ctypes.sizeof(FrameResult),
        )
    return _lib


def set_math_mode(mode: int) -> None:
    """0 = libm (reference pinning, default); 1 = deterministic sin/cos/atan2 (exact parity with the HIP kernels)."""
    lib().fsdo_set_math_mode(ctypes.c_int(mode))


class math_mode:
    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = lib().fsdo_get_math_mode()
        set_math_mode(self.mode)

    def __exit__(self, *a):
        set_math_mode(self.prev)


def _p(a, t=ctypes.c_double):
    return a.ctypes.data_as(ctypes.POINTER(t))


def plan_frame(xyt: np.ndarray, pose: np.ndarray) -> np.ndarray:
    xyt = np.ascontiguousarray(xyt, dtype=np.float64).reshape(-1, 3)
    pose = np.ascontiguousarray(pose, dtype=np.float64)
    out = np.zeros(1, dtype=RESULT_DTYPE)
    lib().fsdo_plan_frame(_p(xyt), ctypes.c_int(len(xyt)), _p(pose), ctypes.c_void_p(out.ctypes.data))
    return out[0]


def plan_frame_prev(xyt: np.ndarray, pose: np.ndarray, prev: np.ndarray | None) -> np.ndarray:
    """Sequential-replay form: prev = the previous (40,4) output of the same planner, or None for a fresh one."""
    xyt = np.ascontiguousarray(xyt, dtype=np.float64).reshape(-1, 3)
    pose = np.ascontiguousarray(pose, dtype=np.float64)
    out = np.zeros(1, dtype=RESULT_DTYPE)
    pp = None if prev is None else _p(np.ascontiguousarray(prev, dtype=np.float64))
    lib().fsdo_plan_frame_prev(_p(xyt), ctypes.c_int(len(xyt)), _p(pose), pp, ctypes.c_void_p(out.ctypes.data))
    return out[0]


def plan_frame_global(xyt: np.ndarray, pose: np.ndarray, prev: np.ndarray | None, global_path: np.ndarray | None) -> np.ndarray:
    """plan_frame_prev with PathPlanner.global_path set (full_pipeline.py:81-82): global_path (n,2) or None."""
    xyt = np.ascontiguousarray(xyt, dtype=np.float64).reshape(-1, 3)
    pose = np.ascontiguousarray(pose, dtype=np.float64)
    out = np.zeros(1, dtype=RESULT_DTYPE)
    pp = None if prev is None else _p(np.ascontiguousarray(prev, dtype=np.float64))
    gp = None if global_path is None else np.ascontiguousarray(global_path, dtype=np.float64).reshape(-1, 2)
    lib().fsdo_plan_frame_global(_p(xyt), ctypes.c_int(len(xyt)), _p(pose), pp, None if gp is None else _p(gp),
                                 ctypes.c_int(0 if gp is None else len(gp)), ctypes.c_void_p(out.ctypes.data))
    return out[0]


def plan_batch(offsets, xyt, poses, n_threads: int = 1) -> np.ndarray:
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    xyt = np.ascontiguousarray(xyt, dtype=np.float64)
    poses = np.ascontiguousarray(poses, dtype=np.float64)
    n = len(offsets) - 1
    out = np.zeros(n, dtype=RESULT_DTYPE)
    lib().fsdo_plan_batch(
        ctypes.c_int(n), _p(offsets, ctypes.c_int32), _p(xyt), _p(poses), ctypes.c_void_p(out.ctypes.data), ctypes.c_int(n_threads)
    )
    return out


def default_path() -> np.ndarray:
    out = np.zeros((PATH_POINTS, 4))
    lib().fsdo_default_path(_p(out))
    return out


def side_configs(xyt, pose, cone_type: int, max_configs: int = 64):
    xyt = np.ascontiguousarray(xyt, dtype=np.float64).reshape(-1, 3)
    pose = np.ascontiguousarray(pose, dtype=np.float64)
    cfg = np.full((max_configs, MAX_LEN), -1, dtype=np.int32)
    costs = np.zeros(max_configs)
    fk = np.full(2, -1, dtype=np.int32)
    c = lib().fsdo_side_configs(
        _p(xyt), ctypes.c_int(len(xyt)), _p(pose), ctypes.c_int(cone_type), _p(cfg, ctypes.c_int32), _p(costs),
        ctypes.c_int(max_configs), _p(fk, ctypes.c_int32),
    )
    return c, cfg[: max(c, 0)], costs[: max(c, 0)], fk


def splprep(trace: np.ndarray, s: float, k: int):
    trace = np.ascontiguousarray(trace, dtype=np.float64)
    u = np.concatenate(([0.0], np.cumsum(np.linalg.norm(np.diff(trace, axis=0), axis=1))))
    m = len(trace)
    x = np.ascontiguousarray(trace[:, 0])
    y = np.ascontiguousarray(trace[:, 1])
    t = np.zeros(m + 2 * k + 2)
    cx = np.zeros_like(t)
    cy = np.zeros_like(t)
    n = ctypes.c_int()
    ier = ctypes.c_int()
    fp = ctypes.c_double()
    rc = lib().fsdo_splprep(
        _p(u), _p(x), _p(y), ctypes.c_int(m), ctypes.c_int(k), ctypes.c_double(s), _p(t), _p(cx), _p(cy),
        ctypes.byref(n), ctypes.byref(ier), ctypes.byref(fp),
    )
    return rc, u, t[: n.value], cx[: n.value], cy[: n.value], ier.value, fp.value


def splev(t, cx, cy, k, u_eval):
    t = np.ascontiguousarray(t)
    cx = np.ascontiguousarray(cx)
    cy = np.ascontiguousarray(cy)
    u_eval = np.ascontiguousarray(u_eval, dtype=np.float64)
    ox = np.zeros(len(u_eval))
    oy = np.zeros(len(u_eval))
    lib().fsdo_splev(_p(t), _p(cx), _p(cy), ctypes.c_int(len(t)), ctypes.c_int(k), _p(u_eval), ctypes.c_long(len(u_eval)), _p(ox), _p(oy))
    return np.column_stack([ox, oy])


class SkidpadPlanner:
    """Stateful skidpad-mission oracle (oracle/skidpad.cpp).  table: BASE_SKIDPAD_PATH (n,2); noise: RandomState(42).randn(1140,3,2)."""

    def __init__(self, table, noise):
        table = np.ascontiguousarray(table, np.float64)
        noise = np.ascontiguousarray(noise, np.float64).ravel()
        L = lib()
        L.fsdo_skidpad_create.restype = ctypes.c_void_p
        self._h = ctypes.c_void_p(L.fsdo_skidpad_create(_p(table), ctypes.c_int(len(table)), _p(noise), ctypes.c_int(len(noise))))

    def step(self, xyt, pose):
        xyt = np.ascontiguousarray(xyt, np.float64).reshape(-1, 3)
        pose = np.ascontiguousarray(pose, np.float64)
        out = np.zeros(1, RESULT_DTYPE)
        info = np.zeros(5)
        lib().fsdo_skidpad_step(self._h, _p(xyt), ctypes.c_int(len(xyt)), _p(pose), ctypes.c_void_p(out.ctypes.data), _p(info))
        return out[0], info

    def reference_centers(self):
        out = np.zeros(4)
        lib().fsdo_skidpad_reference_centers(self._h, _p(out))
        return out.reshape(2, 2)

    def __del__(self):
        try:
            lib().fsdo_skidpad_destroy(self._h)
        except Exception:
            pass


PARAM_ORDER = ["max_n_neighbors", "max_length", "max_dist", "max_dist_to_first", "threshold_directional_angle",
               "threshold_absolute_angle", "min_track_width", "max_search_range", "max_search_angle", "smoothing", "predict_every",
               "maximal_distance_for_valid_path", "mpc_path_length", "max_deg", "mpc_prediction_horizon", "matches_should_be_monotonic",
               "use_unknown_cones"]
PARAM_DEFAULTS = [5, 12, 6.5, 6.0, float(np.deg2rad(40)), float(np.deg2rad(65)), 3.0, 5.0, float(np.deg2rad(50)), 0.2, 0.1, 5.0, 20.0,
                  3, 40, 0, 1]


def param_vector(overrides=None):
    v = dict(zip(PARAM_ORDER, PARAM_DEFAULTS))
    for k, val in (overrides or {}).items():
        assert k in v, k
        v[k] = float(val)
    return np.array([v[k] for k in PARAM_ORDER], dtype=np.float64)


class params:
    """with oracle_lib.params(dict(max_dist=5.5)): ... — the oracle with non-default configuration constants."""

    def __init__(self, overrides):
        self.v = param_vector(overrides)

    def __enter__(self):
        lib().fsdo_set_params(_p(self.v))

    def __exit__(self, *a):
        lib().fsdo_set_params(None)


FIT_KNOTS, FIT_STRIDE, MAX_FITS = 48, 2 + 3 * 48, 6


def plan_frame_capture(xyt, pose):
    """(result row, [(k, n, t, cx, cy), ...]): the frame and every smoothing spline it fitted, in call order."""
    xyt = np.ascontiguousarray(xyt, dtype=np.float64).reshape(-1, 3)
    pose = np.ascontiguousarray(pose, dtype=np.float64)
    out = np.zeros(1, dtype=RESULT_DTYPE)
    buf = np.zeros((MAX_FITS, FIT_STRIDE))
    lib().fsdo_plan_frame_capture.restype = ctypes.c_int
    nf = lib().fsdo_plan_frame_capture(_p(xyt), ctypes.c_int(len(xyt)), _p(pose), ctypes.c_void_p(out.ctypes.data), _p(buf), ctypes.c_int(MAX_FITS))
    fits = []
    for i in range(min(nf, MAX_FITS)):
        k, n = int(buf[i, 0]), int(buf[i, 1])
        nn = min(n, FIT_KNOTS)
        fits.append((k, n, buf[i, 2 : 2 + nn].copy(), buf[i, 2 + FIT_KNOTS : 2 + FIT_KNOTS + nn].copy(),
                     buf[i, 2 + 2 * FIT_KNOTS : 2 + 2 * FIT_KNOTS + nn].copy()))
    return out[0], nf, fits
