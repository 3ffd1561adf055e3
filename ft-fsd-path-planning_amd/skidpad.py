"""Skidpad mission host side (BASELINE config 5): stateful planner instances on one GPU.

Mirrors what ``PathPlanner(MissionTypes.skidpad)`` does in the reference (full_pipeline.py:118-194):
relocalization until it succeeds, then the known-map window path, path returned in the caller's frame,
``relocalization_info`` (relocalization_information.py:12-35).  Constant inputs handed to the library
(include/fsdp.h fsdp_skidpad_set_tables):
  * the known skidpad path (data/skidpad_path.npy = the (5786,2) table of the reference's
    relocalization/skidpad/skidpad_path_data.py — track DATA, not code),
  * the fixed-seed noise table numpy.random.RandomState(42).randn(1140,3,2) (skidpad_relocalizer.py:38,52),
  * the two reference circle centres (skidpad_relocalizer.py:172-183), computed here with NumPy by an own
    restatement of the hyper circle fit (utils/math_utils.py:579-646) so the bits are NumPy's.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from pathlib import Path

import numpy as np

from . import _capi

DATA = Path(__file__).resolve().parent / "data" / "skidpad_path.npy"

INFO_DTYPE = np.dtype([("relocalized", "<i4"), ("index_along_path", "<i4"), ("translation", "<f8", (2,)), ("rotation", "<f8")], align=True)


def hyper_circle_fit(coords: np.ndarray, max_iter: int = 99) -> np.ndarray:
    """Hyper circle fit (moments + Newton on the characteristic polynomial); returns [cx, cy, r]."""
    X, Y = coords[:, 0], coords[:, 1]
    n = X.shape[0]
    Xi, Yi = X - X.mean(), Y - Y.mean()
    Zi = Xi * Xi + Yi * Yi
    Mxy, Mxx, Myy = (Xi * Yi).sum() / n, (Xi * Xi).sum() / n, (Yi * Yi).sum() / n
    Mxz, Myz, Mzz = (Xi * Zi).sum() / n, (Yi * Zi).sum() / n, (Zi * Zi).sum() / n
    Mz = Mxx + Myy
    Cov_xy = Mxx * Myy - Mxy * Mxy
    Var_z = Mzz - Mz * Mz
    A2 = 4 * Cov_xy - 3 * Mz * Mz - Mzz
    A1 = Var_z * Mz + 4.0 * Cov_xy * Mz - Mxz * Mxz - Myz * Myz
    A0 = Mxz * (Mxz * Myy - Myz * Mxy) + Myz * (Myz * Mxx - Mxz * Mxy) - Var_z * Cov_xy
    A22 = A2 + A2
    y, x = A0, 0.0
    for _ in range(max_iter):
        Dy = A1 + x * (A22 + 16.0 * x * x)
        x_new = x - y / Dy
        if x_new == x or not np.isfinite(x_new):
            break
        y_new = A0 + x_new * (A1 + x_new * (A2 + 4.0 * x_new * x_new))
        if abs(y_new) >= abs(y):
            break
        x, y = x_new, y_new
    det = x * x - x * Mz + Cov_xy
    Xc = (Mxz * (Myy - x) - Myz * Mxy) / det / 2.0
    Yc = (Myz * (Mxx - x) - Mxz * Mxy) / det / 2.0
    return np.array([Xc + X.mean(), Yc + Y.mean(), np.sqrt(abs(Xc**2 + Yc**2 + Mz))])


def load_tables(table: np.ndarray | None = None):
    table = np.load(DATA) if table is None else np.ascontiguousarray(table, dtype=np.float64)
    noise = np.random.RandomState(42).randn(1140, 3, 2)
    neg, pos = table[table[:, 1] < -2], table[table[:, 1] > 2]
    ref = np.array([hyper_circle_fit(neg)[:2], hyper_circle_fit(pos)[:2]])  # [right, left]
    half = table[::2]
    mean_distance = float(np.mean(np.linalg.norm(np.diff(half[:10], axis=-2), axis=-1)))
    return table, noise, ref, mean_distance


@dataclass
class RelocalizationInformation:
    """reference relocalization/relocalization_information.py:12-35"""

    translation: np.ndarray
    rotation: float


class SkidpadBatch:
    """n independent skidpad planners advanced in lock-step (one frame of every instance per step)."""

    def __init__(self, n_instances: int = 1, device: int | None = None, table: np.ndarray | None = None):
        self._ctx = _capi.Context(device=device, mission=2)
        self.n = int(n_instances)
        table, noise, ref, mean_distance = load_tables(table)
        self.tables = (table, noise, ref, mean_distance)
        L, h = self._ctx._lib, self._ctx._h
        dp = _capi._dp
        self._ctx._check(
            L.fsdp_skidpad_set_tables(h, dp(table), ctypes.c_int(len(table)), dp(noise), ctypes.c_int(noise.size),
                                      dp(np.ascontiguousarray(ref).ravel()), ctypes.c_double(mean_distance)),
            "fsdp_skidpad_set_tables")
        self.reset()

    def reset(self):
        self._ctx._check(self._ctx._lib.fsdp_skidpad_reset(self._ctx._h, ctypes.c_int(self.n)), "fsdp_skidpad_reset")

    def step(self, cone_offsets, cones_xyt, poses):
        off, cones, poses, n = self._ctx._prep(cone_offsets, cones_xyt, poses)
        assert n == self.n
        res = np.zeros(n, dtype=_capi.RESULT_DTYPE)
        info = np.zeros(n, dtype=INFO_DTYPE)
        self._ctx._check(
            self._ctx._lib.fsdp_skidpad_step(self._ctx._h, ctypes.c_int(n), _capi._ip(off), _capi._dp(cones), _capi._dp(poses),
                                             ctypes.c_void_p(res.ctypes.data), ctypes.c_void_p(info.ctypes.data)),
            "fsdp_skidpad_step")
        return res, info

    def time_path(self, iters: int) -> float:
        t = ctypes.c_float()
        self._ctx._check(self._ctx._lib.fsdp_skidpad_time_path(self._ctx._h, ctypes.c_int(iters), ctypes.byref(t)), "fsdp_skidpad_time_path")
        return float(t.value)
