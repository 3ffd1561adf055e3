"""Acceleration / EBS-test missions (reference full_pipeline.py:46-50,118-136,178-194 +
relocalization/acceleration/acceleration_relocalization.py:120-170).

The reference places the car in a known straight-lane map once: it fits a line through the cones it expects on the
left (0 < y < 2 m in the car frame) with a 100-draw random-subset search, takes the slope as the heading error and from
then on plans along a fixed path table.  That relocalization is a one-off, NumPy-sized computation per planner, so it
stays host code here; everything per frame — the path stage with the known path as ``global_path`` — runs on the GPU
through the ordinary batch entry points (``fsdp_set_global_path``, empty cone lists).

Randomness: the reference draws from NumPy's *global* legacy generator (``np.random.choice``), i.e. its result depends
on whatever the process did before.  Here the draw is explicit: ``seed`` feeds a private ``RandomState``; a reference
run preceded by ``np.random.seed(seed)`` makes the same draws (that is how tests/golden/global_path.npz was captured).
``seed=None`` uses fresh OS entropy, which is the closest to the reference's unseeded default.
"""
from __future__ import annotations

from pathlib import Path
from typing import Optional, Tuple

import numpy as np

_TABLE = Path(__file__).resolve().parent / "data" / "acceleration_path.npy"


def known_path() -> np.ndarray:
    """The relocalizer's known map path (n,2): data of the reference (BASE_ACCELERATION_PATH), shipped as a table."""
    return np.load(_TABLE)


def _rotate(points: np.ndarray, theta: float) -> np.ndarray:
    # utils/math_utils.py:103-117: points @ [[cos, sin], [-sin, cos]] through np.dot
    c, s = np.cos(theta), np.sin(theta)
    return np.dot(points, np.array(((c, -s), (s, c))).T)


class AccelerationRelocalizer:
    """State and arithmetic of the reference's Relocalizer base class (relocalization_base_class.py:22-95) and its
    acceleration subclass, with an explicit random stream."""

    SUBSET, DRAWS = 3, 100  # best_fit(points, subset_size=3, iterations=100)

    def __init__(self, seed: Optional[int] = 0):
        self._rng = np.random.RandomState(seed)
        self.original_position: Optional[np.ndarray] = None
        self.original_direction: Optional[np.ndarray] = None
        self.angle_to_fix: Optional[float] = None

    @property
    def is_relocalized(self) -> bool:
        return self.angle_to_fix is not None

    def attempt(self, cones_by_type, position: np.ndarray, direction: np.ndarray) -> None:
        if self.is_relocalized:
            return
        if self.original_position is None:  # latched on the first attempt, successful or not
            self.original_position = np.array(position, dtype=np.float64)
            self.original_direction = np.array(direction, dtype=np.float64)
        all_cones = np.vstack([np.asarray(c, dtype=np.float64).reshape(-1, 2) for c in cones_by_type])
        if len(all_cones) < 3:
            return
        yaw = np.arctan2(direction[1], direction[0])
        local = _rotate(all_cones - position, -yaw)
        band = local[(local[:, 1] > 0) & (local[:, 1] < 2)]
        band = band[band[:, 0].argsort()]
        if len(band) < 4:
            return
        best, smallest = None, np.inf
        for _ in range(self.DRAWS):
            pts = band[self._rng.choice(band.shape[0], self.SUBSET, replace=False)]
            coeff = np.polyfit(pts[:, 0], pts[:, 1], 1)
            err = np.sum((pts[:, 1] - np.polyval(coeff, pts[:, 0])) ** 2)
            if err < smallest:
                smallest, best = err, coeff
        self.angle_to_fix = np.arctan(best[0]) + yaw

    def to_known_frame(self, position: np.ndarray, yaw) -> Tuple[np.ndarray, float]:
        return _rotate(position - self.original_position, -self.angle_to_fix), yaw - self.angle_to_fix

    def to_original_frame(self, position: np.ndarray, yaw) -> Tuple[np.ndarray, float]:
        return _rotate(position, self.angle_to_fix) + self.original_position, yaw + self.angle_to_fix


# ---- many planners in lock-step -----------------------------------------------------------------------------------------
# NumPy hands np.dot to OpenBLAS, whose kernels fuse their multiply-adds (DESIGN.md "arithmetic contract"), and the path stage's
# sample count hangs on the last bit of the pose it is given — so a vectorised relocalizer must reproduce those fused operations
# exactly, for every planner at once, without a per-planner np.dot.  fma() below is the correctly rounded a * b + c of IEEE 754 built
# from float64 operations only (Boldo & Melquiond 2008, "Emulation of a FMA and correctly rounded sums: proved algorithms using
# rounding to odd": exact product by Dekker's splitting, TwoSum, the inner sum rounded to odd); valid while no intermediate
# overflows or becomes subnormal (coordinates of a race track: metres).  tests/test_cabi_cpu.py holds it against libm's fma.
def _two_sum(a, b):
    s = a + b
    bb = s - a
    return s, (a - (s - bb)) + (b - bb)


def _split(a):
    c = 134217729.0 * a  # 2^27 + 1
    hi = c - (c - a)
    return hi, a - hi


def fma(a, b, c):
    """Correctly rounded a * b + c, elementwise (float64 arrays; finite values of ordinary magnitude)."""
    a, b, c = np.broadcast_arrays(np.asarray(a, np.float64), np.asarray(b, np.float64), np.asarray(c, np.float64))
    with np.errstate(all="ignore"):
        uh = a * b
        ah, al = _split(a)
        bh, bl = _split(b)
        ul = ((ah * bh - uh) + ah * bl + al * bh) + al * bl  # uh + ul = a * b exactly
        th, tl = _two_sum(c, uh)
        v, err = _two_sum(tl, ul)  # v = RN(tl + ul), err its error: round v to odd
        odd = (v.view(np.int64) & 1) != 0
        fix = (err != 0) & ~odd
        up = np.nextafter(v, np.inf)
        down = np.nextafter(v, -np.inf)
        v = np.where(fix, np.where(err > 0, up, down), v)
        return th + v


def rotate_single_points(x, y, theta):
    """utils/math_utils.py:103-117 `rotate(point, theta)` = np.dot(point, R) for ONE point per call — OpenBLAS gemv order,
    fma(x, R0j, y * R1j) (oracle/np_compat.h blas_dot2_single_row) — for arrays of points with their own angles."""
    c, s = np.cos(theta), np.sin(theta)
    return fma(x, c, y * (-s)), fma(x, s, y * c)


def rotate_point_rows(x, y, theta):
    """The same call for an (n >= 2, 2) array of points: gemm order, fma(y, R1j, x * R0j) (blas_dot2)."""
    c, s = np.cos(theta), np.sin(theta)
    return fma(y, -s, x * c), fma(y, c, x * s)


class AccelerationBatch:
    """n acceleration / ebs_test planners advanced in lock-step (the counterpart of SkidpadBatch for these missions): one
    `step(cone_offsets, cones_xyt, poses)` = one `calculate_path_in_global_frame` call of every planner, with the results of
    n `PathPlanner(mission, relocalization_seed=seeds[i])` objects called one after the other — bit for bit (tests/test_global_path.py).

    What runs where: the relocalization (acceleration_relocalization.py:120-170) is a one-off line fit per planner and stays the
    per-planner host code above (`AccelerationRelocalizer.attempt`, only for planners that are not relocalized yet); everything a
    step does for every planner every time — the pose into the known frame, the path stage along the known path, the path back
    into the caller's frame (full_pipeline.py:118-136,178-194) — is one vectorised transform, ONE batched launch sequence on the
    GPU (fsdp_plan_batch_sequential: a planner's previous path travels with it) and one vectorised transform back."""

    def __init__(self, n_planners: int, mission=None, seeds=None, device: int | None = None, params: dict | None = None):
        from . import _capi
        from .planner import MissionTypes

        self.n = int(n_planners)
        self.mission = MissionTypes.acceleration if mission is None else MissionTypes(mission)
        if self.mission not in (MissionTypes.acceleration, MissionTypes.ebs_test):
            raise ValueError("AccelerationBatch plans the acceleration and ebs_test missions")
        seeds = list(range(self.n)) if seeds is None else list(seeds)
        if len(seeds) != self.n:
            raise ValueError("one relocalization seed per planner")
        self.relocalizers = [AccelerationRelocalizer(s) for s in seeds]
        # two contexts: planners that are relocalized plan along the known path, the others from (no) cones and their previous path
        self._known = _capi.Context(device=device, mission=int(self.mission), params=params)
        self._known.set_global_path(known_path())
        self._blind = None
        self._device, self._params = device, params
        self.horizon = self._known.horizon
        self._prev = np.repeat(self._known.default_path()[None, : self.horizon], self.n, axis=0)  # previous_paths[-1] of a fresh planner
        self._orig = np.zeros((self.n, 2))
        self._angle = np.zeros(self.n)
        self._reloc = np.zeros(self.n, dtype=bool)

    @property
    def relocalized(self) -> np.ndarray:
        return self._reloc.copy()

    @property
    def angles(self) -> np.ndarray:
        """angle_to_fix per planner (NaN before relocalization)."""
        return np.where(self._reloc, self._angle, np.nan)

    def step(self, cone_offsets, cones_xyt, poses):
        """Returns (paths (n, horizon, 4) in the callers' frames, status (n,)): status != 0 rows hold NaN — the planner object would
        have raised (planner.raise_for_status names the exception) and, like it, keeps its previous path."""
        from . import _capi

        off, cones, poses, n = _capi.Context._prep(cone_offsets, cones_xyt, poses)
        if n != self.n:
            raise ValueError(f"{n} frames for {self.n} planners")
        position, direction = poses[:, :2], poses[:, 2:]
        # the relocalization attempts of the planners still waiting for one (one-off per planner: host code as in the planner object)
        for i in np.nonzero(~self._reloc)[0]:
            r = self.relocalizers[i]
            r.attempt([cones[off[i] : off[i + 1], :2]], position[i], direction[i])
            if r.is_relocalized:
                self._reloc[i], self._angle[i], self._orig[i] = True, r.angle_to_fix, r.original_position
        k = self._reloc
        pose_k = poses.copy()
        if k.any():
            yaw = np.arctan2(direction[k, 1], direction[k, 0])
            d = position[k] - self._orig[k]
            pose_k[k, 0], pose_k[k, 1] = rotate_single_points(d[:, 0], d[:, 1], -self._angle[k])
            yaw_k = yaw - self._angle[k]
            pose_k[k, 2], pose_k[k, 3] = np.cos(yaw_k), np.sin(yaw_k)
        res_path = np.full((n, self.horizon, 4), np.nan)
        status = np.zeros(n, dtype=np.int32)
        empty_off = np.zeros(1, np.int32)
        for sel, ctx in ((k, self._known), (~k, None)):
            if not sel.any():
                continue
            if ctx is None:
                if self._blind is None:
                    self._blind = _capi.Context(device=self._device, mission=int(self.mission), params=self._params)
                ctx = self._blind
            m = int(sel.sum())
            r = ctx.plan_batch_sequential(np.zeros(m + 1, np.int32), np.zeros((0, 3)), pose_k[sel], self._prev[sel])
            status[sel] = r["status"]
            res_path[sel] = r["path"][:, : self.horizon]
        ok = status == 0
        self._prev[ok] = res_path[ok]  # the path stage keeps its history in the frame it computed in
        res_path[~ok] = np.nan
        back = ok & k
        if back.any():
            ang = self._angle[back][:, None]
            x, y = rotate_point_rows(res_path[back, :, 1], res_path[back, :, 2], ang)
            res_path[back, :, 1] = x + self._orig[back, 0][:, None]
            res_path[back, :, 2] = y + self._orig[back, 1][:, None]
        return res_path, status

    def close(self):
        self._known.close()
        if self._blind is not None:
            self._blind.close()
