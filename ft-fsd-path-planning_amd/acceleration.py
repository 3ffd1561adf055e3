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
