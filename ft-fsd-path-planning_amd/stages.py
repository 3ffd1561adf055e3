"""The three stage classes the reference exposes individually (README.md:78-79), with the same 3-call protocol
``set_new_input(Input dataclass) -> run_*()`` and the same return tuples, backed by the stage-level C-ABI entry
points (fsdp_sort_batch / fsdp_match_batch / fsdp_path_batch):

  ConeSorting   sorting_cones/core_cone_sorting.py:21-136
  ConeMatching  cone_matching/core_cone_matching.py:26-124
  CalculatePath calculate_path/core_calculate_path.py:36-60,514-575   (fresh-planner semantics)

The constructors take the reference's kwargs (defaults = the factories of config.py); they travel to the kernels as the
context's parameter block (include/fsdp.h fsdp_params).  Values outside what the kernels can take (max_n_neighbors > 5,
max_length > 12, max_deg != 3, mpc_prediction_horizon != 40, use_unknown_cones = False, monotonic matching) make the
library refuse the context: nothing is silently substituted.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

from . import _capi
from .planner import ConeTypes, flatten_cones_by_type_array, raise_for_status

_shared_ctx = {}


def _ctx(device=None, params=None, wide: bool = False):
    """One context per (device, parameter set, library build); the eight most recently used are kept (a parameter sweep through
    the stage classes would otherwise pile up contexts, each with its device buffers).  wide: the stage's INPUT needs the wide
    build's shapes (more than 12 sorted cones per side, as a ConeSorting with max_length > 12 returns them); parameters beyond
    the standard shapes select it by themselves (_capi.Context)."""
    key = (device, tuple(sorted((params or {}).items())), bool(wide))
    ctx = _shared_ctx.pop(key, None)
    if ctx is None:
        ctx = _capi.Context(device=device, mission=4, params=params, shapes=_capi.WIDE if wide else None)
    _shared_ctx[key] = ctx  # (most recent last)
    while len(_shared_ctx) > 8:
        _shared_ctx.pop(next(iter(_shared_ctx))).close()
    return ctx


def _overrides(defaults, kwargs):
    out = {}
    for k, v in kwargs.items():
        if k == "experimental_performance_improvements":
            if v:
                raise NotImplementedError("the experimental sorting cache is out of scope (SURVEY.md section 2 row 15)")
            continue
        if k not in defaults:
            raise TypeError(f"unexpected keyword argument {k!r}")
        out[k] = float(v) if isinstance(defaults[k], float) or isinstance(v, float) else v
    return out


_check = raise_for_status


@dataclass
class ConeSortingInput:
    slam_cones: List[np.ndarray] = field(default_factory=lambda: [np.zeros((0, 2)) for _ in ConeTypes])
    slam_position: np.ndarray = field(default_factory=lambda: np.zeros(2))
    slam_direction: np.ndarray = field(default_factory=lambda: np.zeros(2))


class ConeSorting:
    """Default parameters of config.py:33-41 (max_n_neighbors 5, max_dist 6.5, max_dist_to_first 6.0, max_length 12,
    thresholds 40/65 deg, use_unknown_cones True)."""

    DEFAULTS = dict(max_n_neighbors=5, max_dist=6.5, max_dist_to_first=6.0, max_length=12,
                    threshold_directional_angle=np.deg2rad(40), threshold_absolute_angle=np.deg2rad(65), use_unknown_cones=True)

    def __init__(self, device=None, **kwargs):
        self._params = _overrides(self.DEFAULTS, kwargs)
        self.input = ConeSortingInput()
        self._device = device

    def set_new_input(self, slam_input: ConeSortingInput) -> None:
        self.input = slam_input

    def run_cone_sorting(self) -> Tuple[np.ndarray, np.ndarray]:
        xyt = flatten_cones_by_type_array(self.input.slam_cones)
        pose = np.concatenate([np.asarray(self.input.slam_position, float).reshape(2), np.asarray(self.input.slam_direction, float).reshape(2)])
        r = _ctx(self._device, self._params).sort_batch(np.array([0, len(xyt)], np.int32), xyt, pose[None])[0]
        _check(r["status"])
        self.last_result = r
        return xyt[r["left_idx"][: r["n_left"]], :2], xyt[r["right_idx"][: r["n_right"]], :2]


@dataclass
class ConeMatchingInput:
    sorted_cones: List[np.ndarray] = field(default_factory=lambda: [np.zeros((0, 2)) for _ in ConeTypes])
    slam_position: np.ndarray = field(default_factory=lambda: np.zeros(2))
    slam_direction: np.ndarray = field(default_factory=lambda: np.zeros(2))


class ConeMatching:
    """Default parameters of config.py:124-129,162 (min_track_width 3, max_search_range 5, max_search_angle 50 deg,
    matches_should_be_monotonic False — the pipeline's choice, full_pipeline.py:65)."""

    DEFAULTS = dict(min_track_width=3, max_search_range=5, max_search_angle=np.deg2rad(50), matches_should_be_monotonic=False)

    def __init__(self, device=None, **kwargs):
        self._params = _overrides(self.DEFAULTS, kwargs)
        self.input = ConeMatchingInput()
        self._device = device

    def set_new_input(self, cone_matching_input: ConeMatchingInput) -> None:
        self.input = cone_matching_input

    def run_cone_matching(self):
        left = np.asarray(self.input.sorted_cones[int(ConeTypes.LEFT)], float).reshape(-1, 2)
        right = np.asarray(self.input.sorted_cones[int(ConeTypes.RIGHT)], float).reshape(-1, 2)
        longest = max(len(left), len(right))
        if longest > _capi.WIDE.max_len:
            raise _capi.FsdpError(f"at most {_capi.WIDE.max_len} sorted cones per side (config.py:36 max_length; include/fsdp.h shapes)")
        ctx = _ctx(self._device, self._params, wide=longest > _capi.STANDARD.max_len)
        cap = ctx.shapes.max_len
        sl, sr = np.zeros((1, cap, 2)), np.zeros((1, cap, 2))
        sl[0, : len(left)], sr[0, : len(right)] = left, right
        pose = np.concatenate([np.asarray(self.input.slam_position, float).reshape(2), np.asarray(self.input.slam_direction, float).reshape(2)])
        r = ctx.match_batch(sl, [len(left)], sr, [len(right)], pose[None])[0]
        _check(r["status"])
        self.last_result = r
        ml, mr = int(r["n_left_v"]), int(r["n_right_v"])
        return (np.array(r["left_v"][:ml]), np.array(r["right_v"][:mr]), np.array(r["l2r"][:ml], dtype=np.int64), np.array(r["r2l"][:mr], dtype=np.int64))


@dataclass
class PathCalculationInput:
    left_cones: np.ndarray = field(default_factory=lambda: np.zeros((0, 2)))
    right_cones: np.ndarray = field(default_factory=lambda: np.zeros((0, 2)))
    left_to_right_matches: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=int))
    right_to_left_matches: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=int))
    position_global: np.ndarray = field(default_factory=lambda: np.zeros(2))
    direction_global: np.ndarray = field(default_factory=lambda: np.array([1.0, 0.0]))
    global_path: np.ndarray | None = None


class CalculatePath:
    """Defaults of config.py:48,55-59 (smoothing 0.2, predict_every 0.1, max_deg 3, max valid distance 5 m, MPC length
    20 m, horizon 40).  Stateful like the reference object: the path a call returns is ``previous_paths[-1]`` of the next
    call (core_calculate_path.py:572-573); ``stateful=False`` gives every call a fresh object."""

    DEFAULTS = dict(smoothing=0.2, predict_every=0.1, max_deg=3, maximal_distance_for_valid_path=5, mpc_path_length=20, mpc_prediction_horizon=40)

    def __init__(self, device=None, stateful: bool = True, **kwargs):
        self._params = _overrides(self.DEFAULTS, kwargs)
        self.input = PathCalculationInput()
        self._device = device
        self.stateful = stateful
        self._prev = None

    def set_new_input(self, new_input: PathCalculationInput) -> None:
        self.input = new_input

    def run_path_calculation(self):
        i = self.input
        lv, rv = np.asarray(i.left_cones, float).reshape(-1, 2), np.asarray(i.right_cones, float).reshape(-1, 2)
        longest = max(len(lv), len(rv))
        if longest > _capi.WIDE.max_match:
            raise _capi.FsdpError(f"at most {_capi.WIDE.max_match} cones (with virtual ones) per side (include/fsdp.h shapes)")
        ctx = _ctx(self._device, self._params, wide=longest > _capi.STANDARD.max_match)
        res = np.zeros(1, dtype=ctx.result_dtype)
        res["n_left_v"], res["n_right_v"] = len(lv), len(rv)
        res["left_v"][0, : len(lv)], res["right_v"][0, : len(rv)] = lv, rv
        res["l2r"][0, : len(lv)] = np.asarray(i.left_to_right_matches, dtype=np.int32)
        res["r2l"][0, : len(rv)] = np.asarray(i.right_to_left_matches, dtype=np.int32)
        pose = np.concatenate([np.asarray(i.position_global, float).reshape(2), np.asarray(i.direction_global, float).reshape(2)])
        if i.global_path is not None:  # core_calculate_path.py:514-529: the path is drawn from the global path
            ctx.set_global_path(i.global_path)
        try:
            outs, centers = ctx.path_batch_centers(pose[None], res, None if self._prev is None else self._prev[None])
            out = outs[0]
        finally:
            if i.global_path is not None:
                ctx.set_global_path(None)
        _check(out["status"])
        self.last_result = out
        path = np.array(out["path"][: ctx.horizon])
        if self.stateful:
            self._prev = path.copy()
        # second value (core_calculate_path.py:575): the points the first fit was given, from the device
        return path, centers[0]
