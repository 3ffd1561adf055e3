"""Python host mirroring the reference's public interface for the hot path.

Same names, argument meaning and return shapes as
  fsd_path_planning.PathPlanner / ConeTypes / MissionTypes   (reference fsd_path_planning/__init__.py:8-13,
  full_pipeline/full_pipeline.py:53-217, utils/cone_types.py, utils/mission_types.py)
backed by the HIP library through _capi (ctypes, include/fsdp.h).  Semantics for batches:
independent frames, "fresh PathPlanner per frame" (SURVEY.md §8a quirk 12).
"""
from __future__ import annotations

from enum import IntEnum
from typing import Any, List, Sequence, Tuple, Union

import numpy as np

from . import _capi


class ConeTypes(IntEnum):
    """reference utils/cone_types.py:10-19"""

    UNKNOWN = 0
    RIGHT = YELLOW = 1
    LEFT = BLUE = 2
    START_FINISH_AREA = ORANGE_SMALL = 3
    START_FINISH_LINE = ORANGE_BIG = 4


class MissionTypes(IntEnum):
    """reference utils/mission_types.py:11-25"""

    none = 0
    acceleration = 1
    skidpad = 2
    autocross = 3
    trackdrive = 4
    ebs_test = 5
    inspection = 6
    manual_driving = 7


class ReferenceUndefinedError(RuntimeError):
    """The reference raises on this input (status 101-104, include/fsdp.h).  ``raise_for_status`` raises the subclass
    that is also an instance of the exception type the reference raises, so a caller's ``except IndexError`` /
    ``except np.linalg.LinAlgError`` written against the reference keeps firing."""

    def __init__(self, status: int):
        super().__init__(f"the reference implementation raises on this frame (status {status}, see include/fsdp.h)")
        self.status = status


class ReferenceIndexError(ReferenceUndefinedError, IndexError):
    """101: nearby_cone_search.py:88-94 (searchsorted index == len); 102: end_configurations.py:369 (DFS position past
    target_length); 104: core_calculate_path.py:544 (index into an empty array) — IndexError in the reference."""


class ReferenceLinAlgError(ReferenceUndefinedError, np.linalg.LinAlgError):
    """103: core_calculate_path.py:482-483 hands a (40,4) array on in a (n,2) context -> numpy.linalg.LinAlgError
    (a ValueError) raised from the second do_all_mpc_parameter_calculations call, outside the try (:564-570)."""


class CapacityError(_capi.FsdpError):
    """status 2xx: a fixed device capacity was exceeded (the reference's buffers grow without bound)."""

    def __init__(self, status: int):
        super().__init__(f"device capacity exceeded (status {status}, see include/fsdp.h)")
        self.status = status


_STATUS_EXC = {101: ReferenceIndexError, 102: ReferenceIndexError, 103: ReferenceLinAlgError, 104: ReferenceIndexError}


def raise_for_status(status) -> None:
    """Per-frame status -> the exception the reference-shaped single-frame calls raise (0 returns)."""
    status = int(status)
    if status == 0:
        return
    if 100 <= status < 200:
        raise _STATUS_EXC.get(status, ReferenceUndefinedError)(status)
    raise CapacityError(status)


def flatten_cones_by_type_array(cones_by_type) -> np.ndarray:
    """reference sorting_cones/trace_sorter/core_trace_sorter.py:37-54 (own restatement)."""
    if isinstance(cones_by_type, np.ndarray) and cones_by_type.ndim == 2 and cones_by_type.shape[1] == 3:
        return np.ascontiguousarray(cones_by_type, dtype=np.float64)
    parts = []
    for t in (0, 1, 2, 3, 4):
        c = np.asarray(cones_by_type[t], dtype=np.float64).reshape(-1, 2)
        parts.append(np.column_stack([c, np.full(len(c), float(t))]))
    return np.ascontiguousarray(np.concatenate(parts, axis=0)) if parts else np.zeros((0, 3))


def _direction_to_array(direction: Any) -> np.ndarray:
    """reference full_pipeline.py:71-79"""
    direction = np.squeeze(np.array(direction, dtype=np.float64))
    if direction.shape == (2,):
        return direction
    if direction.shape in [(1,), ()]:
        a = float(direction)
        return np.array([np.cos(a), np.sin(a)])
    raise ValueError("direction must be a float or a 2 element array")


def pack_frames(frames: Sequence[Tuple[Any, Any, Any]]):
    """[(cones_by_type | (N,3), position, direction), ...] -> (offsets, cones_xyt, poses)"""
    flat = [flatten_cones_by_type_array(f[0]) for f in frames]
    offsets = np.zeros(len(frames) + 1, dtype=np.int32)
    offsets[1:] = np.cumsum([len(c) for c in flat])
    cones = np.concatenate(flat, axis=0) if flat else np.zeros((0, 3))
    poses = np.array([np.concatenate([np.asarray(f[1], float).reshape(2), _direction_to_array(f[2])]) for f in frames]).reshape(-1, 4)
    return offsets, cones, poses


class PathPlanner:
    """Drop-in for fsd_path_planning.PathPlanner.

    ``calculate_path_in_global_frame`` keeps the reference signature and return values
    (full_pipeline.py:84-207); ``plan_batch`` is the batched form of the same call.  ``relocalization_seed`` feeds the
    random subset search of the acceleration / ebs_test relocalizer (acceleration.py; the reference draws from
    NumPy's global generator there).
    """

    def __init__(self, mission: MissionTypes, experimental_performance_improvements: bool = False, device: int | None = None,
                 stateful: bool = True, relocalization_seed: int | None = 0, params: dict | None = None,
                 devices: Sequence[int] | str | None = None):
        """devices: GPUs ``plan_batch`` shards its frames over from this one process (multi.MultiPlanner): a list of
        device indices, or "all" for every visible GPU; None (default) = the single GPU ``device``.  The single-frame
        call always runs on the first of them."""
        if experimental_performance_improvements:
            # reference README.md:24-27: off by default, changes results, meaningless for independent frames
            raise NotImplementedError("the experimental sorting cache is out of scope (SURVEY.md §2 row 15)")
        self.mission = MissionTypes(mission)
        self.global_path = None
        self._accel = None
        if self.mission in (MissionTypes.acceleration, MissionTypes.ebs_test):
            from .acceleration import AccelerationRelocalizer

            self._accel = AccelerationRelocalizer(relocalization_seed)
        # like the reference object, consecutive calls chain the previous path (core_calculate_path.py:572-573);
        # stateful=False gives every call a fresh planner (what plan_batch does for every frame)
        self.stateful = stateful
        self._prev = None
        self._skid = None
        self._skid_info = None
        self._multi = None
        if devices is not None and self.mission != MissionTypes.skidpad:
            from .multi import MultiPlanner

            self._multi = MultiPlanner(None if devices == "all" else devices, params=params, mission=int(self.mission))
            self._ctx = self._multi.ctx[0]
        elif self.mission == MissionTypes.skidpad:
            from .skidpad import SkidpadBatch

            self._skid = SkidpadBatch(1, device=device, params=params)  # stateful, like the reference's skidpad planner
            self._ctx = self._skid._ctx
        else:
            # params: overrides of the configuration constants by the reference's kwarg names (config.py), e.g.
            # dict(max_dist=5.5, max_length=10, smoothing=0.1); None = the reference's defaults
            self._ctx = _capi.Context(device=device, mission=int(self.mission), params=params)

    @property
    def relocalization_info(self):
        """reference full_pipeline.py:209-217: None without a relocalizer or before relocalization."""
        from .skidpad import RelocalizationInformation

        if self._accel is not None:
            if not self._accel.is_relocalized:
                return None
            # relocalization_information.py:17-35: images of (0, 0) and (1, 0) under the transform to the known frame
            o, _ = self._accel.to_known_frame(np.array([0.0, 0.0]), 0)
            e, _ = self._accel.to_known_frame(np.array([1.0, 0.0]), 0)
            return RelocalizationInformation(np.array([o[0], o[1]]), float(np.arctan2(e[1] - o[1], e[0] - o[0])))
        if self._skid is None or self._skid_info is None or not int(self._skid_info["relocalized"]):
            return None
        return RelocalizationInformation(np.array(self._skid_info["translation"]), float(self._skid_info["rotation"]))

    def set_global_path(self, global_path):  # reference full_pipeline.py:81-82
        """With a global path (n,2) every following path is drawn from it (core_calculate_path.py:514-529); None
        switches back to planning from the matched cones."""
        if self._skid is not None:
            raise RuntimeError("the skidpad mission plans along its own known path")
        self.global_path = None if global_path is None else np.ascontiguousarray(global_path, dtype=np.float64).reshape(-1, 2)
        (self._multi or self._ctx).set_global_path(self.global_path)

    def _accelerate(self, cones, xyt, pose, return_intermediate_results):
        """acceleration / ebs_test (full_pipeline.py:118-140,178-194): relocalize once on the host, then plan along the
        known path in the known frame; sorting and matching are skipped, the path goes back to the caller's frame."""
        acc = self._accel
        position, direction = pose[:2], pose[2:]
        by_type = cones if not (isinstance(cones, np.ndarray) and cones.ndim == 2 and cones.shape[1] == 3) else [xyt[:, :2]]
        acc.attempt(by_type, position, direction)
        if acc.is_relocalized:
            yaw = np.arctan2(direction[1], direction[0])
            position, yaw = acc.to_known_frame(position, yaw)
            direction = np.array([np.cos(yaw), np.sin(yaw)])
            # full_pipeline.py:134: the known path replaces whatever global path was set before, on every call
            from .acceleration import known_path

            kp = known_path()
            if self.global_path is None or self.global_path.shape != kp.shape or not np.array_equal(self.global_path, kp):
                self.set_global_path(kp)
        pose_k = np.concatenate([position, direction])[None]
        off0, none = np.zeros(2, np.int32), np.zeros((0, 3))
        if self._prev is not None:
            r = self._ctx.plan_batch_sequential(off0, none, pose_k, self._prev[None])[0]
        else:
            r = self._ctx.plan_batch(off0, none, pose_k)[0]
        raise_for_status(r["status"])
        path = np.array(r["path"][: self._ctx.horizon])
        self._prev = path.copy()  # the path stage keeps its history in the frame it computed in
        if acc.is_relocalized:
            path[:, 1:3], _ = acc.to_original_frame(path[:, 1:3], np.zeros(len(path)))
        if not return_intermediate_results:
            return path
        e2, ei = np.zeros((0, 2)), np.zeros(0, dtype=int)
        return (path, e2, e2.copy(), e2.copy(), e2.copy(), ei, ei.copy())

    # ---- batched form -------------------------------------------------------------------
    def plan_batch(self, cone_offsets, cones_xyt, poses) -> np.ndarray:
        """Structured array (one row per frame, dtype _capi.RESULT_DTYPE)."""
        if self._skid is not None:
            raise RuntimeError("the skidpad mission is stateful: use skidpad.SkidpadBatch.step for batches of planner instances")
        if self._multi is not None:  # contiguous frame ranges, one per GPU, from this thread (multi.py)
            return self._multi.plan_batch(cone_offsets, cones_xyt, poses)
        return self._ctx.plan_batch(cone_offsets, cones_xyt, poses)

    # ---- reference-shaped single-frame call ---------------------------------------------
    def calculate_path_in_global_frame(
        self,
        cones: List[np.ndarray],
        vehicle_position: np.ndarray,
        vehicle_direction: Union[np.ndarray, float],
        return_intermediate_results: bool = False,
    ):
        vehicle_direction = _direction_to_array(vehicle_direction)
        xyt = flatten_cones_by_type_array(cones)
        pose = np.concatenate([np.asarray(vehicle_position, dtype=np.float64).reshape(2), vehicle_direction])
        if self._skid is not None:
            # skidpad: stateful call; sorting and matching are skipped (full_pipeline.py:138-140)
            res, info = self._skid.step(np.array([0, len(xyt)], np.int32), xyt, pose[None])
            self._skid_info = info[0]
            raise_for_status(res[0]["status"])
            path = np.array(res[0]["path"][: self._ctx.horizon])
            if not return_intermediate_results:
                return path
            e2, ei = np.zeros((0, 2)), np.zeros(0, dtype=int)
            return (path, e2, e2.copy(), e2.copy(), e2.copy(), ei, ei.copy())
        if self._accel is not None:
            return self._accelerate(cones, xyt, pose, return_intermediate_results)
        off1 = np.array([0, len(xyt)], np.int32)
        if self.stateful and self._prev is not None:
            r = self._ctx.plan_batch_sequential(off1, xyt, pose[None], self._prev[None])[0]
        else:
            r = self._ctx.plan_batch(off1, xyt, pose[None])[0]
        raise_for_status(r["status"])
        path = np.array(r["path"][: self._ctx.horizon])  # (mpc_prediction_horizon, 4) like the reference's return value
        if self.stateful:
            self._prev = path.copy()
        if not return_intermediate_results:
            return path
        nl, nr = int(r["n_left"]), int(r["n_right"])
        ml, mr = int(r["n_left_v"]), int(r["n_right_v"])
        sorted_left = xyt[r["left_idx"][:nl], :2]
        sorted_right = xyt[r["right_idx"][:nr], :2]
        return (
            path,
            sorted_left,
            sorted_right,
            np.array(r["left_v"][:ml]),
            np.array(r["right_v"][:mr]),
            np.array(r["l2r"][:ml], dtype=np.int64),
            np.array(r["r2l"][:mr], dtype=np.int64),
        )
