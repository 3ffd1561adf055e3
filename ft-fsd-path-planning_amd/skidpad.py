"""Skidpad mission host side (BASELINE config 5): stateful planner instances on one GPU.

Mirrors what ``PathPlanner(MissionTypes.skidpad)`` does in the reference (full_pipeline.py:118-194):
relocalization until it succeeds, then the known-map window path, path returned in the caller's frame,
``relocalization_info`` (relocalization_information.py:12-35).  Constant inputs handed to the library
(include/fsdp.h fsdp_skidpad_set_tables):
  * the known skidpad path (data/skidpad_path.npy = the (5786,2) table of the reference's
    relocalization/skidpad/skidpad_path_data.py — track DATA, not code),
  * the fixed-seed noise table numpy.random.RandomState(42).randn(1140,3,2) (skidpad_relocalizer.py:38,52),
The two reference circle centres (skidpad_relocalizer.py:172-183) and the table spacing are computed by the library on
the device (csrc/skidpad_kernel.h skid_centers_kernel); ``SkidpadBatch.constants`` reads them back.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from pathlib import Path

import numpy as np

from . import _capi

DATA = Path(__file__).resolve().parent / "data" / "skidpad_path.npy"

INFO_DTYPE = np.dtype([("relocalized", "<i4"), ("index_along_path", "<i4"), ("translation", "<f8", (2,)), ("rotation", "<f8")], align=True)


def load_tables(table: np.ndarray | None = None):
    """(known path table, fixed-seed noise table) — the constant DATA handed to fsdp_skidpad_set_tables; what the
    reference derives from the table (reference centres, spacing) is computed on the device."""
    table = np.load(DATA) if table is None else np.ascontiguousarray(table, dtype=np.float64)
    noise = np.random.RandomState(42).randn(1140, 3, 2)
    return table, noise


@dataclass
class RelocalizationInformation:
    """reference relocalization/relocalization_information.py:12-35"""

    translation: np.ndarray
    rotation: float


class SkidpadBatch:
    """n independent skidpad planners advanced in lock-step (one frame of every instance per step).
    ``devices`` (a list of GPU indices or "all"): the instances are sharded over those GPUs from this one process and the
    object is a ``multi.MultiSkidpadBatch`` with the same methods."""

    def __new__(cls, n_instances: int = 1, device: int | None = None, table: np.ndarray | None = None, params: dict | None = None,
                devices=None):
        if devices is not None:
            from .multi import MultiSkidpadBatch

            return MultiSkidpadBatch(n_instances, None if devices == "all" else devices, table=table, params=params)
        return super().__new__(cls)

    def __init__(self, n_instances: int = 1, device: int | None = None, table: np.ndarray | None = None, params: dict | None = None,
                 devices=None):
        self._ctx = _capi.Context(device=device, mission=2, params=params)
        self.n = int(n_instances)
        table, noise = load_tables(table)
        self.tables = (table, noise)
        L, h = self._ctx._lib, self._ctx._h
        dp = _capi._dp
        self._ctx._check(L.fsdp_skidpad_set_tables(h, dp(table), ctypes.c_int(len(table)), dp(noise), ctypes.c_int(noise.size)),
                         "fsdp_skidpad_set_tables")
        self.reset()

    @property
    def constants(self):
        """(reference centres [[right xy], [left xy]], table spacing) as derived from the table on the device."""
        out = np.zeros(5)
        self._ctx._check(self._ctx._lib.fsdp_skidpad_constants(self._ctx._h, _capi._dp(out)), "fsdp_skidpad_constants")
        return out[:4].reshape(2, 2).copy(), float(out[4])

    def reset(self):
        self._ctx._check(self._ctx._lib.fsdp_skidpad_reset(self._ctx._h, ctypes.c_int(self.n)), "fsdp_skidpad_reset")

    def step(self, cone_offsets, cones_xyt, poses):
        off, cones, poses, n = self._ctx._prep(cone_offsets, cones_xyt, poses)
        assert n == self.n
        res = np.zeros(n, dtype=self._ctx.result_dtype)
        info = np.zeros(n, dtype=INFO_DTYPE)
        self._ctx._check(
            self._ctx._lib.fsdp_skidpad_step(self._ctx._h, ctypes.c_int(n), _capi._ip(off), _capi._dp(cones), _capi._dp(poses),
                                             ctypes.c_void_p(res.ctypes.data), ctypes.c_void_p(info.ctypes.data)),
            "fsdp_skidpad_step")
        return res, info

    def replay(self, frames, depth: int = 32, compact: bool = False):
        """A known sequence of frames [(cone_offsets, cones_xyt, poses), ...] for all planners, submitted ``depth`` steps
        ahead (the counterpart of the reference's frame loop over a recording, demo/json_demo.py:103-131, for many planners
        at once): consecutive steps share their launches (include/fsdp.h, fsdp_skidpad_submit).  Yields (results, info) per
        step, in order — the bits of ``step`` called once per frame."""
        self.set_overlap(depth)
        ring = [_capi.pinned_empty(self.n, self._ctx.path_result_dtype if compact else self._ctx.result_dtype) for _ in range(depth + 1)]
        inflight = []
        for k, f in enumerate(frames):
            if len(inflight) == depth:
                res, info = self.collect(inflight.pop(0))
                yield res.copy(), info
            inflight.append(self.submit(*f, out=ring[k % (depth + 1)]))
        for t in inflight:
            res, info = self.collect(t)
            yield res.copy(), info

    def submit(self, cone_offsets, cones_xyt, poses, out=None, info=None, compact: bool = False) -> "_capi.Ticket":
        """One step as a ticket (fsdp_skidpad_submit).  compact=True (or an ``out`` array of PATH_RESULT_DTYPE): results as
        fsdp_path_result records — path, status, fallback bits: everything a skidpad step produces — 1.3 KB instead of 2.4 KB
        per planner and step (fsdp_skidpad_submit_compact).  Up to the context's overlap depth tickets may be outstanding
        (``set_overlap``, at most 32); steps submitted ahead share their launches, a step that is collected at once gets
        launches of its own."""
        off, cones, poses, n = self._ctx._prep(cone_offsets, cones_xyt, poses)
        assert n == self.n
        if out is None:
            out = _capi.pinned_empty(n, self._ctx.path_result_dtype if compact else self._ctx.result_dtype)
        compact = out.dtype == self._ctx.path_result_dtype
        assert compact or out.dtype == self._ctx.result_dtype
        if info is None:
            info = np.zeros(n, dtype=INFO_DTYPE)
        t = ctypes.c_longlong(-1)
        fn = self._ctx._lib.fsdp_skidpad_submit_compact if compact else self._ctx._lib.fsdp_skidpad_submit
        self._ctx._check(fn(self._ctx._h, ctypes.c_int(n), off.ctypes.data, cones.ctypes.data if len(cones) else None, poses.ctypes.data,
                            out.ctypes.data, info.ctypes.data, ctypes.byref(t)), "fsdp_skidpad_submit")
        return _capi.Ticket(int(t.value), out, info, (off, cones, poses))

    def collect(self, ticket):
        self._ctx._check(self._ctx._lib.fsdp_collect(self._ctx._h, ctypes.c_longlong(ticket.id)), "fsdp_collect")
        ticket._keep = None
        return ticket.out, ticket.info

    def set_overlap(self, depth: int):
        self._ctx.set_overlap(depth)

    def close(self):
        """Destroy the context (its device buffers, streams and page-locked blocks) now instead of when the object is collected."""
        self._ctx.close()

    def time_groups(self, enable: bool = True):
        """HIP events around the packed kernels of every group of steps of the replays that follow (fsdp_skidpad_time_groups)."""
        self._ctx._check(self._ctx._lib.fsdp_skidpad_time_groups(self._ctx._h, ctypes.c_int(1 if enable else 0)), "fsdp_skidpad_time_groups")

    def group_times(self):
        """({kernel name: summed ms}, groups, (instance, step) pairs) since time_groups(True)."""
        ms = (ctypes.c_float * 5)()
        ng, nf = ctypes.c_int(), ctypes.c_longlong()
        names = ctypes.create_string_buffer(256)
        self._ctx._check(self._ctx._lib.fsdp_skidpad_group_times(self._ctx._h, ms, ctypes.byref(ng), ctypes.byref(nf), names, 256), "fsdp_skidpad_group_times")
        return dict(zip(names.value.decode().split(","), [float(x) for x in ms])), int(ng.value), int(nf.value)

    def time_path(self, iters: int) -> float:
        t = ctypes.c_float()
        self._ctx._check(self._ctx._lib.fsdp_skidpad_time_path(self._ctx._h, ctypes.c_int(iters), ctypes.byref(t)), "fsdp_skidpad_time_path")
        return float(t.value)
