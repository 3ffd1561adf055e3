"""Runs the REFERENCE (imported read-only from /root/reference with an identity-jit numba
stub) and captures per-frame outputs + intermediates.  Build-container only: the reference
never travels to the GPU box; what travels are the .npz fixtures this produces
(tests/golden/make_golden.py).
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference")


def available() -> bool:
    return (REF / "fsd_path_planning").is_dir()


_loaded = {}


def load():
    if _loaded:
        return _loaded
    sys.path.insert(0, str(HERE / "_refstubs"))
    sys.path.insert(1, str(REF))
    import fsd_path_planning  # noqa
    from fsd_path_planning import ConeTypes, MissionTypes, PathPlanner
    from fsd_path_planning.sorting_cones.trace_sorter import core_trace_sorter as cts

    _loaded.update(dict(PathPlanner=PathPlanner, MissionTypes=MissionTypes, ConeTypes=ConeTypes, cts=cts))
    return _loaded


def split_by_type(xyt):
    return [np.ascontiguousarray(xyt[xyt[:, 2] == t, :2]) for t in range(5)]


def run_frame(xyt: np.ndarray, pose: np.ndarray, flattened: bool = True):
    """Fresh PathPlanner (trackdrive) on one frame.  Returns dict with status 'ok' or the
    exception class name.  The frame is handed over as the pre-flattened (N,3) array
    (accepted by the sorter, core_trace_sorter.py:40-41) so that index spaces coincide."""
    m = load()
    cts = m["cts"]
    captured = {}
    orig = cts.calc_final_configs_for_left_and_right

    def wrapper(ls, lc, rs, rc, cones, pos, d):
        out = orig(ls, lc, rs, rc, cones, pos, d)
        captured["left_config"] = np.array(out[0], dtype=np.int64)
        captured["right_config"] = np.array(out[1], dtype=np.int64)
        captured["left_costs"] = None if ls is None else np.array(ls)
        captured["right_costs"] = None if rs is None else np.array(rs)
        captured["left_configs"] = None if lc is None else np.array(lc)
        captured["right_configs"] = None if rc is None else np.array(rc)
        return out

    cts.calc_final_configs_for_left_and_right = wrapper
    try:
        pp = m["PathPlanner"](m["MissionTypes"].trackdrive)
        cones = np.ascontiguousarray(xyt, dtype=float) if flattened else split_by_type(xyt)
        try:
            out = pp.calculate_path_in_global_frame(cones, pose[:2].copy(), pose[2:].copy(), return_intermediate_results=True)
        except Exception as e:  # noqa
            return dict(status=type(e).__name__, msg=str(e), **captured)
    finally:
        cts.calc_final_configs_for_left_and_right = orig
    path, sl, sr, lv, rv, l2r, r2l = out
    lc = captured["left_config"]
    rc = captured["right_config"]
    return dict(
        status="ok",
        path=np.array(path),
        left_config=lc[lc != -1],
        right_config=rc[rc != -1],
        left_v=np.array(lv),
        right_v=np.array(rv),
        l2r=np.array(l2r, dtype=np.int64),
        r2l=np.array(r2l, dtype=np.int64),
        left_costs=captured["left_costs"],
        right_costs=captured["right_costs"],
        left_configs=captured["left_configs"],
        right_configs=captured["right_configs"],
    )
