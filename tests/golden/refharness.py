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


def knn_boundary_tie(xyt: np.ndarray) -> bool:
    """True when some cone has an EXACT tie among its 8 smallest squared distances (the reference's own expansion-form
    matrix, utils/math_utils.py:120-150): create_adjacency_matrix takes the 5 nearest with a full unstable np.argsort
    (adjacency_matrix.py:55), so which of the tied cones is a neighbour depends on the NumPy build.  Only the
    mirror-symmetric demo scenarios do this; the parity tests skip the end-configuration counts on such frames (the final
    configurations agree anyway)."""
    from fsd_path_planning.utils.math_utils import my_cdist_sq_euclidean

    xy = np.ascontiguousarray(xyt[:, :2], dtype=float)
    if len(xy) < 3:
        return False
    d = my_cdist_sq_euclidean(xy, xy).copy()
    np.fill_diagonal(d, np.inf)
    d.sort(axis=1)
    head = d[:, : min(8, d.shape[1])]
    return bool((np.diff(head, axis=1) == 0).any())


def planner_with_params(m, params):
    """A reference PathPlanner whose three stage objects are built with `params` (the kwargs of config.py's factories,
    overridden by name) — exactly what a user of the stage classes would construct by hand."""
    pp = m["PathPlanner"](m["MissionTypes"].trackdrive)
    if not params:
        return pp
    from fsd_path_planning import config as cfg
    from fsd_path_planning.calculate_path.core_calculate_path import CalculatePath
    from fsd_path_planning.cone_matching.core_cone_matching import ConeMatching
    from fsd_path_planning.sorting_cones.core_cone_sorting import ConeSorting

    mission = m["MissionTypes"].trackdrive
    sk = cfg.get_cone_sorting_config(mission)
    sk["experimental_performance_improvements"] = False
    mk = cfg.get_default_matching_kwargs(mission)
    mk["matches_should_be_monotonic"] = False  # the pipeline's choice (full_pipeline.py:65)
    pk = {**cfg.get_path_calculation_config(mission), **cfg.get_cone_fitting_config(mission)}
    for k, v in params.items():
        hit = False
        for d in (sk, mk, pk):
            if k in d:
                d[k] = v
                hit = True
        assert hit, k
    pp.cone_sorting = ConeSorting(**sk)
    pp.cone_matching = ConeMatching(**mk)
    pp.pathing = CalculatePath(**pk)
    return pp


def run_frame(xyt: np.ndarray, pose: np.ndarray, flattened: bool = True, params=None):
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
    # first_k per side: TraceSorter.select_first_k_starting_cones (core_trace_sorter.py:409-465), called once per side
    orig_fk = cts.TraceSorter.select_first_k_starting_cones
    captured["first_k"] = {}
    captured["first_k_tie"] = {}

    def fk_wrapper(self, car_position, car_direction, cones, cone_type):
        out = orig_fk(self, car_position, car_direction, cones, cone_type)
        captured["first_k"][int(cone_type)] = None if out is None else np.array(out, dtype=np.int64)
        # select_starting_cone orders the candidates with np.argsort (unstable): on an EXACT distance tie of the two
        # closest candidates the pick depends on the NumPy build (AVX-512 argsort here) — recorded so that the parity
        # tests can skip the start-cone comparison on such frames (mirror-symmetric demo scenarios)
        dist, valid = self.mask_cone_can_be_first_in_config(car_position, car_direction, cones, cone_type)
        d = np.where(valid, dist, np.inf)
        captured["first_k_tie"][int(cone_type)] = bool(np.isfinite(d).any() and (d == d.min()).sum() > 1)
        return out

    cts.TraceSorter.select_first_k_starting_cones = fk_wrapper
    # the smoothing splines of the frame: every scipy splprep call inside calculate_path_in_global_frame, in call order
    # (utils/spline_fit.py:117; the planner's constructor fits its initial path before the wrapper is armed)
    import fsd_path_planning.utils.spline_fit as sfm

    orig_splprep = sfm.splprep
    captured["fits"] = []
    armed = [False]

    def splprep_wrapper(x, **kw):
        res = orig_splprep(x, **kw)
        if armed[0]:
            (t, c, k), _u = res
            captured["fits"].append((int(k), np.array(t), np.array(c[0]), np.array(c[1])))
        return res

    sfm.splprep = splprep_wrapper
    try:
        pp = planner_with_params(m, params)
        armed[0] = True
        cones = np.ascontiguousarray(xyt, dtype=float) if flattened else split_by_type(xyt)
        try:
            out = pp.calculate_path_in_global_frame(cones, pose[:2].copy(), pose[2:].copy(), return_intermediate_results=True)
        except Exception as e:  # noqa
            return dict(status=type(e).__name__, msg=str(e), **captured)
    finally:
        cts.calc_final_configs_for_left_and_right = orig
        cts.TraceSorter.select_first_k_starting_cones = orig_fk
        sfm.splprep = orig_splprep
    path, sl, sr, lv, rv, l2r, r2l = out
    captured["knn_tie"] = knn_boundary_tie(xyt)
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
        first_k=captured["first_k"],
        first_k_tie=captured["first_k_tie"],
        knn_tie=captured["knn_tie"],
        fits=captured["fits"],
    )
