"""Shared parity rules (used by the oracle tests on CPU and the HIP tests on the GPU).

Bar (BASELINE.json north_star): sorted cone index arrays bit-equal; spline xy / curvature
samples within 1e-5 L-inf.

One documented exception class — "sample-count flip": the reference decides the number of
dense path samples as ceil(max_u / (path_length/40/3)) (path_parameterization.py:146,
spline_fit.py:43) where max_u and path_length are the same sum rounded two different ways,
i.e. the ratio is mathematically exactly 120 and the outcome (120 or 121 samples) hangs on
the last bit.  It is reproduced bit-exactly as long as every float feeding it is bit-exact;
the only inputs that are not reproducible across libm implementations are the sin/cos/atan2
values of the arc extension (core_calculate_path.py:316-321; NumPy's AVX-512 atan2 differs
from glibc's in ~8 % of arguments).  Frames that took the arc branch are therefore allowed to
differ by exactly that flip (first column step pattern), and are counted.
"""
from __future__ import annotations

import numpy as np

PATH_TOL = 1e-5
ARC_FLAG = 16


class ArcLibm:
    """tests/golden/arc_libm_level.npz (arc_libm_golden.py): the reference's paths on the arc-extension frames of a golden set
    with NumPy's SIMD sin / cos / arctan2 kernels disabled (libm level), and the frames on which that run differs from the
    AVX-512 run the goldens were captured with — i.e. on which the REFERENCE is not reproducible across machines."""

    def __init__(self, golden_dir, name):
        f = np.load(golden_dir / "arc_libm_level.npz")
        self.name = name
        self.frames = f[f"{name}__frames"].tolist() if f"{name}__frames" in f else []
        self._path = f[f"{name}__path"] if self.frames else None
        self._differs = f[f"{name}__differs"] if self.frames else None
        # expected differences (> 1e-5) of a result computed with `math` = "libm" (the oracle's default mode: host libm) or "det"
        # (det_math.h: kernels, emulator, oracle in det mode) from the two captures
        self._exp = {(m, lvl): (f[f"{name}__{m}_vs_{lvl}"] if self.frames else None) for m in ("libm", "det") for lvl in ("libm_level", "avx512")}
        self._idx = {fr: i for i, fr in enumerate(self.frames)}

    def __contains__(self, k):
        return int(k) in self._idx

    def path(self, k):
        return self._path[self._idx[int(k)]]

    def differs(self, k):
        return bool(self._differs[self._idx[int(k)]])

    def expected(self, k, math, level):
        return bool(self._exp[(math, level)][self._idx[int(k)]])

    def flips(self, math):
        """Frames on which a result computed with this math differs from the AVX-512 golden (the committed <set>.npz)."""
        return sorted(fr for fr in self.frames if self.expected(fr, math, "avx512"))

    @property
    def reference_differs_from_itself(self):
        return sorted(fr for fr in self.frames if self.differs(fr))


def _linf(p, q):
    e = np.abs(p - q)
    return 0.0 if np.isnan(e).all() else float(np.nanmax(e))


def compare_frame(res, g, k, require_exact_match=True, arc=None, math="det"):
    """res: one structured result row (oracle_lib.RESULT_DTYPE-like); g: golden dict; k: frame.
    Returns (category, detail).  category in {'ok','ref_undefined','flip','IDX','MATCH','PATH','STATUS'}
    arc (ArcLibm of the set): on arc-extension frames the result is held against BOTH captures of the reference (libm level and
    AVX-512) and must differ from each (by the sample-count flip) on exactly the frames recorded for its `math` ("libm": host
    libm — none against the libm level, the reference's own self-differences against AVX-512; "det": det_math.h — additionally the
    two frames on which the correctly rounded sin / cos / atan2 differ from glibc's last bit).  'flip' = differs from the AVX-512
    golden as recorded; any unrecorded difference, or a recorded one that is missing, is 'PATH'."""
    ref_ok = bool(g["ok"][k])
    st = int(res["status"])
    if not ref_ok:
        return ("ref_undefined", "") if st >= 100 else ("STATUS", f"reference raised {g['exc'][k]} but status={st}")
    if st != 0:
        return "STATUS", f"status {st} but reference returned normally"
    nl, nr = int(g["n_left"][k]), int(g["n_right"][k])
    if int(res["n_left"]) != nl or int(res["n_right"]) != nr or not (
        np.array_equal(res["left_idx"][:nl], g["left_idx"][k][:nl]) and np.array_equal(res["right_idx"][:nr], g["right_idx"][k][:nr])
    ):
        return "IDX", f"{res['left_idx']} vs {g['left_idx'][k]} | {res['right_idx']} vs {g['right_idx'][k]}"
    # per-stage intermediates (fixtures generated with them): start cones, number of end configurations after the
    # post-filters and the cost of the best one, per side — a compensating error between stages would show here
    if "first_k_left" in g and "first_k_left" in res.dtype.names:
        for si, side in enumerate(("left", "right")):
            if g["first_k_tie"][k, si]:
                continue  # exact tie of the two closest start candidates: the reference's pick is its argsort's (unstable)
            if not np.array_equal(res[f"first_k_{side}"], g[f"first_k_{side}"][k]):
                return "IDX", f"first_k_{side}: {res[f'first_k_{side}']} vs {g[f'first_k_{side}'][k]}"
            if g["knn_tie"][k]:
                continue  # exact tie among nearest neighbours: the reference's adjacency is its argsort's (unstable)
            if int(res[f"n_configs_{side}"]) != int(g[f"n_configs_{side}"][k]):
                return "IDX", f"n_configs_{side}: {res[f'n_configs_{side}']} vs {g[f'n_configs_{side}'][k]}"
            if int(g[f"n_configs_{side}"][k]) > 0:
                a_, b_ = float(res[f"best_cost_{side}"]), float(g[f"best_cost_{side}"][k])
                if not abs(a_ - b_) <= 1e-9 * max(1.0, abs(b_)):
                    return "IDX", f"best_cost_{side}: {a_} vs {b_}"
    ml, mr = int(g["n_left_v"][k]), int(g["n_right_v"][k])
    if int(res["n_left_v"]) != ml or int(res["n_right_v"]) != mr:
        return "MATCH", "virtual cone counts differ"
    if not (np.array_equal(res["l2r"][:ml], g["l2r"][k][:ml]) and np.array_equal(res["r2l"][:mr], g["r2l"][k][:mr])):
        return "MATCH", "match indices differ"
    dv = max(np.abs(res["left_v"][:ml] - g["left_v"][k][:ml]).max(initial=0), np.abs(res["right_v"][:mr] - g["right_v"][k][:mr]).max(initial=0))
    if (require_exact_match and dv != 0) or dv > 1e-9:
        return "MATCH", f"virtual cone positions differ by {dv}"
    p, q = res["path"], g["path"][k]
    if not np.array_equal(np.isnan(p), np.isnan(q)):
        return "PATH", "nan pattern differs"
    e = np.nanmax(np.abs(p - q)) if not np.isnan(q).all() else 0.0
    is_arc = bool(int(res["path_fallback"]) & ARC_FLAG)
    if arc is not None and (is_arc or k in arc):
        if not (is_arc and k in arc):
            return "PATH", f"arc-extension frame according to {'the result' if is_arc else 'the fixture'} only"
        for level, ref_path, err in (("libm_level", arc.path(k), _linf(p, arc.path(k))), ("avx512", q, e)):
            if arc.expected(k, math, level) != (err > PATH_TOL):
                return "PATH", f"arc frame: L-inf {err} against the reference at the {level} level, recorded for {math} math: {'differs' if arc.expected(k, math, level) else 'equal'}"
            if err > PATH_TOL and not is_sample_count_flip(p, ref_path):
                return "PATH", f"arc frame: L-inf {err} against the {level} capture is not the sample-count flip"
        return ("flip", e) if e > PATH_TOL else ("ok", e)
    if e <= PATH_TOL:
        return "ok", e
    if int(res["path_fallback"]) & ARC_FLAG and is_sample_count_flip(p, q):
        return "flip", e
    return "PATH", f"L-inf {e}"


def is_sample_count_flip(p, q):
    """True when both paths start identically and their arc-length columns are the 40-index
    resamplings of 120 vs 121 (or L vs L+-1) dense samples of the same step."""
    rows = ~np.isnan(p[:, 0])  # (a record's rows beyond the context's horizon are NaN: the wide build's 64 against a horizon of 40)
    p, q = p[rows], q[rows]
    if abs(p[0, 1] - q[0, 1]) > 1e-5 or abs(p[0, 2] - q[0, 2]) > 1e-5:
        return False
    sp, sq = p[-1, 0], q[-1, 0]
    step = max(np.diff(p[:, 0]).min(), 1e-9) / 2.0
    return abs(sp - sq) < 2.5 * step * 2 and abs(sp - sq) > 1e-5


def assert_intermediates_equal(res, ref, ok, cost_rtol=0.0):
    """Device (or emulated kernels) against the oracle: per-stage intermediates of the sorting stage — start cones per
    side and number of end configurations after the post-filters exactly, cost of the best one (where a side has
    configurations at all) bit for bit under the emulator (same libm) and within cost_rtol on the GPU (the costs hold
    atan2 / acos values of the device's libm; no discrete decision depends on their last bit in these sets)."""
    for f in ("first_k_left", "first_k_right", "n_configs_left", "n_configs_right"):
        assert np.array_equal(res[f][ok], ref[f][ok]), f
    for side in ("left", "right"):
        has = ok & (ref[f"n_configs_{side}"] > 0)
        a_, b_ = res[f"best_cost_{side}"][has], ref[f"best_cost_{side}"][has]
        assert (np.abs(a_ - b_) <= cost_rtol * np.maximum(1.0, np.abs(b_))).all(), f"best_cost_{side}"


def host_libm_is_the_fixture_machines(golden_dir):
    """Bit-for-bit comparisons of host-libm results with the reference hold where libm returns what it returned when the goldens
    were captured (glibc's sin / cos / atan2 / pow differ in a last bit between its FMA and non-FMA variants): probe values
    recorded with the captures (arc_libm_level.npz) + one argument on which glibc's pow(x, 2) is not the rounded product."""
    import math

    f = np.load(golden_dir / "arc_libm_level.npz")
    here = np.array([math.atan2(0.3, 1.7), math.sin(1.234567), math.cos(2.3456789), math.atan2(-2.5, 0.11)])
    pow_probe = math.pow(float.fromhex("-0x1.b504fa57a4d82p+2"), 2.0) == float.fromhex("0x1.7504ff64002b3p+5")  # (x * x ends in ...2b2)
    return bool(np.array_equal(here, f["probe_libm"]) and pow_probe)
