#!/usr/bin/env python3
"""The reference's outputs on ARC-EXTENSION frames at the libm level of NumPy's CPU dispatch -> tests/golden/arc_libm_level.npz

Why: the committed goldens were captured with this NumPy build's default dispatch (AVX-512: its own SVML-derived
arctan2 / sin / cos); with those kernels disabled (NPY_DISABLE_CPU_FEATURES) NumPy calls libm, and on a few arc frames
(core_calculate_path.py:301-324: atan2, 49 x sin / cos feed the float chain that decides 120 <-> 121 dense samples) the
reference then returns a path that differs from its own AVX-512 run by 0.1-0.2 m (profiles/r03_reference_dispatch_levels.txt).
The oracle with host libm reproduces the libm-level reference on every arc frame (asserted below); the kernels (and the
oracle in det-math mode) use correctly rounded sin / cos / atan2 (det_math.h: the device cannot call glibc), which is glibc's
value except for a last bit on a few arguments — on 2 of the 215 arc frames of all sets that bit flips the sample count
(params_sort 98, params_no_unknown 6).  So the parity tests assert, per golden set and per math (tests/parity.py ArcLibm):
    * host-libm results: equal (1e-5) to the libm-level reference captured HERE on every arc frame, and different from the
      AVX-512 goldens on EXACTLY the frames on which the reference differs from itself (`<set>__differs`: fuzz 339 and 347,
      params_no_unknown 97);
    * det-math results (kernels, emulator, oracle det mode): the same, plus exactly the recorded frames
      (`<set>__det_vs_libm_level`, `<set>__det_vs_avx512`) — a third flip, or a missing one, fails.
Build-container only (imports /root/reference through refharness).  Stored: frame indices, paths, the differs flags, the
dispatch level and libm probe values.  No reference source is stored.

    python tests/golden/arc_libm_golden.py            # parent: children per level, writes the .npz
"""
import json
import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
LIBM_LEVEL = "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX2 FMA3"  # -> SSE4.2: sin / cos / arctan2 from libm
SETS = ["scenarios", "cfg2_color", "cfg3_nocolor", "cfg4_200cones", "cfg4_noisy_nocolor", "fuzz", "big_frames", "lattice",
        "odd_inputs", "nonfinite_cones", "nonfinite_poses",
        "params_sort", "params_path", "params_monotonic", "params_deg2", "params_deg1", "params_horizon", "params_no_unknown"]
# parameter sets beyond the standard shapes (make_golden.py --params-r5): (F, 64, 4) paths, checked with the oracle's wide build.
# `--wide` captures only these and MERGES them into the committed file (the other sets' captures stay as they are), after
# checking that this machine's NumPy / libm return the probe values stored with them.
SETS_WIDE = ["params_wide_sort", "params_wide_horizon", "params_wide_all"]


def set_params(g):
    if "param_names" not in g:
        return None
    prm = {}
    for k, v in zip(g["param_names"].tolist(), g["param_values"].tolist()):
        prm[k] = bool(v) if k in ("matches_should_be_monotonic", "use_unknown_cones") else (int(v) if float(v).is_integer() and k in ("max_deg", "mpc_prediction_horizon", "max_n_neighbors", "max_length") else float(v))
    return prm


def child(out_path):
    import numpy as np

    sys.path.insert(0, str(HERE))
    sys.path.insert(0, str(HERE.parent))
    import refharness

    todo = json.loads(os.environ["ARC_TODO"])
    out = {}
    for name, frames in todo.items():
        g = np.load(HERE / f"{name}.npz")
        prm = set_params(g)
        flattened = not (prm and prm.get("use_unknown_cones") is False)
        paths = np.full((len(frames), g["path"].shape[1], 4), np.nan)
        for k, f in enumerate(frames):
            xyt = g["cones"][g["offsets"][f]: g["offsets"][f + 1]]
            with np.errstate(all="ignore"):
                r = refharness.run_frame(xyt, g["poses"][f], params=prm, flattened=flattened)
            if r["status"] == "ok":
                paths[k, : len(r["path"])] = r["path"]
        out[name] = paths
    out["probe"] = np.array([np.arctan2(0.3, 1.7), np.sin(1.234567), np.cos(2.3456789), np.arctan2(-2.5, 0.11)])
    np.savez(out_path, **out)


def main(wide=False):
    import numpy as np

    sys.path.insert(0, str(HERE.parent))
    if wide:
        import oracle_lib_wide as oracle_lib
    else:
        import oracle_lib

    todo = {}
    for name in (SETS_WIDE if wide else SETS):
        g = np.load(HERE / f"{name}.npz")
        prm = set_params(g)
        with oracle_lib.math_mode(1):
            if prm:
                with oracle_lib.params(prm):
                    o = oracle_lib.plan_batch(g["offsets"], g["cones"], g["poses"], n_threads=4)
            else:
                o = oracle_lib.plan_batch(g["offsets"], g["cones"], g["poses"], n_threads=4)
        arc = [int(f) for f in np.nonzero((o["path_fallback"] & 16) != 0)[0] if g["ok"][f] and o["status"][f] == 0]
        if arc:
            todo[name] = arc
    res = {}
    for level, dis in (("default", ""), ("libm", LIBM_LEVEL)):
        out = f"/tmp/arc_libm_{level}.npz"
        env = dict(os.environ, ARC_TODO=json.dumps(todo))
        if dis:
            env["NPY_DISABLE_CPU_FEATURES"] = dis
        subprocess.run([sys.executable, __file__, "--child", out], env=env, check=True, stderr=subprocess.DEVNULL)
        res[level] = np.load(out)

    def err(a, b):
        e = np.abs(a - b).reshape(len(a), -1)
        return np.where(np.isnan(e), 0.0, e).max(axis=1)

    store = {"level": np.array("NPY_DISABLE_CPU_FEATURES=" + LIBM_LEVEL), "probe_default": res["default"]["probe"], "probe_libm": res["libm"]["probe"]}
    if wide:
        old = dict(np.load(HERE / "arc_libm_level.npz"))
        assert np.array_equal(old["probe_default"], store["probe_default"]) and np.array_equal(old["probe_libm"], store["probe_libm"]), \
            "this machine's NumPy / libm differ from the one the committed captures were taken on"
        store = old
    for name, frames in todo.items():
        g = np.load(HERE / f"{name}.npz")
        # the default-level run must BE the committed golden (same machine, same NumPy): otherwise the two captures are not comparable
        same = err(res["default"][name], g["path"][frames])
        assert (same == 0).all(), (name, "default-level rerun differs from the committed golden", same.max())
        d = err(res["libm"][name], g["path"][frames])
        store[f"{name}__frames"] = np.array(frames, np.int32)
        store[f"{name}__path"] = res["libm"][name]
        store[f"{name}__differs"] = d > 1e-5
        # the oracle in its two math modes against both captures (recorded, so that the tests assert the exact frames):
        # host libm == the libm-level reference everywhere; det_math.h (correctly rounded, what the kernels use) differs from
        # glibc's last bit on a few arguments, and where that flips the sample count the frame is listed here
        prm = set_params(g)
        for mode, key in ((0, "libm"), (1, "det")):
            with oracle_lib.math_mode(mode):
                if prm:
                    with oracle_lib.params(prm):
                        o = oracle_lib.plan_batch(g["offsets"], g["cones"], g["poses"], n_threads=4)
                else:
                    o = oracle_lib.plan_batch(g["offsets"], g["cones"], g["poses"], n_threads=4)
            store[f"{name}__{key}_vs_libm_level"] = err(o["path"][frames], res["libm"][name]) > 1e-5
            store[f"{name}__{key}_vs_avx512"] = err(o["path"][frames], g["path"][frames]) > 1e-5
        assert not store[f"{name}__libm_vs_libm_level"].any(), (name, "the oracle with host libm must equal the reference at the libm level")
        assert np.array_equal(store[f"{name}__libm_vs_avx512"], store[f"{name}__differs"]), name
        print(f"{name:22s} arc frames {len(frames):3d}   reference(libm level) vs reference(AVX-512 golden): {int((d > 1e-5).sum())} differ by > 1e-5"
              f" {[int(f) for f, x in zip(frames, d) if x > 1e-5]}, {int((d > 0).sum())} in any bit;  det_math.h vs libm level: "
              f"{[int(f) for f, x in zip(frames, store[name + '__det_vs_libm_level']) if x]}, vs AVX-512: {[int(f) for f, x in zip(frames, store[name + '__det_vs_avx512']) if x]}")
    np.savez_compressed(HERE / "arc_libm_level.npz", **store)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        main(wide="--wide" in sys.argv)
