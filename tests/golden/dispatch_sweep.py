#!/usr/bin/env python3
"""Is the REFERENCE itself machine-dependent on arc-extension frames?  

The reference decides the dense-sample count of its output as ceil(max_u / (path_length / 40 / 3)), a ratio that is exactly
120 in exact arithmetic (DESIGN.md "arithmetic contract"): a last-bit difference anywhere upstream flips 120 <-> 121 and
moves the 40 output samples by 0.1-0.2 m.  On frames that take the circular-arc extension (core_calculate_path.py:301-324)
libm values (atan2, 49 x sin / cos) enter that chain, and NumPy evaluates them with whatever SIMD kernels its CPU dispatch
picks on the machine at hand.  This script runs the reference (imported, build container only) on the arc frames of
tests/golden/fuzz.npz and on the skidpad replay under three dispatch levels of the SAME NumPy build —
    default (AVX-512 here) | AVX2 + FMA3 (AVX-512 disabled) | SSE4.2 (AVX-512, AVX2, FMA3 disabled)
via NPY_DISABLE_CPU_FEATURES — and counts the frames whose returned path differs between levels by more than 1e-5.
   python tests/golden/dispatch_sweep.py            # parent: runs the three children, prints the table
"""
import json
import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
LEVELS = {
    "avx512 (default)": "",
    "avx2+fma3": "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL",
    "sse42": "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX2 FMA3",
}


def child(out_path):
    import numpy as np

    sys.path.insert(0, str(HERE))
    sys.path.insert(0, str(HERE.parent))
    import refharness

    g = np.load(HERE / "fuzz.npz")
    arc = json.loads(os.environ["ARC_FRAMES"])
    paths = np.full((len(arc), 40, 4), np.nan)
    for k, f in enumerate(arc):
        xyt = g["cones"][g["offsets"][f]: g["offsets"][f + 1]]
        r = refharness.run_frame(xyt, g["poses"][f])
        if r["status"] == "ok":
            paths[k] = r["path"]
    # skidpad replay (stateful, one planner)
    m = refharness.load()
    d = json.load(open("/root/reference/fsd_path_planning/demo/skidpad.json"))
    pp = m["PathPlanner"](m["MissionTypes"].skidpad)
    sk = []
    for fr in d:
        cones = [np.array(c, dtype=float).reshape(-1, 2) for c in fr["slam_cones"]]
        sk.append(np.array(pp.calculate_path_in_global_frame(cones, np.array(fr["car_position"], float), np.array(fr["car_direction"], float))))
    feats = np.show_config(mode="dicts")["SIMD Extensions"]["found"] if hasattr(np, "show_config") else []
    # which of them survived NPY_DISABLE_CPU_FEATURES is what matters: probe the arctan2 of one argument pair that is known to differ
    np.savez(out_path, fuzz_arc=paths, skid=np.array(sk), probe=np.array([np.arctan2(0.3, 1.7), np.sin(1.234567), np.cos(2.3456789)]),
             feats=np.array(feats))


def main():
    import numpy as np

    sys.path.insert(0, str(HERE.parent))
    import oracle_lib

    g = np.load(HERE / "fuzz.npz")
    with oracle_lib.math_mode(1):
        o = oracle_lib.plan_batch(g["offsets"], g["cones"], g["poses"])
    arc = [int(f) for f in np.nonzero((o["path_fallback"] & 16) != 0)[0] if g["ok"][f]]
    res = {}
    for name, dis in LEVELS.items():
        out = f"/tmp/dispatch_{abs(hash(name))}.npz"
        env = dict(os.environ, ARC_FRAMES=json.dumps(arc))
        if dis:
            env["NPY_DISABLE_CPU_FEATURES"] = dis
        subprocess.run([sys.executable, __file__, "--child", out], env=env, check=True, stderr=subprocess.DEVNULL)
        res[name] = np.load(out)
    names = list(LEVELS)

    def differ(a, b):
        e = np.abs(a - b).reshape(len(a), -1)
        e = np.where(np.isnan(e), 0.0, e).max(axis=1)
        return int((e > 1e-5).sum()), int((e > 0).sum())

    print(f"arc frames of fuzz.npz: {len(arc)}; skidpad frames: {len(res[names[0]]['skid'])}")
    print("libm probe values per level (atan2, sin, cos):")
    for n in names:
        print(f"  {n:18s}", [float(x).hex() for x in res[n]["probe"]])
    print("frames whose path differs between two dispatch levels of the same NumPy build  [> 1e-5 | any bit]:")
    for i in range(len(names)):
        for j in range(i + 1, len(names)):
            fa = differ(res[names[i]]["fuzz_arc"], res[names[j]]["fuzz_arc"])
            sa = differ(res[names[i]]["skid"], res[names[j]]["skid"])
            print(f"  {names[i]:18s} vs {names[j]:10s}: fuzz arc frames {fa[0]:3d} | {fa[1]:3d}   skidpad {sa[0]:3d} | {sa[1]:3d}")
    # against the committed goldens (captured at the default level) and the oracle's two math modes
    gold = g["path"][arc]
    sk = np.load(HERE / "skidpad_sequence.npz")["path"]
    with oracle_lib.math_mode(0):
        o_libm = oracle_lib.plan_batch(g["offsets"], g["cones"], g["poses"])
    for n in names:
        print(f"  {n:18s} vs committed golden: fuzz arc {differ(res[n]['fuzz_arc'], gold)[0]:3d}   skidpad {differ(res[n]['skid'], sk)[0]:3d}"
              f"   | vs oracle(det): fuzz arc {differ(res[n]['fuzz_arc'], o['path'][arc])[0]:3d}   vs oracle(host libm): {differ(res[n]['fuzz_arc'], o_libm['path'][arc])[0]:3d}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        main()
