#!/usr/bin/env python3
"""Decision margins of the sorting stage: how close does any discrete decision that hangs on a libm
value (atan2 / acos against a threshold, the arg-min over costs) ever come to its tie?  The kernels take those values from
the device's libm (<= 1 ulp: ~4e-16 absolute on an angle), the oracle from glibc; a decision can only differ between them
when its margin is of that order.  CPU only (the oracle, single thread); sets = the committed goldens + fresh synthetic
frame sets of the fuzz sweep's classes.   python tools/sort_margins.py > profiles/r03_sort_decision_margins.txt"""
import ctypes, importlib, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib
pkg = importlib.import_module("ft-fsd-path-planning_amd")
pkg._capi.DEFAULT_OPTIONS.update(pkg._capi.options_from_env())  # FSDP_PACK / FSDP_PATH_MODE / ... of this tool's shell -> fsdp_set_option
NAMES = ["start-cone bearing (sign, pi/10, 4pi/5)", "second cone on the vehicle's side (sign, 5 deg)", "|turn| vs absolute threshold (65 deg)",
         "turn vs directional threshold (40 deg)", "sign flip of consecutive turns (signs, 1.3 rad)", "acos thresholds, in the cosine (150/90/30 deg)",
         "wrong-direction cost terms (sign, 40 deg)", "cost arg-min: relative gap to the runner-up", "combination of the sides (turn signs)"]
L = oracle_lib.lib()
L.fsdo_margins_enable.argtypes = [ctypes.c_int]
sets = []
for name in ("scenarios", "cfg2_color", "cfg3_nocolor", "cfg4_200cones", "cfg4_noisy_nocolor", "fuzz", "lattice"):
    g = np.load(ROOT / "tests" / "golden" / f"{name}.npz")
    sets.append((f"golden {name}", g["offsets"], g["cones"], g["poses"]))
k = 0
for per_side in (24, 64, 100):
    for color in (True, False):
        for tn, fn in ((0.15, 0.0), (0.3, 0.0), (0.0, 0.1), (0.0, 0.3)):
            k += 1
            sets.append((f"synthetic {per_side}/side colour={color} track noise {tn} frame noise {fn}",
                         *pkg.synth.make_replay_batch(1024, per_side, tn, seed=5000 + k, color=color, frame_noise=fn, random_pose=fn > 0)))
L.fsdo_margins_enable(1)
frames = 0
for name, off, cones, poses in sets:
    oracle_lib.plan_batch(off, cones, poses, n_threads=1)
    frames += len(off) - 1
out = np.zeros(54)
L.fsdo_margins_get(out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
L.fsdo_margins_enable(0)
print(f"{frames} frames in {len(sets)} sets (7 golden sets + 24 fresh synthetic sets of 1024 frames)")
print(f"{'decision class':58s} {'decisions':>11s} {'exactly 0':>10s} {'min margin > 0':>15s} {'< 1e-6':>8s} {'< 1e-9':>8s} {'< 1e-12':>8s}")
for i, n in enumerate(NAMES):
    m, cnt, a, b, c, z = out[6 * i: 6 * i + 6]
    print(f"{n:58s} {int(cnt):11d} {int(z):10d} {m:15.3e} {int(a):8d} {int(b):8d} {int(c):8d}")
print("'exactly 0': value == threshold bit for bit — both angles come from identical operands (straight and lattice tracks of the demo")
print("scenarios / lattice.npz), so the difference is 0.0 under any libm; the margins listed are those of all other decisions")
print("libm error on these values: <= 1 ulp, i.e. <= 4.4e-16 absolute on an angle, <= 1.1e-16 on a cosine, ~1e-15 relative on a cost")
