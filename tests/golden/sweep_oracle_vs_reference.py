#!/usr/bin/env python3
"""Wide random comparison of the oracle with the REFERENCE itself (build container only: imports /root/reference through
refharness.py).  Not a fixture generator: nothing is stored; it widens the pin of tests/golden/*.npz by fresh random frame
sets.  python tests/golden/sweep_oracle_vs_reference.py [frames_per_set]   -> category counts per set (tests/parity.py)."""
import collections
import importlib
import itertools
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent))
import make_golden  # noqa: E402
import oracle_lib  # noqa: E402
import parity  # noqa: E402

synth = importlib.import_module("ft-fsd-path-planning_amd.synth")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
total = collections.Counter()
seed = 500
for per_side, track_noise, frame_noise, colour in itertools.product((24, 64, 100), (0.1, 0.3), (0.0, 0.15, 0.3, 0.5), (True, False)):
    seed += 1
    off, cones, poses = synth.make_replay_batch(N, per_side, track_noise, seed=seed, frame_noise=frame_noise,
                                                random_pose=frame_noise > 0, color=colour)
    g = make_golden.capture(off, cones, poses)
    res = oracle_lib.plan_batch(g["offsets"], g["cones"], g["poses"])  # libm mode: what the reference's NumPy uses
    cats = collections.Counter()
    for k in range(N):
        cat, detail = parity.compare_frame(res[k], g, k)
        cats[cat] += 1
        if cat not in ("ok", "ref_undefined", "flip"):
            print("   MISMATCH", per_side, track_noise, frame_noise, colour, "frame", k, cat, str(detail)[:200], flush=True)
    total.update(cats)
    print(f"cones/side {per_side:3d} track sigma {track_noise} frame sigma {frame_noise} colour {int(colour)}: {dict(cats)}", flush=True)
print("TOTAL", dict(total))
