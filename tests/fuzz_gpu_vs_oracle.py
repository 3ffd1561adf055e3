#!/usr/bin/env python3
"""Wide random comparison of the HIP path with the oracle (det-math mode), every path-stage instantiation — a manual
sweep for the GPU box (not collected by pytest):  python tests/fuzz_gpu_vs_oracle.py [frames_per_set]
Prints one line per (set, instantiation) with the number of frames that differ in status, indices or path."""
import importlib
import itertools
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib  # noqa: E402

pkg = importlib.import_module("ft-fsd-path-planning_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
# fsdp_set_option pins the path-stage instantiation; "default" = the library's own choice (a blocking call of 2048 frames or
# more is pipelined in chunks, include/fsdp.h)
MODES = {"mono64": {"path_mode": 1}, "split16": {"path_mode": 2, "pack": 1}, "packed8": {"path_mode": 2, "pack": 2, "fit_g": 8},
         "packed8_fit4": {"path_mode": 2, "pack": 2, "fit_g": 4}, "default": {}}
ctxs = {name: pkg.Context(device=0, options=opt) for name, opt in MODES.items()}
bad_total = 0
seed = 100
for per_side, track_noise, frame_noise, colour in itertools.product((24, 64, 100), (0.1, 0.3), (0.0, 0.15, 0.3, 0.5), (True, False)):
    seed += 1
    off, cones, poses = pkg.synth.make_replay_batch(N, per_side, track_noise, seed=seed, frame_noise=frame_noise,
                                                    random_pose=frame_noise > 0, color=colour)
    with oracle_lib.math_mode(1):
        ref = oracle_lib.plan_batch(off, cones, poses, n_threads=os.cpu_count() or 1)
    ok = ref["status"] == 0
    for name, ctx in ctxs.items():
        res = ctx.plan_batch(off, cones, poses)
        bad = res["status"] != ref["status"]
        for f in ("n_left", "n_right", "left_idx", "right_idx", "n_left_v", "n_right_v", "l2r", "r2l", "left_v", "right_v", "path_fallback"):
            d = res[f] != ref[f]
            bad |= ok & (d.reshape(len(d), -1).any(axis=1))
        err = np.abs(res["path"] - ref["path"]).reshape(len(ok), -1)
        err = np.where(np.isnan(err), 0.0, err).max(axis=1)
        bad |= ok & (err > 1e-9)
        bad_total += int(bad.sum())
        print(f"cones/side {per_side:3d} track sigma {track_noise} frame sigma {frame_noise} colour {int(colour)} {name:12s}: "
              f"{int(bad.sum())} of {len(ok)} frames differ (status ok {int(ok.sum())}, arc {int(((ref['path_fallback'] & 16) != 0).sum())})"
              + (f"  first: {np.nonzero(bad)[0][:5].tolist()}" if bad.any() else ""), flush=True)
print("TOTAL differing frames:", bad_total)
