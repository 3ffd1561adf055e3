#!/usr/bin/env python3
"""The wide build (libfsdp_hip_wide.so: -DFSDP_WIDE_SHAPES, include/fsdp.h) measured next to the standard one: the bench batch
(4096 frames x 128 cones, coloured / colourless) with the DEFAULT parameters on both builds — what the larger shapes cost by
themselves — and with parameters only the wide build takes (max_n_neighbors 8, max_length 16, horizon 64).  One JSON object per
line: one pass at a time (HIP-event kernel times) and ten passes in flight."""
import importlib
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
pkg = importlib.import_module("ft-fsd-path-planning_amd")
pkg._capi.DEFAULT_OPTIONS.update(pkg._capi.options_from_env())  # FSDP_PACK / FSDP_PATH_MODE / ... of this tool's shell -> fsdp_set_option


def run(name, ctx, off, cones, poses, steps=10, ov=10):
    ctx.set_overlap(1)
    ctx.upload(off, cones, poses)
    ctx.time_runs(2)
    tot, st = ctx.time_runs(steps)
    names = ctx.stage_names()
    ctx.set_overlap(ov)
    ctx.time_runs(ov)
    tot2, _ = ctx.time_runs(4 * steps)
    res = ctx.download()
    ctx.set_overlap(1)
    n = len(off) - 1
    print(json.dumps({"case": name, "build": ctx.shapes.name, "result_bytes": ctx.result_dtype.itemsize, "frames": n,
                      "frames_per_s_serial": round(n / (tot / steps) * 1e3), "frames_per_s_10_in_flight": round(n / (tot2 / (4 * steps)) * 1e3),
                      "kernel_ms_serial": {k: round(v / steps, 3) for k, v in zip(names, st)},
                      "ok_share": round(float((res["status"] == 0).mean()), 4),
                      "longest_side": int(max(res["n_left"].max(), res["n_right"].max()))}), flush=True)


for color in (True, False):
    batch = pkg.synth.make_replay_batch(4096, 64, 0.15, seed=1, color=color)
    tag = "coloured" if color else "colourless"
    std = pkg.Context(device=0)
    run(f"{tag}, default parameters", std, *batch)
    std.close()
    wide_default = pkg.Context(device=0, shapes=pkg.WIDE)
    run(f"{tag}, default parameters", wide_default, *batch)
    wide_default.close()
    wide = pkg.Context(device=0, params=dict(max_n_neighbors=8, max_length=16, mpc_prediction_horizon=64))
    run(f"{tag}, max_n_neighbors 8 / max_length 16 / horizon 64", wide, *batch)
    wide.close()
