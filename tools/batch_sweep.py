#!/usr/bin/env python3
"""Kernel times vs batch size (resident batches, HIP events): shows the single-wave latency floor and where the
chip fills up.  One JSON line per batch size."""
import importlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
pkg = importlib.import_module("ft-fsd-path-planning_amd")
pkg._capi.DEFAULT_OPTIONS.update(pkg._capi.options_from_env())  # FSDP_PACK / FSDP_PATH_MODE / ... of this tool's shell -> fsdp_set_option
import os
if os.environ.get("AB_LIB"):  # an experiment build (tools/build_variant.sh)
    pkg._capi.LIB_PATH = Path(os.environ["AB_LIB"])
ctx = pkg.Context(device=0)
sizes = [int(a) for a in sys.argv[1:]] or [64, 256, 512, 1024, 1536, 2048, 2560, 3072, 4096, 5120, 6144, 8192, 16384]
off, cones, poses = pkg.synth.make_replay_batch(max(sizes), 64, 0.15, seed=1, color=True)
if os.environ.get("AB_REPEAT"):  # a batch whose frames repeat with this period (experiments with aliased scratch arenas)
    import numpy as np
    per = int(os.environ["AB_REPEAT"])
    n = max(sizes)
    counts = np.diff(off)[:per]
    reps = -(-n // per)
    cones = np.tile(cones[: off[per]], (reps, 1))
    poses = np.tile(poses[:per], (reps, 1))[:n]
    off = np.concatenate([[0], np.cumsum(np.tile(counts, reps))]).astype(np.int32)[: n + 1]
    cones = cones[: off[n]]
for n in sizes:
    ctx.upload(off[: n + 1], cones[: off[n]], poses[:n])
    ctx.time_runs(2)
    tot, st = ctx.time_runs(10)
    names = ctx.stage_names()
    path = sum(st[2:]) / 10
    print(json.dumps({"frames": n, "ms": round(tot / 10, 3), "kernel_ms": {k: round(v / 10, 3) for k, v in zip(names, st)},
                      "path_stage_us_per_frame": round(path / n * 1e3, 3), "frames_per_s": round(n / (tot / 10) * 1e3)}), flush=True)
