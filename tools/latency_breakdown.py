#!/usr/bin/env python3
"""Single-frame latency (BASELINE metric, second half): where the time of one fsdp_plan_batch(n_frames = 1) call goes.
Wall-clock p50 of the whole call (host buffers, PCIe-inclusive), HIP-event durations of every kernel of the pass, and the
per-section cycle profile of the one-frame path kernel (profiling build).  Run on the GPU box."""
import importlib
import json
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
pkg = importlib.import_module("ft-fsd-path-planning_amd")
pkg._capi.DEFAULT_OPTIONS.update(pkg._capi.options_from_env())  # FSDP_PACK / FSDP_PATH_MODE / ... of this tool's shell -> fsdp_set_option
ctx = pkg.Context(device=0)
off, cones, poses = pkg.synth.make_replay_batch(4096, 64, 0.15, seed=1, color=True)
out = {}
for name, k in (("frame 0", 0), ("frame 2048", 2048)):
    o1 = np.array([0, off[k + 1] - off[k]], np.int32)
    c1, p1 = cones[off[k] : off[k + 1]], poses[k : k + 1]
    lat = []
    for _ in range(400):
        t0 = time.perf_counter()
        ctx.plan_batch(o1, c1, p1)
        lat.append(time.perf_counter() - t0)
    ctx.upload(o1, c1, p1)
    ctx.time_runs(20)
    tot, st = ctx.time_runs(100)
    names = ctx.stage_names()
    out[name] = {"wall_p50_us": float(np.median(lat[100:]) * 1e6), "wall_p10_us": float(np.percentile(lat[100:], 10) * 1e6),
                 "kernels_us": {n: round(v / 100 * 1e3, 1) for n, v in zip(names, st)}, "kernels_sum_us": round(sum(st) / 100 * 1e3, 1),
                 "pass_us_events": round(tot / 100 * 1e3, 1)}
# (the CPU figure to hold against this — the oracle on one core — is in the bench line: cpu_baseline.single_thread_us_per_frame)
print(json.dumps(out, indent=1))
print("== section profile of the one-frame path kernel (64 lanes per frame), 256 frames ==")
sys.stdout.flush()
subprocess.run([sys.executable, str(ROOT / "tools" / "section_profile.py"), "256"])
