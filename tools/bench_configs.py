#!/usr/bin/env python3
"""All BASELINE.json configs on one MI355X (resident batches, HIP-event kernel times).  Prints one JSON object per
config; used to fill the tables in DESIGN.md / README.md.  Config 1 (single frame, CPU reference) is the golden
"Hairpin" scenario + one synthetic frame through the batch = 1 C-ABI call (latency)."""
import importlib
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
pkg = importlib.import_module("ft-fsd-path-planning_amd")
pkg._capi.DEFAULT_OPTIONS.update(pkg._capi.options_from_env())  # FSDP_PACK / FSDP_PATH_MODE / ... of this tool's shell -> fsdp_set_option
ctx = pkg.Context(device=0)


OVERLAP = 10


def run(name, off, cones, poses, steps=10):
    ctx.set_overlap(1)
    ctx.upload(off, cones, poses)
    ctx.time_runs(2)
    tot, st = ctx.time_runs(steps)  # one pass after the other: per-launch kernel durations
    names = ctx.stage_names()
    ov = min(OVERLAP, max(2, 131072 // (len(off) - 1)))  # like bench.py: bound the intermediates of the passes in flight
    ctx.set_overlap(ov)
    ctx.time_runs(ov)
    tot2, _ = ctx.time_runs(4 * steps)  # passes in flight (how bench.py runs)
    res = ctx.download()
    ctx.set_overlap(1)
    n = len(off) - 1
    hist = {int(k): int(v) for k, v in zip(*np.unique(res["status"], return_counts=True))}
    arc = int(((res["path_fallback"] & 16) != 0).sum())
    print(json.dumps({"config": name, "frames": n, "cones_per_frame": int((off[1:] - off[:-1]).mean()),
                      "ms_per_batch_serial": round(tot / steps, 3), "frames_per_s_serial": round(n / (tot / steps) * 1e3),
                      "ms_per_batch_overlapped": round(tot2 / (4 * steps), 3), "frames_per_s_overlapped": round(n / (tot2 / (4 * steps)) * 1e3),
                      "pass_overlap": ov, "kernel_ms_serial": {k: round(v / steps, 3) for k, v in zip(names, st)},
                      "status_histogram": hist, "arc_extension_frames": arc}), flush=True)


g = np.load(ROOT / "tests" / "golden" / "scenarios.npz")
k = 12  # "Hairpin" (18 L + 19 R), SURVEY 8d config 1 substitute
o1 = np.array([0, g["offsets"][k + 1] - g["offsets"][k]], np.int32)
c1 = g["cones"][g["offsets"][k] : g["offsets"][k + 1]]
lat = []
for _ in range(300):
    t0 = time.perf_counter()
    ctx.plan_batch(o1, c1, g["poses"][k][None])
    lat.append(time.perf_counter() - t0)
print(json.dumps({"config": "cfg1: single Hairpin frame through fsdp_plan_batch (host buffers, PCIe-inclusive)",
                  "p50_us": float(np.median(lat) * 1e6), "p99_us": float(np.percentile(lat, 99) * 1e6)}))
run("cfg2: 4096 replay frames, 128 coloured cones", *pkg.synth.make_replay_batch(4096, 64, 0.15, seed=1, color=True))
run("cfg3: 4096 replay frames, 128 cones, no colour", *pkg.synth.make_replay_batch(4096, 64, 0.15, seed=1, color=False))
run("cfg4 (per-GPU shard of the 65 536): 8192 frames x 200 cones, sigma 0.1", *pkg.synth.make_config4_shard(0, 8192, 100, 0.1, seed=7), steps=5)
run("cfg4-robustness: 2048 frames x 200 cones, sigma 0.3, no colour", *pkg.synth.make_replay_batch(2048, 100, 0.0, seed=8, frame_noise=0.3, random_pose=True, color=False), steps=5)
run("scale: 32768 replay frames, 128 coloured cones", *pkg.synth.make_replay_batch(32768, 64, 0.15, seed=1, color=True), steps=3)
