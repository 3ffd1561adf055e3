#!/usr/bin/env python3
"""How many frames of the streaming leg's batches (bench.py: seeds seed0 + 1000 + k) leave the packed path kernels for the exact route,
and what the same batches do RESIDENT (no PCIe) at the stream's depth — the right yardstick for the stream's rate."""
import importlib, json, os, sys, time
from pathlib import Path
os.environ.setdefault("GPU_MAX_HW_QUEUES", "22")
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
pkg = importlib.import_module("ft-fsd-path-planning_amd")
ctx = pkg.Context(device=0, options={"plan_chunks": 1})
rows = []
for k in range(0, 100, 9):
    off, cones, poses = pkg.synth.make_replay_batch(4096, 64, 0.15, seed=1 + 1000 + k, color=True)
    ctx.set_overlap(1)
    res = ctx.plan_batch(off, cones, poses)
    nk, _, _ = ctx.debug_refit()
    ctx.set_overlap(20)
    ctx.upload(off, cones, poses)
    ctx.time_runs(20)
    t0 = time.perf_counter(); ctx.time_runs(60, collect=False); ctx.sync(); el = time.perf_counter() - t0
    rows.append({"batch": k, "frames_on_the_exact_route": int((nk < 0).sum()), "max_knots_fast": int(nk.max()), "resident_20_in_flight_frames_per_s": round(4096 * 60 / el),
                 "kernels": ctx.stage_names()})
    print(json.dumps(rows[-1]), flush=True)
print(json.dumps({"mean_resident_frames_per_s": round(float(np.mean([r["resident_20_in_flight_frames_per_s"] for r in rows]))),
                  "mean_frames_on_the_exact_route": float(np.mean([r["frames_on_the_exact_route"] for r in rows]))}))
