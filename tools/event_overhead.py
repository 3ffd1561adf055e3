#!/usr/bin/env python3
"""What the per-kernel HIP events of the timed region cost: 20 passes through fsdp_time_runs (events around every launch)
against 20 plain fsdp_run calls, same box, alternating.  python tools/event_overhead.py"""
import importlib, sys, time, os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module('ft-fsd-path-planning_amd')
ctx = pkg.Context(device=0)
off, cones, poses = pkg.synth.make_replay_batch(4096, 64, 0.15, seed=1, color=True)
ctx.set_overlap(10); ctx.upload(off, cones, poses)
for _ in range(20): ctx.run()
ctx.sync()
ctx.time_reserve(20)
a, b = [], []
for rep in range(7):
    ctx.sync(); t0 = time.perf_counter(); ctx.time_runs(20, collect=False); ctx.sync(); a.append(20 * 4096 / (time.perf_counter() - t0))
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(20): ctx.run()
    ctx.sync(); b.append(20 * 4096 / (time.perf_counter() - t0))
print("time_runs(20):", [round(r / 1e6, 3) for r in sorted(a)])
print("20 x run()   :", [round(r / 1e6, 3) for r in sorted(b)])
