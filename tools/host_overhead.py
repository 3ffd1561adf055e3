#!/usr/bin/env python3
"""Host side of a 20-pass run: wall time of fsdp_time_runs against the HIP-event time of the same passes, and the same passes
through 20 x fsdp_run + fsdp_sync (no per-kernel events).  Run on the GPU box."""
import importlib, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
pkg = importlib.import_module('ft-fsd-path-planning_amd')
ctx = pkg.Context(device=0)
off, cones, poses = pkg.synth.make_replay_batch(4096, 64, 0.15, seed=1, color=True)
ctx.set_overlap(10); ctx.upload(off, cones, poses)
for _ in range(5): ctx.run()
ctx.sync()
for rep in range(4):
    t0 = time.perf_counter()
    tot, st = ctx.time_runs(20)
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    print(f"time_runs wall {1e3*(t1-t0):.3f} ms, events total {tot:.3f} ms, sync after {1e3*(t2-t1):.3f} ms")
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20): ctx.run()
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    print(f"20 x run(): enqueue {1e3*(t1-t0):.3f} ms, until sync {1e3*(t2-t0):.3f} ms -> {20*4096/(t2-t0):.0f} frames/s")
