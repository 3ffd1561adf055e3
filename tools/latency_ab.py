#!/usr/bin/env python3
"""A/B of library builds on the latency figures: p50 of the single-frame call and one serial pass of the bench batch, per kernel.
python tools/latency_ab.py lib1.so lib2.so ...   (each in a child process)"""
import subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
code = r'''
import sys, importlib, time, json, numpy as np
from pathlib import Path
sys.path.insert(0, %r)
pkg = importlib.import_module("ft-fsd-path-planning_amd")
pkg._capi.DEFAULT_OPTIONS.update(pkg._capi.options_from_env())  # FSDP_PACK / FSDP_PATH_MODE / ... of this tool's shell -> fsdp_set_option
pkg._capi.LIB_PATH = Path(sys.argv[1])
ctx = pkg.Context(device=0)
off, cones, poses = pkg.synth.make_replay_batch(4096, 64, 0.15, seed=1, color=True)
o1, c1, p1 = off[:2], cones[:off[1]], poses[:1]
lat = []
for _ in range(400):
    t = time.perf_counter(); ctx.plan_batch(o1, c1, p1); lat.append(time.perf_counter() - t)
ctx.upload(off, cones, poses); ctx.time_runs(3); tot, st = ctx.time_runs(10)
print(json.dumps({"p50_single_frame_us": round(float(np.median(lat[100:]) * 1e6), 1), "serial_pass_ms": round(tot / 10, 3),
                  "kernel_ms": dict(zip(ctx.stage_names(), [round(x / 10, 3) for x in st]))}))
''' % str(ROOT)
for so in sys.argv[1:]:
    r = subprocess.run([sys.executable, "-c", code, so], capture_output=True, text=True)
    print(f"{Path(so).name:28s}", r.stdout.strip() or r.stderr[-300:], flush=True)
