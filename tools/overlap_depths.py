#!/usr/bin/env python3
"""ms per 4096-frame pass with 1 .. 16 passes in flight (steady state: 48 passes; and a 20-pass run as the driver times it) (fsdp_set_overlap) for a library build (path of a .so, or
`default`).  Usage on the GPU box: python tools/overlap_depths.py default"""
import importlib, sys, json
from pathlib import Path
import os
os.environ.setdefault('GPU_MAX_HW_QUEUES', '20')
ROOT = Path(__file__).resolve().parent.parent; sys.path.insert(0, str(ROOT))
pkg = importlib.import_module('ft-fsd-path-planning_amd')
so = sys.argv[1]
if so != 'default':
    pkg._capi.LIB_PATH = Path(so)
ctx = pkg.Context(device=0)
off, cones, poses = pkg.synth.make_replay_batch(4096, 64, 0.15, seed=1, color=True)
res = {}
for d in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16):
    ctx.set_overlap(d); ctx.upload(off, cones, poses); ctx.time_runs(2 * d)
    tot, st = ctx.time_runs(96)
    tot20 = min(ctx.time_runs(20)[0] for _ in range(3))
    res[d] = {"ms_per_pass": round(tot / 96, 3), "frames_per_s": round(4096 / (tot / 96) * 1e3),
              "frames_per_s_20_pass_run": round(4096 * 20 / tot20 * 1e3)}
print(so.split('/')[-1], res)
