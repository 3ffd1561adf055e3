#!/usr/bin/env python3
"""ms per 4096-frame pass with 1 .. 8 passes in flight (fsdp_set_overlap) for a library build (path of a .so, or
`default`).  Usage on the GPU box: python tools/overlap_depths.py default"""
import importlib, sys, json
from pathlib import Path
import os
os.environ.setdefault('GPU_MAX_HW_QUEUES', '12')
ROOT = Path(__file__).resolve().parent.parent; sys.path.insert(0, str(ROOT))
pkg = importlib.import_module('ft-fsd-path-planning_amd')
so = sys.argv[1]
if so != 'default':
    pkg._capi.LIB_PATH = Path(so)
ctx = pkg.Context(device=0)
off, cones, poses = pkg.synth.make_replay_batch(4096, 64, 0.15, seed=1, color=True)
res = {}
for d in (1, 2, 3, 4, 5, 6, 8):
    ctx.set_overlap(d); ctx.upload(off, cones, poses); ctx.time_runs(2 * d)
    tot, st = ctx.time_runs(48)
    res[d] = {"ms_per_pass": round(tot / 48, 3), "frames_per_s": round(4096 / (tot / 48) * 1e3)}
print(so.split('/')[-1], res)
