#!/usr/bin/env python3
"""A/B of library builds (lib/variants/*.so + the product) on the workloads that live on the exact route (path_retry_kernel):
config 4r (2048 frames x 200 cones, sigma 0.3, colourless: ~3 % of the frames) and a global-path batch (every frame: the 128 m
slices need more knots than the packed kernels keep).  One pass at a time and ten in flight; result hashes must agree."""
import importlib, json, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
libs = sorted((ROOT / "ft-fsd-path-planning_amd" / "lib" / "variants").glob("*.so")) + [ROOT / "ft-fsd-path-planning_amd" / "lib" / "libfsdp_hip.so"]
code = """
import importlib, sys, json, time, hashlib
import numpy as np
sys.path.insert(0, %r)
from pathlib import Path
pkg = importlib.import_module('ft-fsd-path-planning_amd')
pkg._capi.LIB_PATH = Path(sys.argv[1])
out = {}
def run(name, ctx, off, cones, poses, steps):
    ctx.set_overlap(1); ctx.upload(off, cones, poses); ctx.time_runs(2)
    tot, st = ctx.time_runs(steps)
    names = ctx.stage_names()
    ctx.set_overlap(10); ctx.time_runs(10)
    tot2, _ = ctx.time_runs(4 * steps)
    deep = {}
    for ov in (16, 24, 32):
        ctx.set_overlap(ov); ctx.time_runs(ov)
        t3, _ = ctx.time_runs(3 * ov)
        deep[ov] = round((len(off) - 1) / (t3 / (3 * ov)) * 1e3)
    ctx.set_overlap(10)
    res = ctx.download()
    h = hashlib.sha256()
    for k in res.dtype.names: h.update(np.ascontiguousarray(res[k]).tobytes())
    n = len(off) - 1
    out[name] = {'serial_fps': round(n / (tot / steps) * 1e3), 'overlapped_fps': round(n / (tot2 / (4 * steps)) * 1e3),
                 'retry_ms_serial': round(dict(zip(names, st)).get('path_retry_kernel', 0) / steps, 3), 'hash': h.hexdigest()[:12], 'overlapped_fps_by_depth': deep}
ctx = pkg.Context(device=0)
run('cfg4r', ctx, *pkg.synth.make_replay_batch(2048, 100, 0.0, seed=8, frame_noise=0.3, random_pose=True, color=False), steps=5)
left, right, centre_fn = pkg.synth.closed_track(40, 33)
gp = np.array([centre_fn(s)[0] for s in np.linspace(0, 1, 600, endpoint=False)])
ctx2 = pkg.Context(device=0)
ctx2.set_global_path(gp)
off, cones, poses = pkg.synth.make_replay_batch(2048, 40, 0.15, seed=33, color=True)
run('global_path_2048', ctx2, off, cones, poses, steps=3)
print(json.dumps(out))
""" % str(ROOT)
for so in libs:
    r = subprocess.run([sys.executable, "-c", code, str(so)], capture_output=True, text=True)
    print(f"{so.name:30s}", r.stdout.strip() or r.stderr[-400:], flush=True)
