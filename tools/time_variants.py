#!/usr/bin/env python3
"""Time kernel-stage durations for alternative builds of the library (lib/variants/*.so) on cfg2 / cfg3 / cfg4 batches."""
import importlib, json, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
pkg = importlib.import_module("ft-fsd-path-planning_amd")
pkg._capi.DEFAULT_OPTIONS.update(pkg._capi.options_from_env())  # FSDP_PACK / FSDP_PATH_MODE / ... of this tool's shell -> fsdp_set_option
batches = {"cfg2": pkg.synth.make_replay_batch(4096, 64, 0.15, seed=1, color=True),
           "cfg3": pkg.synth.make_replay_batch(4096, 64, 0.15, seed=1, color=False),
           "cfg4": pkg.synth.make_replay_batch(8192, 100, 0.0, seed=7, frame_noise=0.1, random_pose=True)}
for so in sorted((ROOT / "ft-fsd-path-planning_amd" / "lib" / "variants").glob("*.so")) + [ROOT / "ft-fsd-path-planning_amd" / "lib" / "libfsdp_hip.so"]:
    import subprocess
    code = f"""
import importlib, sys, json
sys.path.insert(0, {str(ROOT)!r})
pkg = importlib.import_module('ft-fsd-path-planning_amd')
from pathlib import Path
pkg._capi.LIB_PATH = Path({str(so)!r})
import numpy as np
ctx = pkg.Context(device=0)
out = {{}}
for name, args in [('cfg2', dict(n=4096, k=64, tn=0.15, seed=1, color=True)), ('cfg3', dict(n=4096, k=64, tn=0.15, seed=1, color=False)), ('cfg4', dict(n=8192, k=100, tn=0.0, seed=7, color=True, fn=0.1, rp=True))]:
    off, cones, poses = pkg.synth.make_replay_batch(args['n'], args['k'], args['tn'], seed=args['seed'], color=args['color'], frame_noise=args.get('fn', 0.0), random_pose=args.get('rp', False))
    ctx.set_overlap(1); ctx.upload(off, cones, poses); ctx.run(); ctx.sync()
    tot, st = ctx.time_runs(5)
    ctx.set_overlap(2); ctx.time_runs(4); tot2, st2 = ctx.time_runs(12)
    out[name] = [round(x / 5, 3) for x in st] + ['serial step', round(tot / 5, 3), 'overlapped step', round(tot2 / 12, 3)]
print(json.dumps({{'so': Path({str(so)!r}).name, 'sort/match/path ms': out}}))
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-400:])
