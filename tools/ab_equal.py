#!/usr/bin/env python3
"""Do the experiment builds under lib/variants/ return the product library's bits?  The bench batch (4096 x 128 coloured), a
noisy colourless batch (exact-route frames) and the 398 fuzz frames, every result field compared byte for byte."""
import importlib, json, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
libs = sorted((ROOT / "ft-fsd-path-planning_amd" / "lib" / "variants").glob("*.so")) + [ROOT / "ft-fsd-path-planning_amd" / "lib" / "libfsdp_hip.so"]
code = """
import importlib, sys, hashlib, json, os
import numpy as np
sys.path.insert(0, %r)
from pathlib import Path
pkg = importlib.import_module('ft-fsd-path-planning_amd')
pkg._capi.LIB_PATH = Path(sys.argv[1])
ctx = pkg.Context(device=0)
out = {}
sets = {'bench': pkg.synth.make_replay_batch(4096, 64, 0.15, seed=1, color=True),
        'noisy': pkg.synth.make_replay_batch(2048, 100, 0.0, seed=8, frame_noise=0.3, random_pose=True, color=False, lateral_noise=0.5, heading_noise=0.2)}
g = np.load(Path(%r) / 'tests' / 'golden' / 'fuzz.npz')
sets['fuzz'] = (g['offsets'], g['cones'], g['poses'])
for name, (off, cones, poses) in sets.items():
    r = ctx.plan_batch(off, cones, poses)
    h = hashlib.sha256()
    for k in r.dtype.names:
        h.update(np.ascontiguousarray(r[k]).tobytes())
    out[name] = h.hexdigest()[:16] + ' st0=%%d' %% int((r['status'] == 0).sum())
print(json.dumps(out))
""" % (str(ROOT), str(ROOT))
ref = None
for so in libs:
    r = subprocess.run([sys.executable, "-c", code, str(so)], capture_output=True, text=True)
    line = r.stdout.strip() or r.stderr[-300:]
    if ref is None:
        ref = line
    print(f"{so.name:50s} {'SAME ' if line == ref else 'DIFF '} {line}", flush=True)
