#!/usr/bin/env python3
"""A/B timing of library builds in ONE gpurun call (boxes differ by +-10 %): every *.so under
ft-fsd-path-planning_amd/lib/variants/ plus the product library, wall-clock frames/s of the bench workload
(4096 x 128 coloured cones, AB_OVERLAPS passes in flight; median and best of seven 96-pass runs), two interleaved rounds, env per variant from
its file name: name__KEY=VAL__KEY=VAL.so"""
import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
libs = sorted((ROOT / "ft-fsd-path-planning_amd" / "lib" / "variants").glob("*.so")) + [ROOT / "ft-fsd-path-planning_amd" / "lib" / "libfsdp_hip.so"]
extra_env = [e for e in sys.argv[1:] if "=" in e]
code = """
import importlib, sys, json, time, os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, %r)
from pathlib import Path
pkg = importlib.import_module('ft-fsd-path-planning_amd')
pkg._capi.LIB_PATH = Path(sys.argv[1])
ctx = pkg.Context(device=0)
off, cones, poses = pkg.synth.make_replay_batch(4096, 64, 0.15, seed=1, color=True)
out = {}
for ov in [int(x) for x in os.environ.get("AB_OVERLAPS", "4,1").split(",")]:
    ctx.set_overlap(ov); ctx.upload(off, cones, poses)
    for _ in range(4): ctx.run()
    ctx.sync()
    rates = []
    for rep in range(7):
        t0 = time.perf_counter()
        for _ in range(96): ctx.run()
        ctx.sync()
        rates.append(96 * 4096 / (time.perf_counter() - t0))
    out['overlap%%d' %% ov] = [round(sorted(rates)[3]), round(max(rates))]  # median, best
print(json.dumps(out))
""" % str(ROOT)
for rnd in range(2):
    for so in libs:
        env = dict(os.environ)
        for kv in so.stem.split("__")[1:] + extra_env:
            k, v = kv.split("=")
            env[k] = v
        r = subprocess.run([sys.executable, "-c", code, str(so)], capture_output=True, text=True, env=env)
        print(f"{so.name:60s}", r.stdout.strip() or r.stderr[-300:], flush=True)
