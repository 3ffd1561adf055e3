#!/usr/bin/env python3
"""Where a stream of batches spends its time on the host: duration of every fsdp_submit / fsdp_collect call."""
import importlib, json, os, sys, time
from pathlib import Path
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
pkg = importlib.import_module("ft-fsd-path-planning_amd")
pkg._capi.DEFAULT_OPTIONS.update(pkg._capi.options_from_env())  # FSDP_PACK / FSDP_PATH_MODE / ... of this tool's shell -> fsdp_set_option
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 10
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 60
per = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
ctx = pkg.Context(device=0)
batches = []
for k in range(nb):
    off, cones, poses = pkg.synth.make_replay_batch(per, 64, 0.15, seed=2000 + k, color=True)
    batches.append((pkg.pinned_copy(off, np.int32), pkg.pinned_copy(cones, np.float64), pkg.pinned_copy(poses, np.float64)))
outs = [pkg.pinned_empty(per, pkg.RESULT_DTYPE) for _ in range(nb)]
ctx.set_overlap(depth)
for k in range(min(depth, nb)):
    ctx.collect(ctx.submit(*batches[k], out=outs[k]))
ctx.collect(ctx.submit(*batches[0], out=outs[0]))
ev = []
inflight = []
T0 = time.perf_counter()
for k in range(nb):
    if len(inflight) == ctx.ticket_capacity:
        a = time.perf_counter(); ctx.collect(inflight.pop(0)); b = time.perf_counter()
        ev.append(("collect", a - T0, b - a))
    a = time.perf_counter(); inflight.append(ctx.submit(*batches[k], out=outs[k])); b = time.perf_counter()
    ev.append(("submit", a - T0, b - a))
for t in inflight:
    a = time.perf_counter(); ctx.collect(t); b = time.perf_counter()
    ev.append(("collect", a - T0, b - a))
el = time.perf_counter() - T0
sub = np.array([e[2] for e in ev if e[0] == "submit"]); col = np.array([e[2] for e in ev if e[0] == "collect"])
print(json.dumps({"depth": depth, "batches": nb, "frames": per, "Mframes_s": per * nb / el / 1e6, "total_ms": el * 1e3,
                  "submit_ms": {"mean": sub.mean() * 1e3, "p50": float(np.median(sub)) * 1e3, "max": sub.max() * 1e3, "sum": sub.sum() * 1e3},
                  "collect_ms": {"mean": col.mean() * 1e3, "p50": float(np.median(col)) * 1e3, "max": col.max() * 1e3, "sum": col.sum() * 1e3}}))
if os.environ.get("DUMP"):
    for e in ev:
        print("%s %.3f %.3f" % (e[0], e[1] * 1e3, e[2] * 1e3))
