#!/usr/bin/env python3
"""Thread-scaling of the CPU oracle (the bench's cpu_baseline leg) on this host: frames/s for 1..N threads, plus the
CPU resources the container really has (affinity mask, cgroup quota).  Diagnostic for bench.py's `cores` figure."""
import importlib
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent  # (lives under tests/: it times the oracle)
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib  # noqa: E402

pkg = importlib.import_module("ft-fsd-path-planning_amd")
off, cones, poses = pkg.synth.make_replay_batch(4096, 64, 0.15, seed=1, color=True)
info = {"cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(p):
        info[p] = open(p).read().strip()
print(json.dumps(info))
oracle_lib.plan_batch(off[:257], cones[: off[256]], poses[:256], n_threads=8)
nt = 1
while nt <= (os.cpu_count() or 1):
    n = min(4096, max(256, 64 * nt))
    t = time.perf_counter()
    oracle_lib.plan_batch(off[: n + 1], cones[: off[n]], poses[:n], n_threads=nt)
    el = time.perf_counter() - t
    print(json.dumps({"threads": nt, "frames": n, "frames_per_s": n / el, "per_thread": n / el / nt}))
    nt *= 2
