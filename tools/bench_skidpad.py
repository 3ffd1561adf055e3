#!/usr/bin/env python3
"""BASELINE config 5: 1024 skidpad planner instances replaying demo/skidpad.json (golden copy) with rigidly perturbed
starts on one MI355X.  Reports frames/s over the whole stateful sequence, relocalization success, kernel time."""
import importlib
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import skidpad_support as sk  # noqa: E402

pkg = importlib.import_module("ft-fsd-path-planning_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
g = sk.load_sequence(ROOT / "tests" / "golden")
tf = sk.perturbed_instances(g, n)
T = len(g["poses"])
batches = [sk.batch_for_step(g, t, tf) for t in range(T)]
batch = pkg.SkidpadBatch(n, device=0)
for t in range(3):  # warm-up on a throw-away state (reference demo does the same, json_demo.py:89-94)
    batch.step(*batches[t])
batch.reset()
t0 = time.perf_counter()
status = np.zeros(n, np.int64)
for t in range(T):
    res, info = batch.step(*batches[t])
    status += res["status"] != 0
el = time.perf_counter() - t0
kms = batch.time_path(10) / 10
print(json.dumps({"config": "BASELINE configs[4]: skidpad, batch=%d perturbed starts x %d frames" % (n, T),
                  "frames_per_s_incl_pcie": n * T / el, "seconds": el, "relocalized": int(info["relocalized"].sum()),
                  "frames_with_nonzero_status": int(status.sum()), "skid_path_kernel_ms_per_step": kms,
                  "frames_per_s_kernel_only": n / (kms * 1e-3)}))
