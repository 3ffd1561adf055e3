#!/usr/bin/env python3
"""Streaming rate (different batches host -> host, fsdp_submit / fsdp_collect) over depth and batch size."""
import importlib, json, os, sys, time
from pathlib import Path
os.environ.setdefault("GPU_MAX_HW_QUEUES", "22")
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import bench
pkg = importlib.import_module("ft-fsd-path-planning_amd")
pkg._capi.DEFAULT_OPTIONS.update(pkg._capi.options_from_env())  # FSDP_PACK / FSDP_PATH_MODE / ... of this tool's shell -> fsdp_set_option
ctx = pkg.Context(device=0)
SETS = ((4096, 100), (8192, 50), (2048, 200))
DEPTHS = (1, 2, 4, 6, 10, 16, 20)
if len(sys.argv) > 1:  # python tools/stream_probe.py 4096 10,16,20
    SETS = ((int(sys.argv[1]), 409600 // int(sys.argv[1])),)
    DEPTHS = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else DEPTHS
for per_gpu, nb in SETS:
    for depth in DEPTHS:
        r = bench.streaming_leg(pkg, ctx, per_gpu, depth, nb, 1)
        print(json.dumps({"frames_per_batch": per_gpu, "depth": depth, "Mframes_s_compact_records": r["value"] / 1e6,
                          "Mframes_s_full_records": r["full_records_frames_per_s"] / 1e6,
                          "one_at_a_time_compact": r["one_batch_at_a_time_frames_per_s"] / 1e6,
                          "one_at_a_time_full": r["one_batch_at_a_time_full_records_frames_per_s"] / 1e6,
                          "one_at_a_time_pageable": r["one_batch_at_a_time_pageable_frames_per_s"] / 1e6,
                          "h2d_GBps": r["pcie_GBps"]["h2d"], "d2h_GBps": r["pcie_GBps"]["d2h"], "pcie_ceiling_GBps": r["pcie_ceiling_GBps"],
                          "reruns": r["passes_rerun_for_routes"]}), flush=True)
