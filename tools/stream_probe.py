#!/usr/bin/env python3
"""Streaming rate (different batches host -> host, fsdp_submit / fsdp_collect) over depth and batch size."""
import importlib, json, os, sys, time
from pathlib import Path
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import bench
pkg = importlib.import_module("ft-fsd-path-planning_amd")
pkg._capi.DEFAULT_OPTIONS.update(pkg._capi.options_from_env())  # FSDP_PACK / FSDP_PATH_MODE / ... of this tool's shell -> fsdp_set_option
ctx = pkg.Context(device=0)
for per_gpu, nb in ((4096, 40), (8192, 20), (2048, 80)):
    for depth in (1, 2, 4, 6, 10, 16):
        r = bench.streaming_leg(pkg, ctx, per_gpu, depth, nb, 1)
        print(json.dumps({"frames_per_batch": per_gpu, "depth": depth, "Mframes_s_compact_records": r["value"] / 1e6,
                          "Mframes_s_full_records": r["full_records_frames_per_s"] / 1e6,
                          "one_at_a_time_compact": r["one_batch_at_a_time_frames_per_s"] / 1e6,
                          "one_at_a_time_full": r["one_batch_at_a_time_full_records_frames_per_s"] / 1e6,
                          "one_at_a_time_pageable": r["one_batch_at_a_time_pageable_frames_per_s"] / 1e6,
                          "h2d_GBps": r["pcie_GBps"]["h2d"], "d2h_GBps": r["pcie_GBps"]["d2h"], "pcie_ceiling_GBps": r["pcie_ceiling_GBps"],
                          "reruns": r["passes_rerun_for_routes"]}), flush=True)
