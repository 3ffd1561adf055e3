#!/usr/bin/env python3
"""bench.py's host -> host streaming leg (100 different 4096-frame batches) at 10 / 14 / 16 / 20 pass slots: frames/s, one batch at a
time, equality with a serial plan_batch.  (Round 4: 4.43 / 4.31 / 4.54 / 4.47 M — the stream does not care about the depth.)"""
import importlib, sys, json, os
sys.path.insert(0, os.getcwd())
import bench
pkg = importlib.import_module("ft-fsd-path-planning_amd")
pkg._capi.DEFAULT_OPTIONS.update(pkg._capi.options_from_env())  # FSDP_PACK / FSDP_PATH_MODE / ... of this tool's shell -> fsdp_set_option
ctx = pkg.Context(device=0)
for depth in (10, 14, 16, 20):
    r = bench.streaming_leg(pkg, ctx, 4096, depth, 100, 1)
    print(depth, round(r["value"]), round(r["one_batch_at_a_time_frames_per_s"]), r["last_batch_equals_serial_plan_batch"], flush=True)
