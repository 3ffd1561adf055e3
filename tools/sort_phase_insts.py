#!/usr/bin/env python3
"""Runs ONE sorting pass of the bench batch with the library given as argv[1] (instruction accounting builds: the sorting
stage stops after phase FSDP_SORT_STOP; results are meaningless, SQ_INSTS_VALU under rocprofv3 --pmc is the point)."""
import importlib, sys, os
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
pkg = importlib.import_module("ft-fsd-path-planning_amd")
pkg._capi.DEFAULT_OPTIONS.update(pkg._capi.options_from_env())  # FSDP_PACK / FSDP_PATH_MODE / ... of this tool's shell -> fsdp_set_option
pkg._capi.LIB_PATH = Path(sys.argv[1])
ctx = pkg.Context(device=0)
off, cones, poses = pkg.synth.make_replay_batch(4096, 64, 0.15, seed=1, color=True)
ctx.upload(off, cones, poses)
import ctypes
ctx._lib.fsdp_profile_select(ctx._h, 1) if hasattr(ctx._lib, "fsdp_profile_select") else None
for _ in range(3):
    ctx.run()
ctx.sync()
