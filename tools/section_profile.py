#!/usr/bin/env python3
"""Per-section cycle accounting of the path kernel (profiling build, -DFSDP_PROFILE; not the product .so).
Run on the GPU box:  python tools/section_profile.py  -> prints mean cycles per frame per section."""
import ctypes
import importlib
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
PKG = ROOT / "ft-fsd-path-planning_amd"
so = ROOT / "gpurun_out" / "libfsdp_prof.so"
so.parent.mkdir(exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                "-DFSDP_PROFILE", str(PKG / "csrc" / "fsdp_lib.hip"), "-o", str(so)], check=True, capture_output=True)
pkg = importlib.import_module("ft-fsd-path-planning_amd")
pkg._capi.LIB_PATH = so
ctx = pkg.Context(device=0)
off, cones, poses = pkg.synth.make_replay_batch(4096, 64, 0.15, seed=1, color=True)
ctx.upload(off, cones, poses)
ctx.run()
ctx.sync()
out = np.zeros((4096, 32), np.int64)
rc = ctx._lib.fsdp_profile_path(ctx._h, ctypes.c_void_p(out.ctypes.data))
assert rc == 0
names = {0: "whole kernel", 1: "fit#1 (incl. parameter)", 4: "fit#2", 5: "eval#2 + cut", 7: "fit#3", 8: "eval#3", 9: "curvature windows",
         18: "filter + sample", 19: "build_parameter (all fits)", 10: "fit: basis prep (lanes)", 11: "fit: Givens pipeline", 12: "fit: fp serial sum",
         13: "fit: back substitution", 14: "fit: residual pass", 15: "fit: fpknot", 16: "fit: part-2 Givens+back", 17: "fit: f(p) pass"}
m = out.mean(axis=0)
print(f"{'section':<28}{'mean cycles/frame':>20}{'% of kernel':>14}")
for k in sorted(names):
    print(f"{names[k]:<28}{m[k]:>20.0f}{100 * m[k] / m[0]:>13.1f}%")
