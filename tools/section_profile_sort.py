#!/usr/bin/env python3
"""Per-section cycle accounting of the sorting kernel (profiling build, -DFSDP_PROFILE; not the product .so).
Run on the GPU box:  python tools/section_profile_sort.py [n_frames] [--no-colour]"""
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
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-sink-insts-to-avoid-spills=1", "-fPIC", "-shared",
                "-DFSDP_PROFILE", str(PKG / "csrc" / "fsdp_lib.hip"), "-o", str(so), "-ldl"], check=True, capture_output=True)
pkg = importlib.import_module("ft-fsd-path-planning_amd")
pkg._capi.DEFAULT_OPTIONS.update(pkg._capi.options_from_env())  # FSDP_PACK / FSDP_PATH_MODE / ... of this tool's shell -> fsdp_set_option
pkg._capi.LIB_PATH = so
ctx = pkg.Context(device=0)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(args[0]) if args else 4096
colour = "--no-colour" not in sys.argv
off, cones, poses = pkg.synth.make_replay_batch(N, 64, 0.15, seed=1, color=colour)
ctx.upload(off, cones, poses)
ctx.run()
ctx.sync()
out = np.zeros((N, 32), np.int64)
assert ctx._lib.fsdp_profile_select(ctx._h, 1) == 0
assert ctx._lib.fsdp_profile_path(ctx._h, ctypes.c_void_p(out.ctypes.data)) == 0
names = {1: "S4 start cones", 2: "S5 kNN adjacency + reach", 3: "S8 DFS (both sides at once)", 4: "S10 post filters", 5: "S12 cones on either side",
         6: "S11 costs + pick"}
m = out.mean(axis=0)
tot = sum(m[k] for k in names)
print(f"{N} frames, colour={colour}; both sides summed; mean cycles per frame in sort_one_side: {tot:.0f}")
for k in sorted(names):
    print(f"{names[k]:<28}{m[k]:>14.0f}{100 * m[k] / tot:>9.1f}%")
