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
KERNELS = {"fit": 1, "prep": 2, "finish": 3}
which = next((a[len("--kernel="):] for a in sys.argv[1:] if a.startswith("--kernel=")), "fit")
sys.argv = [a for a in sys.argv if not a.startswith("--kernel=")]
prebuilt = PKG / "lib" / "variants" / f"prof_{which}.so"   # tools/build_variant.sh prof_<kernel> -DFSDP_PROFILE -DFSDP_PROFILE_KERNEL=<k>
if prebuilt.exists():
    so = prebuilt
else:
  subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-sink-insts-to-avoid-spills=1", "-fPIC", "-shared",
                "-DFSDP_PROFILE", f"-DFSDP_PROFILE_KERNEL={KERNELS[which]}", str(PKG / "csrc" / "fsdp_lib.hip"), "-o", str(so), "-ldl"], check=True, capture_output=True)
import os

os.environ.setdefault("FSDP_PACK", "1")  # the packed kernels the overlapped bench runs (a single pass alone would get 16 lanes per frame)
pkg = importlib.import_module("ft-fsd-path-planning_amd")
pkg._capi.DEFAULT_OPTIONS.update(pkg._capi.options_from_env())  # FSDP_PACK / FSDP_PATH_MODE / ... of this tool's shell -> fsdp_set_option
pkg._capi.LIB_PATH = so
ctx = pkg.Context(device=0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
off, cones, poses = pkg.synth.make_replay_batch(N, 64, 0.15, seed=1, color=True)
ctx.upload(off, cones, poses)
ctx.run()
ctx.sync()
out = np.zeros((N, 32), np.int64)
rc = ctx._lib.fsdp_profile_path(ctx._h, ctypes.c_void_p(out.ctypes.data))
assert rc == 0
# the path kernel packs several frames into a wavefront (FSDP_PATH_G lanes per frame; default for one pass of <= 4096
# frames: 16): one row per wavefront (group 0's lane 0 keeps the clock; the groups run in lock-step, so its sections span
# the wavefront's time in them)
import os
# large batches run the three-kernel path stage: the profile below is the selected kernel's (--kernel=fit|prep|finish;
# fit_kernel: FSDP_FIT_G lanes per frame, prep / finish: 8)
G = (int(os.environ.get("FSDP_FIT_G", "4")) if which == "fit" else 8) if N > 1024 else 64
print(f"== {which} kernel ==")
if os.environ.get("SECTION_DUMP"):
    np.save(os.environ["SECTION_DUMP"], out)
FPW = 64 // G
out = out[: (N + FPW - 1) // FPW]
tot = out[:, 0]
rows = out[:, 20 : 20 + min(FPW, 4)]   # data rows pushed through the Givens pipeline per frame (first 4 frames of the wavefront)
pits = out[:, 24 : 24 + min(FPW, 4)]   # smoothing-parameter iterations per frame
print(f"QR data rows per frame: min {rows.min()}, median {int(np.median(rows))}, p99 {int(np.percentile(rows, 99))}, max {rows.max()};"
      f" per wavefront max: median {int(np.median(rows.max(axis=1)))}, max {rows.max(axis=1).max()}")
print(f"p-iterations per frame: min {pits.min()}, median {int(np.median(pits))}, max {pits.max()}")
cc = np.corrcoef(rows.max(axis=1), tot)[0, 1]
print(f"correlation(wavefront cycles, max QR rows of its frames) = {cc:.3f}; cycles per max-row: {np.median(tot / rows.max(axis=1)):.0f}")
order = np.argsort(tot)
for q in (0, len(order) // 2, len(order) - 1):
    w = order[q]
    print(f"  wavefront {w}: cycles {tot[w]}, rows {rows[w].tolist()}, p-iters {pits[w].tolist()}")

print(f"{N} frames, {len(out)} wavefronts; cycles per wavefront: min {tot.min()}, median {int(np.median(tot))}, "
      f"p90 {int(np.percentile(tot, 90))}, p99 {int(np.percentile(tot, 99))}, max {tot.max()}")
names = {0: "whole kernel", 2: "centre points (select side, match loop)", 3: "eval#1 (dense path update)", 6: "overwrite_if_too_far", 30: "mpc_prepare", 1: "fit#1 (incl. parameter)", 4: "fit#2", 5: "eval#2 + cut", 7: "fit#3", 8: "eval#3", 9: "curvature windows",
         18: "filter + sample", 19: "build_parameter (all fits)", 10: "fit: basis prep (lanes)", 11: "fit: Givens pipeline", 12: "fit: fp serial sum",
         13: "fit: back substitution", 14: "fit: residual pass", 15: "fit: fpknot", 16: "fit: part-2 Givens+back", 17: "fit: f(p) pass", 28: "  f(p): terms (lanes)", 29: "  f(p): serial sum"}
if which == "fit" and os.environ.get("FIT_DETAIL"):
    names.update({1: "fit: pass prologue (zero band, reciprocals, first fetch)", 2: "fit: pass epilogue (flush, back subst., bookkeeping)",
                  3: "fit: p start (sum of the diagonal)", 4: "fit: fpdisc", 5: "fit: smoothing iteration: band copy", 6: "fit: smoothing iteration: reciprocals"})
    for k in (7, 8, 9, 18, 19, 30): names.pop(k, None)
m = out.mean(axis=0)
print(f"{'section':<42}{'mean cycles/wave ':>20}{'% of kernel':>14}")
for k in sorted(names):
    print(f"{names[k]:<42}{m[k]:>20.0f}{100 * m[k] / m[0]:>13.1f}%")
