"""ctypes binding of the host-SIMT-emulated kernels (tests/emu/libfsdp_emu.so).
TEST INFRASTRUCTURE: executes the kernel sources of ft-fsd-path-planning_amd/csrc on the CPU
(one fiber per lane) so their logic can be compared with the oracle without a GPU."""
from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
EMU_DIR = ROOT / "tests" / "emu"
# tests/emu_lib_wide.py runs this file a second time as its own module with WIDE_SHAPES preset: the kernel sources compiled
# with -DFSDP_WIDE_SHAPES (tests/emu/Makefile libfsdp_emu_wide.so), compared with oracle_lib_wide
WIDE_SHAPES = bool(globals().get("WIDE_SHAPES", False))
LIB = EMU_DIR / ("libfsdp_emu_wide.so" if WIDE_SHAPES else "libfsdp_emu.so")

MAX_LEN, MAX_MATCH, PATH_POINTS = (16, 32, 64) if WIDE_SHAPES else (12, 24, 40)

SORT_DTYPE = np.dtype(
    [
        ("status", "<i4"), ("n_left", "<i4"), ("n_right", "<i4"),
        ("left_idx", "<i4", (MAX_LEN,)), ("right_idx", "<i4", (MAX_LEN,)),
        ("n_configs_left", "<i4"), ("n_configs_right", "<i4"),
        ("first_k_left", "<i4", (2,)), ("first_k_right", "<i4", (2,)),
        ("best_cost_left", "<f8"), ("best_cost_right", "<f8"),
    ],
    align=True,
)
MATCH_DTYPE = np.dtype(
    [
        ("status", "<i4"), ("n_left_v", "<i4"), ("n_right_v", "<i4"), ("pad", "<i4"),
        ("left_v", "<f8", (MAX_MATCH, 2)), ("right_v", "<f8", (MAX_MATCH, 2)),
        ("l2r", "<i4", (MAX_MATCH,)), ("r2l", "<i4", (MAX_MATCH,)),
    ],
    align=True,
)
PATH_DTYPE = np.dtype(
    [("path", "<f8", (PATH_POINTS, 4)), ("status", "<i4"), ("fallback", "<i4"), ("n_dense", "<i4"), ("pad", "<i4")],
    align=True,
)

_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.run(["make", "-s", "-C", str(EMU_DIR)], check=True)
        _lib = ctypes.CDLL(str(LIB))
        assert _lib.emu_sizeof_sort_out() == SORT_DTYPE.itemsize
    return _lib


def _p(a, t=ctypes.c_double):
    return a.ctypes.data_as(ctypes.POINTER(t))


def sort(offsets, cones, poses):
    offsets = np.ascontiguousarray(offsets, np.int32)
    cones = np.ascontiguousarray(cones, np.float64)
    poses = np.ascontiguousarray(poses, np.float64)
    n = len(offsets) - 1
    out = np.zeros(n, SORT_DTYPE)
    lib().emu_sort(ctypes.c_int(n), _p(offsets, ctypes.c_int32), _p(cones), _p(poses), ctypes.c_void_p(out.ctypes.data))
    return out


def match(offsets, cones, poses, sorted_out):
    offsets = np.ascontiguousarray(offsets, np.int32)
    cones = np.ascontiguousarray(cones, np.float64)
    poses = np.ascontiguousarray(poses, np.float64)
    n = len(offsets) - 1
    assert lib().emu_sizeof_match_out() == MATCH_DTYPE.itemsize
    out = np.zeros(n, MATCH_DTYPE)
    lib().emu_match(ctypes.c_int(n), _p(offsets, ctypes.c_int32), _p(cones), _p(poses),
                    ctypes.c_void_p(sorted_out.ctypes.data), ctypes.c_void_p(out.ctypes.data))
    return out


# lanes per frame of the one-kernel path stage (8 / 16 / 64), and the three-kernel path stage the library launches for
# large batches with 4 or 8 lanes per frame in its fit kernel (1004 / 1008)
PATH_GROUP_SIZES = (8, 16, 64, 1004, 1008, 1016, 2008)  # 2008: the wide (32-knot) three kernels


def last_retries():
    """Frames the last split-pipeline launch handed to the exact re-plan kernel."""
    return int(lib().emu_last_retries())


def path(poses, matched, group=8):
    poses = np.ascontiguousarray(poses, np.float64)
    n = len(poses)
    assert lib().emu_sizeof_path_out() == PATH_DTYPE.itemsize
    out = np.zeros(n, PATH_DTYPE)
    rc = lib().emu_path_g(ctypes.c_int(group), ctypes.c_int(n), _p(poses), ctypes.c_void_p(matched.ctypes.data),
                          ctypes.c_void_p(out.ctypes.data))
    assert rc == 0, f"no path kernel for group size {group}"
    return out


def default_path():
    out = np.zeros((PATH_POINTS, 4))
    lib().emu_default_path(_p(out))
    return out


def plan(offsets, cones, poses, group=8):
    """Full emulated pipeline; returns a structured array shaped like oracle_lib.RESULT_DTYPE.  group = lanes per frame
    of the path kernel (8: eight frames per wavefront, 16, or 64: one frame per wavefront)."""
    if WIDE_SHAPES:
        import oracle_lib_wide as oracle_lib
    else:
        import oracle_lib

    s = sort(offsets, cones, poses)
    m = match(offsets, cones, poses, s)
    lib().emu_sort_remap(ctypes.c_int(len(s)), ctypes.c_void_p(s.ctypes.data))  # (use_unknown_cones = False: indices back to the caller's array)
    p = path(poses, m, group)
    res = np.zeros(len(s), oracle_lib.RESULT_DTYPE)
    for k in ("n_left", "n_right", "left_idx", "right_idx", "n_configs_left", "n_configs_right", "first_k_left",
              "first_k_right", "best_cost_left", "best_cost_right"):
        res[k] = s[k]
    for k in ("n_left_v", "n_right_v", "left_v", "right_v", "l2r", "r2l"):
        res[k] = m[k]
    res["path"] = p["path"]
    res["path_fallback"] = p["fallback"]
    st = s["status"].copy()
    st = np.where(m["status"] != 0, m["status"], st)
    st = np.where(p["status"] != 0, p["status"], st)
    res["status"] = st
    return res, p["n_dense"]


SKID_STATE_DTYPE = np.dtype(
    [("has_original", "<i4"), ("relocalized", "<i4"), ("index_along_path", "<i4"), ("reloc_step", "<i4"), ("index_hist", "<i4", (32,)),
     ("orig", "<f8", (4,)),
     ("translation", "<f8", (2,)), ("right_calc", "<f8", (2,)), ("rotation", "<f8"), ("prev", "<f8", (PATH_POINTS, 4))],
    align=True,
)
SKID_INFO_DTYPE = np.dtype([("relocalized", "<i4"), ("index_along_path", "<i4"), ("translation", "<f8", (2,)), ("rotation", "<f8")], align=True)


def skidpad_constants(table):
    """(reference centres (2,2) [right, left], table spacing) from csrc/skidpad_kernel.h skid_centers_kernel."""
    table = np.ascontiguousarray(table, np.float64)
    out = np.zeros(5)
    lib().emu_skidpad_constants(_p(table), ctypes.c_int(len(table)), _p(out))
    return out[:4].reshape(2, 2).copy(), float(out[4])


class SkidpadEmu:
    """The skidpad kernels under the emulator: n planner instances, stateful."""

    def __init__(self, n, table, noise, ref, mean_distance):
        assert lib().emu_sizeof_skid_state() == SKID_STATE_DTYPE.itemsize and lib().emu_sizeof_skid_info() == SKID_INFO_DTYPE.itemsize
        self.n = n
        self.half = np.ascontiguousarray(table[::2], np.float64)
        self.noise = np.ascontiguousarray(noise, np.float64).ravel()
        self.ref = np.ascontiguousarray(ref, np.float64).ravel()
        self.md = float(mean_distance)
        self.states = np.zeros(n, SKID_STATE_DTYPE)
        self.states["prev"] = default_path()
        self.sync = np.zeros(n + 1, np.uint32)  # ticket counter + steps published per instance
        self.ticket_base = np.zeros(1, np.uint32)
        self.step_no = 0

    def step(self, offsets, cones, poses):
        (res,) = self.steps([(offsets, cones, poses)])
        return res

    def steps_packed(self, frames, lanes=16):
        """The same as ``steps`` through the packed kernels (select -> prep -> fit -> finish -> commit, lanes per frame 16
        or 8 with a 4-lane fit) -> ([(out, info), ...], number of steps the planners' own wavefronts had to plan)."""
        k = len(frames)
        offs = [np.ascontiguousarray(f[0], np.int32) for f in frames]
        cones = [np.ascontiguousarray(f[1], np.float64).reshape(-1, 3) for f in frames]
        poses = [np.ascontiguousarray(f[2], np.float64) for f in frames]
        outs = [np.zeros(self.n, PATH_DTYPE) for _ in range(k)]
        infos = [np.zeros(self.n, SKID_INFO_DTYPE) for _ in range(k)]
        ptrs = lambda arrs: (ctypes.c_void_p * k)(*[a.ctypes.data for a in arrs])
        serial = lib().emu_skidpad_steps_packed(
            ctypes.c_int(lanes), ctypes.c_int(self.n), ctypes.c_int(k), ctypes.c_int(self.step_no), ptrs(offs), ptrs(cones), ptrs(poses),
            ctypes.c_void_p(self.states.ctypes.data), _p(self.half), ctypes.c_int(len(self.half)), _p(self.noise),
            ctypes.c_int(len(self.noise)), _p(self.ref), ctypes.c_double(self.md), ptrs(outs), ptrs(infos),
            ctypes.c_void_p(self.sync.ctypes.data))
        self.step_no += k
        return list(zip(outs, infos)), int(serial)

    def steps(self, frames):
        """len(frames) <= 16 consecutive steps [(offsets, cones, poses), ...] in one skid_path_kernel launch (one wavefront
        per (instance, step): csrc/skidpad_kernel.h "Steps in flight") -> [(out, info), ...]."""
        k = len(frames)
        offs = [np.ascontiguousarray(f[0], np.int32) for f in frames]
        cones = [np.ascontiguousarray(f[1], np.float64).reshape(-1, 3) for f in frames]
        poses = [np.ascontiguousarray(f[2], np.float64) for f in frames]
        outs = [np.zeros(self.n, PATH_DTYPE) for _ in range(k)]
        infos = [np.zeros(self.n, SKID_INFO_DTYPE) for _ in range(k)]
        ptrs = lambda arrs: (ctypes.c_void_p * k)(*[a.ctypes.data for a in arrs])
        lib().emu_skidpad_steps(
            ctypes.c_int(self.n), ctypes.c_int(k), ctypes.c_int(self.step_no), ptrs(offs), ptrs(cones), ptrs(poses),
            ctypes.c_void_p(self.states.ctypes.data), _p(self.half), ctypes.c_int(len(self.half)), _p(self.noise),
            ctypes.c_int(len(self.noise)), _p(self.ref), ctypes.c_double(self.md), ptrs(outs), ptrs(infos),
            ctypes.c_void_p(self.sync.ctypes.data), ctypes.c_void_p(self.ticket_base.ctypes.data))
        self.step_no += k
        return list(zip(outs, infos))


class params:
    """with emu_lib.params(dict(max_dist=5.5)): ... — the kernel sources with non-default configuration constants."""

    def __init__(self, overrides):
        import oracle_lib

        self.v = oracle_lib.param_vector(overrides)
        self.d = oracle_lib.param_vector(None)

    def __enter__(self):
        lib().emu_set_params(_p(self.v))

    def __exit__(self, *a):
        lib().emu_set_params(_p(self.d))
