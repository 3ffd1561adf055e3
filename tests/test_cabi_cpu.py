"""CPU-side checks of the product boundary: the C-ABI library loads and exports every symbol
include/fsdp.h declares, the result struct layout matches the Python mirror, and — without a GPU —
the product fails loudly instead of falling back to anything."""
import ctypes
import importlib
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as ge

    ge.build_hip()
    return importlib.import_module("ft-fsd-path-planning_amd")


def test_library_exports_every_declared_symbol(pkg):
    header = (ROOT / "include" / "fsdp.h").read_text()
    declared = set(re.findall(r"\b(fsdp_[a-z0-9_]+)\s*\(", header))
    declared -= {"fsdp_ctx"}
    assert declared == set(pkg._capi.EXPORTED_SYMBOLS)
    # both builds of the sources (include/fsdp.h: the standard shapes and -DFSDP_WIDE_SHAPES) export the whole ABI and report
    # the shapes the binding mirrors
    for shapes in (pkg._capi.STANDARD, pkg._capi.WIDE):
        lib = ctypes.CDLL(str(shapes.lib_path))
        for sym in sorted(declared):
            assert hasattr(lib, sym), (shapes.name, sym)
        assert lib.fsdp_result_size() == shapes.result_dtype.itemsize
        got = (ctypes.c_int32 * 4)()
        lib.fsdp_shapes(got)
        assert list(got) == [shapes.max_len, shapes.max_neighbors, shapes.max_match, shapes.path_points]
    assert list(got) == [16, 8, 32, 64] and pkg._capi.WIDE.result_dtype.itemsize == 3528
    header_wide = re.search(r"#ifdef FSDP_WIDE_SHAPES(.*?)#else", header, re.S).group(1)
    assert [int(v) for v in re.findall(r"#define FSDP_\w+ (\d+)", header_wide)] == [16, 8, 32, 64]


def test_no_cpu_fallback(pkg):
    lib = pkg._capi.load()
    if lib.fsdp_device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(pkg.FsdpError, match="no HIP device"):
        pkg.PathPlanner(pkg.MissionTypes.trackdrive)


def test_product_does_not_import_oracle():
    """The product never imports, links or executes anything under oracle/ (or the emulator)."""
    pat = re.compile(r"import\s+oracle_lib|from\s+oracle|liboracle|#include\s*[<\"].*oracle|libfsdp_emu|hip_emu\.h|import\s+emu_lib")
    for p in (ROOT / "ft-fsd-path-planning_amd").rglob("*"):
        if p.suffix in (".py", ".h", ".hip", ".cpp") and p.is_file():
            assert not pat.search(p.read_text()), p


def test_host_side_packing(pkg):
    frames = [([np.zeros((0, 2)), np.array([[1.0, -1.5]]), np.array([[1.0, 1.5], [4.0, 1.5]]), [], []], [0.0, 0.0], 0.0),
              (np.array([[0.0, 1.0, 2.0]]), np.array([1.0, 2.0]), np.array([0.0, 1.0]))]
    off, cones, poses = pkg.pack_frames(frames)
    assert off.tolist() == [0, 3, 4]
    assert cones[:3, 2].tolist() == [1.0, 2.0, 2.0]  # reference flatten order: UNKNOWN, RIGHT, LEFT, ...
    assert np.allclose(poses[0], [0, 0, 1, 0]) and np.allclose(poses[1], [1, 2, 0, 1])
    with pytest.raises(ValueError):
        pkg.planner._direction_to_array([1.0, 2.0, 3.0])
    assert int(pkg.ConeTypes.LEFT) == 2 and int(pkg.ConeTypes.YELLOW) == 1 and int(pkg.MissionTypes.trackdrive) == 4


def test_replay_loader_matches_reference_schema(pkg, golden_dir, tmp_path):
    """load_data_json / remove_color_info / mission-by-filename mirror demo/json_demo.py:38-51,255-275 (own code)."""
    import json

    g = np.load(golden_dir / "skidpad_sequence.npz")
    frames = []
    for t in range(5, 30):
        xyt = g["cones"][g["offsets"][t] : g["offsets"][t + 1]]
        frames.append({"car_position": g["poses"][t, :2].tolist(), "car_direction": g["poses"][t, 2:].tolist(),
                       "slam_cones": [xyt[xyt[:, 2] == k, :2].tolist() for k in range(5)]})
    f = tmp_path / "my_skidpad_run.json"
    f.write_text(json.dumps(frames))
    pos, dirs, obs = pkg.replay.load_data_json(f)
    assert pos.shape == (25, 2) and dirs.shape == (25, 2) and len(obs) == 25 and len(obs[0]) == 5
    assert all(o.shape[1] == 2 for o in obs[3])
    pos2, dirs2, obs2 = pkg.replay.load_data_json(f, remove_color_info=True)
    for a, b in zip(obs, obs2):
        assert sum(len(x) for x in a) == len(b[0]) and all(len(x) == 0 for x in b[1:])
        assert np.array_equal(np.concatenate(a), b[0])  # stacked in type order
    # the packed form the loader is built around equals the frames it was written from, and pack_frames of the per-type lists
    rec = pkg.replay.Recording.from_json(f)
    assert np.array_equal(rec.cones, g["cones"][g["offsets"][5] : g["offsets"][30]]) and np.array_equal(rec.poses, g["poses"][5:30])
    assert np.array_equal(rec.offsets, g["offsets"][5:31] - g["offsets"][5])
    off, cones, poses = pkg.planner.pack_frames(list(zip(obs, pos, dirs)))
    assert np.array_equal(off, rec.offsets) and np.array_equal(cones, rec.cones) and np.array_equal(poses, rec.poses)
    assert (rec.without_color().cones[:, 2] == 0).all() and np.array_equal(rec.without_color().cones[:, :2], rec.cones[:, :2])
    assert pkg.replay.select_mission_by_filename(f.name) == pkg.MissionTypes.skidpad
    assert pkg.replay.select_mission_by_filename("accel_run.json") == pkg.MissionTypes.acceleration
    assert pkg.replay.select_mission_by_filename("fsg_19_2_laps.json") == pkg.MissionTypes.trackdrive


def test_bench_extras_never_fail_the_line():
    """bench.py's config-5 extra runs tools/bench_skidpad.py as a child process; without a GPU (here) the child fails, and
    the extra reports that instead of raising — the headline line must come out whatever happens to its extras."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_module", ROOT / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    r = bench.skidpad_leg(n_instances=8, timeout=120)
    assert r["value"] is None and "error" in r


def test_in_process_shards_are_the_ranks_shards():
    """multi.shard_ranges (one process, a context per GPU) cuts a batch exactly where dist.frame_range cuts it for ranks when
    the counts divide, and always into contiguous ranges that cover the batch with sizes differing by at most one."""
    import importlib

    pkg = importlib.import_module("ft-fsd-path-planning_amd")
    for n in (0, 1, 7, 4096, 65536, 65537):
        for g in (1, 2, 3, 4, 8):
            r = pkg.multi.shard_ranges(n, g)
            assert len(r) == g and r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [hi - lo for lo, hi in r]
            assert max(sizes) - min(sizes) <= 1
            if n % g == 0:
                assert r == [pkg.dist.frame_range(k, g, n) for k in range(g)]


def test_vectorised_fma_is_correctly_rounded_and_reproduces_numpy_dot():
    """acceleration.fma (float64 operations only: Dekker product, TwoSum, inner sum rounded to odd) is the IEEE fused multiply-add —
    against exact rational arithmetic on random, cancelling and double-rounding-prone operands — and with it the two orders in
    which NumPy / OpenBLAS rotate points (one point: gemv, several: gemm) are reproduced bit for bit for many angles at once."""
    import importlib
    from fractions import Fraction

    acc = importlib.import_module("ft-fsd-path-planning_amd.acceleration")
    rng = np.random.default_rng(0)
    n = 3000
    a = rng.normal(0, 10 ** rng.uniform(-3, 3, n))
    b = rng.normal(0, 10 ** rng.uniform(-3, 3, n))
    for c in (rng.normal(0, 10 ** rng.uniform(-3, 3, n)), -(a * b) * (1 + rng.integers(-4, 5, n) * 2.0 ** -52), (a * b) * 2.0 ** rng.integers(-60, 60, n)):
        got = acc.fma(a, b, c)
        want = np.array([float(Fraction(float(x)) * Fraction(float(y)) + Fraction(float(z))) for x, y, z in zip(a, b, c)])
        assert np.array_equal(got, want)
    theta = rng.uniform(-3, 3, 200)
    pts = rng.normal(0, 40, (200, 2))
    x, y = acc.rotate_single_points(pts[:, 0], pts[:, 1], theta)
    rows = rng.normal(0, 40, (200, 7, 2))
    rx, ry = acc.rotate_point_rows(rows[:, :, 0], rows[:, :, 1], theta[:, None])
    for i in range(200):
        assert np.array_equal(acc._rotate(pts[i], theta[i]), [x[i], y[i]])           # np.dot(1-D, 2-D): gemv
        assert np.array_equal(acc._rotate(rows[i], theta[i]), np.column_stack([rx[i], ry[i]]))  # np.dot(2-D, 2-D): gemm


def test_compact_record_mirror_and_option_plumbing(pkg):
    """fsdp_compact_result (include/fsdp.h) and its NumPy mirror: 1384 bytes in the standard build (the bytes SURVEY 8d counts as a frame's
    output), field offsets as the struct lays them out; and the library has no environment switch for its behaviour — the product sources
    call getenv once (the path of librccl), tools translate their shell's FSDP_* variables through _capi.options_from_env."""
    c = pkg._capi.STANDARD.compact_dtype
    assert c.itemsize == 1384 and pkg.COMPACT_DTYPE is c
    assert [c.fields[f][1] for f in ("path", "left_idx", "right_idx", "status", "n_left", "n_right", "path_fallback", "n_dense")] == [0, 1280, 1328, 1376, 1380, 1381, 1382, 1383]
    assert pkg._capi.WIDE.compact_dtype.itemsize == 64 * 32 + 2 * 16 * 4 + 8
    header = (ROOT / "include" / "fsdp.h").read_text()
    for name in pkg._capi.OPTION_NAMES:
        assert f'"{name}"' in header, name  # every option the binding knows is documented at fsdp_set_option
    assert pkg._capi.options_from_env({"FSDP_PACK": "1", "FSDP_PATH_MODE": "split", "FSDP_FIT_G": "8", "FSDP_ALWAYS_ROUTE": "1"}) == {
        "pack": 2, "path_mode": 2, "fit_g": 8, "always_route": 1}
    assert pkg._capi.options_from_env({}) == {} and pkg._capi.DEFAULT_OPTIONS == {}
    src = "".join(p.read_text() for p in (ROOT / "ft-fsd-path-planning_amd" / "csrc").glob("*") if p.suffix in (".h", ".hip"))
    assert re.findall(r'getenv\("(\w+)"\)', src) == ["FSDP_RCCL_LIB"]
