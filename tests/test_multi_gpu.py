"""One process, several GPU contexts (multi.MultiPlanner / MultiSkidpadBatch): contiguous frame
ranges — or planner instances — per context, submitted from one host thread, must return the bytes of ONE context that
plans everything.  The 1-GPU box runs two / three contexts on device 0; with a second GPU visible the same tests also run
on devices [0, 1]."""
import importlib

import numpy as np
import pytest

import skidpad_support as sk

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return importlib.import_module("ft-fsd-path-planning_amd")


def _device_sets(pkg):
    n = int(pkg._capi.load().fsdp_device_count())
    sets = [[0, 0], [0, 0, 0]]
    if n >= 2:
        sets.append([0, 1])
    if n >= 4:
        sets.append(list(range(n)))
    return sets


def _same(a, b):
    return a.dtype == b.dtype and all(np.ascontiguousarray(a[k]).tobytes() == np.ascontiguousarray(b[k]).tobytes() for k in a.dtype.names)


def test_shard_ranges_cover_the_batch(pkg):
    for n in (0, 1, 2, 5, 4096, 65536, 65537):
        for g in (1, 2, 3, 8):
            r = pkg.multi.shard_ranges(n, g)
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(hi - lo for lo, hi in r) - min(hi - lo for lo, hi in r) <= 1


@pytest.mark.parametrize("config", ["cfg2", "cfg4", "noisy_nocolor", "ragged"])
def test_sharded_plan_batch_equals_one_context(pkg, config):
    """BASELINE configs 2 and 4 (one GPU's shard of it) and two awkward batches: frames on the exact routes (sort_big /
    path_retry), empty frames, frame counts that do not divide."""
    if config == "cfg2":
        off, cones, poses = pkg.synth.make_replay_batch(4096, 64, 0.15, seed=1, color=True)
    elif config == "cfg4":
        off, cones, poses = pkg.synth.make_config4_shard(0, 4099, 100, 0.1, seed=7)
    elif config == "noisy_nocolor":
        off, cones, poses = pkg.synth.make_replay_batch(1501, 100, 0.0, seed=8, frame_noise=0.3, random_pose=True, color=False,
                                                        lateral_noise=0.5, heading_noise=0.2)
    else:
        o, c, p = pkg.synth.make_replay_batch(257, 40, 0.15, seed=3, color=True)
        counts = np.diff(o).copy()
        counts[::7] = 0          # empty frames
        counts[5::11] = 2        # fewer than three cones
        keep = np.concatenate([np.arange(o[k], o[k] + counts[k]) for k in range(len(counts))]).astype(int)
        off, cones, poses = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32), c[keep], p
    one = pkg.Context(device=0)
    ref = one.plan_batch(off, cones, poses)
    for devs in _device_sets(pkg):
        mp = pkg.MultiPlanner(devs)
        got = mp.plan_batch(off, cones, poses)
        assert _same(got, ref), (config, devs)
        # previous paths travel with their frames
        prev = np.repeat(one.default_path()[None], len(poses), axis=0)
        prev[:, :, 1] += np.linspace(-0.5, 0.5, len(poses))[:, None]
        assert _same(mp.plan_batch(off, cones, poses, prev_paths=prev), one.plan_batch_sequential(off, cones, poses, prev)), (config, devs)
        # compact records (fsdp_submit_compact on every shard): the same fields, pageable and page-locked (zero-copy slices)
        want = one.plan_batch(off, cones, poses, compact=True)
        assert all(np.array_equal(want[f], ref[f], equal_nan=True) for f in ("path", "left_idx", "right_idx", "status"))
        assert _same(mp.plan_batch(off, cones, poses, compact=True), want), (config, devs)
        pin = (pkg.pinned_copy(off, np.int32), pkg.pinned_copy(cones, np.float64), pkg.pinned_copy(poses, np.float64))
        assert _same(mp.plan_batch(*pin, compact=True), want), (config, devs)
        mp.close()


def test_stream_of_batches_over_two_contexts(pkg):
    """plan_stream keeps `depth` batches in flight on every context; results in order, bytes of serial plan_batch calls,
    for batch sizes from 0 to 3000 frames (staging buffers are reused and regrown)."""
    sizes = [700, 0, 1, 3000, 64, 1200, 5, 2048, 333, 900]
    batches = [pkg.synth.make_replay_batch(max(n, 1), 64, 0.15, seed=20 + k, color=bool(k % 2)) for k, n in enumerate(sizes)]
    batches = [(o[: n + 1], c[: o[n]], p[:n]) for (o, c, p), n in zip(batches, sizes)]
    one = pkg.Context(device=0)
    refs = [one.plan_batch(*b) for b in batches]
    for devs in _device_sets(pkg)[:2] + _device_sets(pkg)[2:3]:
        mp = pkg.MultiPlanner(devs, overlap=3)
        got = list(mp.plan_stream(batches))
        assert len(got) == len(refs)
        for k, (a, b) in enumerate(zip(got, refs)):
            assert _same(a, b), (devs, k)
        mp.close()


def test_pathplanner_devices_kwarg(pkg):
    off, cones, poses = pkg.synth.make_replay_batch(600, 64, 0.15, seed=4, color=True)
    ref = pkg.PathPlanner(pkg.MissionTypes.trackdrive, device=0).plan_batch(off, cones, poses)
    pp = pkg.PathPlanner(pkg.MissionTypes.trackdrive, devices=[0, 0])
    assert _same(pp.plan_batch(off, cones, poses), ref)
    # the reference-shaped single-frame call still works on such a planner (first device)
    xyt = cones[off[0] : off[1]]
    path = pp.calculate_path_in_global_frame(xyt, poses[0, :2], poses[0, 2:])
    assert np.array_equal(path, ref[0]["path"][: pp._ctx.horizon])
    all_gpus = pkg.PathPlanner(pkg.MissionTypes.trackdrive, devices="all")
    assert _same(all_gpus.plan_batch(off, cones, poses), ref)


def test_sharded_skidpad_instances_equal_one_batch(pkg, golden_dir):
    """64 skidpad planners over 90 frames of the recording (relocalization inside), rigidly perturbed starts: sharded by
    instance over two / three contexts, one step at a time and as a replay submitted ahead == one SkidpadBatch."""
    n = 64
    g = sk.load_sequence(golden_dir)
    tf = sk.perturbed_instances(g, n)
    frames = [sk.batch_for_step(g, t, tf) for t in range(90)]
    one = pkg.SkidpadBatch(n, device=0)
    ref = []
    for f in frames:
        res, info = one.step(*f)
        ref.append((res.copy(), info.copy()))
    assert ref[-1][1]["relocalized"].mean() > 0.9
    for devs in _device_sets(pkg):
        mb = pkg.SkidpadBatch(n, devices=devs)
        assert isinstance(mb, pkg.MultiSkidpadBatch) and sum(hi - lo for lo, hi in mb.ranges) == n
        for t in range(30):
            res, info = mb.step(*frames[t])
            assert _same(res, ref[t][0]) and _same(info, ref[t][1]), (devs, t)
        for t, (res, info) in enumerate(mb.replay(frames[30:], 16), start=30):
            assert _same(res, ref[t][0]) and _same(info, ref[t][1]), (devs, t)


def test_page_locked_batch_is_sharded_without_a_host_copy(pkg):
    """A batch that already lives in page-locked memory goes to its GPUs as slices of the caller's own arrays (fsdp_submit with
    cone_offsets[0] != 0, Context.submit_slice): same bytes as one context, no staging, and the result is the page-locked block
    the GPUs wrote — handed back, not copied, and reused for a later batch once it has been dropped."""
    capi = pkg._capi
    off, cones, poses = pkg.synth.make_replay_batch(3001, 64, 0.15, seed=5, color=True)
    one = pkg.Context(device=0)
    ref = one.plan_batch(off, cones, poses)
    prev = np.repeat(one.default_path()[None], len(poses), axis=0)
    prev[:, :, 2] += np.linspace(-0.4, 0.4, len(poses))[:, None]
    ref_prev = one.plan_batch_sequential(off, cones, poses, prev)
    p_off, p_cones, p_poses, p_prev = capi.pinned_copy(off, np.int32), capi.pinned_copy(cones), capi.pinned_copy(poses), capi.pinned_copy(one.pad_paths(prev))
    assert capi.is_pinned(p_cones) and not capi.is_pinned(cones) and not capi.is_pinned(p_cones[1:].reshape(-1)[:-1][::2])
    for devs in _device_sets(pkg):
        mp = pkg.MultiPlanner(devs)
        got = mp.plan_batch(p_off, p_cones, p_poses)
        assert mp.zero_copy_batches == 1 and mp.staged_batches == 0
        assert _same(got, ref), devs
        assert capi.is_pinned(got)
        where = got.ctypes.data
        del got
        got = mp.plan_batch(p_off, p_cones, p_poses, prev_paths=p_prev)
        assert got.ctypes.data == where  # the block of the dropped result
        assert _same(got, ref_prev), devs
        assert mp.zero_copy_batches == 2 and mp.host_frames == 2 * len(poses) and mp.host_seconds > 0
        # the same batch from pageable arrays: staged by the contexts' worker threads, same bytes
        got2 = mp.plan_batch(off, cones, poses)
        assert mp.staged_batches == 1 and _same(got2, ref), devs
        # a stream of page-locked batches, results kept: every one owns its block
        outs = list(mp.plan_stream([(p_off, p_cones, p_poses)] * 5, depth=2))
        assert len({o.ctypes.data for o in outs}) == 5 and all(_same(o, ref) for o in outs)
        mp.close()


def test_slices_of_a_batch_through_the_c_abi(pkg):
    """include/fsdp.h: cone_offsets[0] need not be 0.  Slices [lo, hi) of one batch — pageable and page-locked, default and
    UNKNOWN-filtering contexts (the three input routes: copy engine, the sorting kernel reading the host buffer, the staging
    kernel) — equal the same frames planned as a batch of their own; sorted indices stay frame-relative."""
    capi = pkg._capi
    off, cones, poses = pkg.synth.make_replay_batch(900, 64, 0.15, seed=6, color=True)
    rng = np.random.default_rng(0)
    cones = cones.copy()
    cones[rng.random(len(cones)) < 0.3, 2] = 0.0  # some UNKNOWN cones for the filtering context
    for params in (None, {"use_unknown_cones": False}):
        ctx = pkg.Context(device=0, params=params)
        ref = ctx.plan_batch(off, cones, poses)
        for pinned in (False, True):
            o, c, p = (capi.pinned_copy(off, np.int32), capi.pinned_copy(cones), capi.pinned_copy(poses)) if pinned else (off, cones, poses)
            out = capi.pinned_empty(len(poses), capi.RESULT_DTYPE)
            out["status"] = -7
            ctx.set_overlap(3)
            tickets = [ctx.submit_slice(lo, hi, o, c, p, None, out) for lo, hi in ((0, 300), (300, 301), (301, 900))]
            for t in tickets:
                ctx.collect(t)
            ctx.set_overlap(1)
            assert _same(out, ref), (params, pinned)
    # the blocking entry point takes a slice as well
    import ctypes

    ctx = pkg.Context(device=0)
    ref = ctx.plan_batch(off, cones, poses)
    lo, hi = 123, 457
    out = np.zeros(hi - lo, dtype=capi.RESULT_DTYPE)
    rc = ctx._lib.fsdp_plan_batch(ctx._h, hi - lo, capi._ip(off[lo:]), capi._dp(cones), capi._dp(poses[lo:]), ctypes.c_void_p(out.ctypes.data))
    assert rc == 0 and _same(out, ref[lo:hi])


def test_sharded_skidpad_compact_results(pkg, golden_dir):
    """MultiSkidpadBatch forwards `compact` (round-4 advisor: it used to rely on the part re-deriving it from out.dtype) and
    refuses an `out` of the other dtype; compact replays equal the one-context compact replay."""
    n = 32
    g = sk.load_sequence(golden_dir)
    tf = sk.perturbed_instances(g, n)
    frames = [sk.batch_for_step(g, t, tf) for t in range(40)]
    one = pkg.SkidpadBatch(n, device=0)
    ref = [(r.copy(), i.copy()) for r, i in one.replay(frames, 8, compact=True)]
    mb = pkg.SkidpadBatch(n, devices=[0, 0])
    got = list(mb.replay(frames, 8, compact=True))
    assert got[0][0].dtype == pkg._capi.PATH_RESULT_DTYPE
    for t, ((r, i), (rr, ri)) in enumerate(zip(got, ref)):
        assert _same(r, rr) and _same(i, ri), t
    mb.reset()
    with pytest.raises(ValueError):
        mb.submit(*frames[0], out=pkg._capi.pinned_empty(n, pkg._capi.RESULT_DTYPE), compact=True)
