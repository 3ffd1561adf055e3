import importlib, sys, numpy as np
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("ft-fsd-path-planning_amd")
ctx = pkg.Context(device=0)
off, cones, poses = pkg.synth.make_replay_batch(4096, 64, 0.15, seed=1, color=True)
pin = (pkg.pinned_copy(off, np.int32), pkg.pinned_copy(cones, np.float64), pkg.pinned_copy(poses, np.float64))
ctx.set_overlap(10)
ref = ctx.collect(ctx.submit(*pin)).copy()
print(ctx.stage_names())
for rep in range(3):
    tickets = []
    for i in range(6):
        tickets.append(ctx.submit(*pin)); print(i, ctx.stage_names()[3], end='; ')
    print()
    for i, t in enumerate(tickets):
        r = ctx.collect(t)
        bad = {f: int((r[f] != ref[f]).reshape(len(r), -1).any(axis=1).sum()) for f in r.dtype.names if not np.array_equal(r[f], ref[f], equal_nan=(r[f].dtype.kind == 'f'))}
        if bad:
            k = np.nonzero((r["status"] != ref["status"]) | (np.nan_to_num(r["path"]) != np.nan_to_num(ref["path"])).reshape(len(r), -1).any(axis=1))[0]
            print(rep, i, bad, k[:10], r["status"][k[:5]], ref["status"][k[:5]], r["path_fallback"][k[:5]], ref["path_fallback"][k[:5]])
print("done")
tickets = [ctx.submit(*pin) for _ in range(6)]
for i, t in enumerate(tickets):
    r = ctx.collect(t)
    a = np.frombuffer(r.tobytes(), np.uint8).reshape(len(r), -1); b = np.frombuffer(ref.tobytes(), np.uint8).reshape(len(r), -1)
    d = np.nonzero(a != b)
    print(i, len(d[0]), np.unique(d[1])[:20], np.unique(d[0])[:10])
    if len(d[0]):
        f = d[0][0]; print(r[f]["best_cost_left"], ref[f]["best_cost_left"], r[f]["first_k_left"], ref[f]["first_k_left"], a[f, np.unique(d[1])[:8]], b[f, np.unique(d[1])[:8]])
