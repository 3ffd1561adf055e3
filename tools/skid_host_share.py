import importlib, sys, time, os
from pathlib import Path
import numpy as np
ROOT = Path("/root/repo"); sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import skidpad_support as sk
pkg = importlib.import_module("ft-fsd-path-planning_amd")
pkg._capi.DEFAULT_OPTIONS.update(pkg._capi.options_from_env())  # FSDP_PACK / FSDP_PATH_MODE / ... of this tool's shell -> fsdp_set_option
n = int(sys.argv[1]); DEPTH = 32
g = sk.load_sequence(ROOT / "tests" / "golden")
tf = sk.perturbed_instances(g, n)
T = len(g["poses"])
batches = [tuple(pkg.pinned_copy(a, dt) for a, dt in zip(sk.batch_for_step(g, t, tf), (np.int32, np.float64, np.float64))) for t in range(T)]
outs = [pkg.pinned_empty(n, pkg.RESULT_DTYPE) for _ in range(DEPTH + 1)]
b = pkg.SkidpadBatch(n, device=0); b.set_overlap(DEPTH)
for rep in range(2):
    b.reset()
    ts = tc = 0.0
    t0 = time.perf_counter(); inflight = []
    for t in range(T):
        if len(inflight) == DEPTH:
            a = time.perf_counter(); b.collect(inflight.pop(0)); tc += time.perf_counter() - a
        a = time.perf_counter(); inflight.append(b.submit(*batches[t], out=outs[t % (DEPTH + 1)])); ts += time.perf_counter() - a
    for tk in inflight:
        a = time.perf_counter(); b.collect(tk); tc += time.perf_counter() - a
    el = time.perf_counter() - t0
    print(f"n {n}: {n*T/el/1e6:.2f} M frames/s, per step {el/T*1e6:.0f} us: submit {ts/T*1e6:.0f} us, collect {tc/T*1e6:.0f} us")
