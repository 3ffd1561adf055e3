import sys, importlib, json
sys.path.insert(0,'/root/repo')
import importlib.util
spec = importlib.util.spec_from_file_location("bench_module", "/root/repo/bench.py")
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
pkg = importlib.import_module("ft-fsd-path-planning_amd")
pkg._capi.DEFAULT_OPTIONS.update(pkg._capi.options_from_env())  # FSDP_PACK / FSDP_PATH_MODE / ... of this tool's shell -> fsdp_set_option
for nctx, depth in ((4,3),(4,5),(2,5),(2,8),(1,10)):
    r = bench.multi_planner_leg(pkg, [0]*nctx, 4096, depth, max(8, 100//nctx), 1, None)
    print(nctx, depth, {k:(round(v,4) if isinstance(v,float) else v) for k,v in r.items() if k in ('value','host_us_per_submitted_frame','pageable_input_frames_per_s','error')}, flush=True)
