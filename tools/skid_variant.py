#!/usr/bin/env python3
"""tools/bench_skidpad.py against another build of the library:  python tools/skid_variant.py <lib.so> [n_instances]"""
import importlib, runpy, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
pkg = importlib.import_module("ft-fsd-path-planning_amd")
pkg._capi.DEFAULT_OPTIONS.update(pkg._capi.options_from_env())  # FSDP_PACK / FSDP_PATH_MODE / ... of this tool's shell -> fsdp_set_option
pkg._capi.LIB_PATH = Path(sys.argv[1]).resolve()
sys.argv = [str(ROOT / "tools" / "bench_skidpad.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
