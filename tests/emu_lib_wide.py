"""The host-SIMT-emulated kernels compiled with the wide build's shapes (-DFSDP_WIDE_SHAPES: max_length <= 16,
max_n_neighbors <= 8, horizon <= 64): tests/emu_lib.py executed once more as this module, bound to
tests/emu/libfsdp_emu_wide.so.  TEST INFRASTRUCTURE ONLY."""
import importlib.util
import sys
from pathlib import Path

_spec = importlib.util.spec_from_file_location(__name__, Path(__file__).with_name("emu_lib.py"))
_mod = importlib.util.module_from_spec(_spec)
_mod.WIDE_SHAPES = True
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
