"""The CPU oracle with the record shapes of the library's wide build (max_length <= 16, max_n_neighbors <= 8, horizon <= 64):
tests/oracle_lib.py executed once more as this module, bound to oracle/liboracle_wide.so.  TEST INFRASTRUCTURE ONLY."""
import importlib.util
import sys
from pathlib import Path

_spec = importlib.util.spec_from_file_location(__name__, Path(__file__).with_name("oracle_lib.py"))
_mod = importlib.util.module_from_spec(_spec)
_mod.WIDE_SHAPES = True
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
