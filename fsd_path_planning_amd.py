"""Import alias: the package directory is named ``ft-fsd-path-planning_amd`` (not a Python
identifier), so ``import fsd_path_planning_amd`` loads it through importlib."""
import importlib
import sys

_pkg = importlib.import_module("ft-fsd-path-planning_amd")
sys.modules[__name__] = _pkg
