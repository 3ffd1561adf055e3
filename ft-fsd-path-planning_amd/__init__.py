"""MI355X-native batched PathPlanner hot path (host side).

``from fsd_path_planning_amd import PathPlanner, ConeTypes, MissionTypes`` mirrors
``from fsd_path_planning import ...`` of the reference.  The compute lives in
lib/libfsdp_hip.so (hand-written HIP kernels, C ABI in include/fsdp.h).
"""
from .planner import (CapacityError, ConeTypes, MissionTypes, PathPlanner, ReferenceIndexError, ReferenceLinAlgError,  # noqa: F401
                      ReferenceUndefinedError, flatten_cones_by_type_array, pack_frames, raise_for_status)
from . import acceleration, dist, multi, replay, skidpad, stages, synth  # noqa: F401
from .stages import CalculatePath, ConeMatching, ConeSorting, ConeMatchingInput, ConeSortingInput, PathCalculationInput  # noqa: F401
from .skidpad import SkidpadBatch  # noqa: F401
from .acceleration import AccelerationBatch  # noqa: F401
from .multi import MultiPlanner, MultiSkidpadBatch  # noqa: F401
from ._capi import COMPACT_DTYPE, STANDARD, WIDE, Context, FsdpError, PATH_RESULT_DTYPE, RESULT_DTYPE, pinned_copy, pinned_empty  # noqa: F401
