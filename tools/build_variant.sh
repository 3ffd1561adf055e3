#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG ...]: an experiment build of the library -> ft-fsd-path-planning_amd/lib/variants/NAME.so
# (the flags of __graft_entry__.py HIP_FLAGS + the given ones); A/B runs: tools/ab_variants.py
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p "$ROOT/ft-fsd-path-planning_amd/lib/variants"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -sink-insts-to-avoid-spills=1 "$@" -fPIC -shared \
  "$ROOT/ft-fsd-path-planning_amd/csrc/fsdp_lib.hip" -o "$ROOT/ft-fsd-path-planning_amd/lib/variants/$name.so" -ldl 2>&1 | grep -v "warning: failed to meet occupancy\|^ *[0-9]* |\|^ *| *\^\|warning generated\|^In file included" || true
echo "built $name"
