#!/bin/bash
# Build a VARIANT of the engine library for an A/B on the GPU box: one source of csrc/ recompiled with extra -D flags and
# linked with the product's other objects (csrc/build/*.o must exist: run `python -c "import __graft_entry__ as g; g.build()"`).
#   tools/build_variant.sh <name> <source.hip> [-DMACRO=.. ...]      ->  tools/bin/libamp_<name>.so
# e.g.  tools/build_variant.sh env_tl env_step.hip -DAMP_ENV_TIMELINE        (tools/env_timeline.py)
#       tools/build_variant.sh aux0 disc.hip -DAMP_L1_STORE_AUX=0             (tools/ab_bench.sh gpurun_out/x tools/bin/libamp_aux0.so)
# tools/bin/ travels to the GPU box with the snapshot and is git-ignored.
set -euo pipefail
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/humanoid_amp_amd/csrc
mkdir -p "$root/tools/bin"
obj=/tmp/amp_variant_$name.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -I "$root/include" -I "$csrc" "$@" -c -o "$obj" "$csrc/$src"
objs=()
for o in core motion env_step compact command disc disc_train ring convert hot_step calibrate; do
  if [ "$o.hip" = "$src" ]; then objs+=("$obj"); else objs+=("$csrc/build/$o.o"); fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/bin/libamp_$name.so" "${objs[@]}"
echo "$root/tools/bin/libamp_$name.so"
