#!/bin/bash
# Development A/B builds: tools/build_variant.sh <name> '<sed expression>' [file]  ->  tools/_ab/libisac_hip_<name>.so (git-ignored; travels with gpurun).
# The tree's sources are copied to a scratch directory, edited there with sed, and built with the flags of _build.py.
set -eu
NAME=$1; EXPR=$2; FILE=${3:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd); PKG=$ROOT/5g_based_system_level_integrated_sensing_and_communication_simulator_amd
W=/tmp/isac_var_$NAME; rm -rf $W; mkdir -p $W/pkg $W/include; cp -r $PKG/csrc $W/pkg/; rm -rf $W/pkg/csrc/build; cp $ROOT/include/isac.h $W/include/
# csrc includes "../../include/isac.h" relative to csrc
if [ -n "$FILE" ]; then sed -i -E "$EXPR" $W/pkg/csrc/$FILE; else sed -i -E "$EXPR" $W/pkg/csrc/*.h* ; fi
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function -Wno-unused-variable -Wno-unused-value -Wno-unused-result -ffp-contract=on"
cd $W/pkg/csrc
for f in capi echo rdm music cdl cdl_os cqi los; do /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o $f.o & done; wait
mkdir -p $ROOT/tools/_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/_ab/libisac_hip_$NAME.so *.o -Wl,-soname,libisac_hip.so -Wl,--no-undefined
echo built tools/_ab/libisac_hip_$NAME.so
