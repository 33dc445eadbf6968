#!/bin/bash
# Run a command once per A/B library (tools/_ab/libisac_hip_<name>.so, or "tree" = the library built from the tree):
#   bash tools/ab_libs.sh "tree r3 philox7" python tools/_er_variant_probe.py
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); LIB=$ROOT/5g_based_system_level_integrated_sensing_and_communication_simulator_amd/libisac_hip.so
NAMES=$1; shift
cp $LIB /tmp/libisac_hip_tree.so
for n in $NAMES; do
  if [ "$n" = tree ]; then cp /tmp/libisac_hip_tree.so $LIB; else cp $ROOT/tools/_ab/libisac_hip_$n.so $LIB; fi
  echo "== $n: $*"
  "$@" 2>&1 | tail -${AB_TAIL:-3}
done
cp /tmp/libisac_hip_tree.so $LIB
