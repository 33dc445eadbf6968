#!/bin/bash
# A/B library for tools/stress_race.py: the tree's sources with the two round-5 fixes taken OUT again --
#   (i)  upload_now = plain hipMemcpy on the NULL stream (before c98d090), (ii) the SINR mean copied back into a pageable stack variable (before c8f9239)
# -> tools/_ab/libisac_hip_prefix.so (git-ignored; travels with gpurun).  Shows the NULL-stream race at a rate, or shows that it cannot be provoked.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd); PKG=$ROOT/5g_based_system_level_integrated_sensing_and_communication_simulator_amd
W=/tmp/isac_var_prefix; rm -rf $W; mkdir -p $W/pkg $W/include; cp -r $PKG/csrc $W/pkg/; rm -rf $W/pkg/csrc/build; cp $ROOT/include/isac.h $W/include/
python3 - "$W/pkg/csrc" <<'PY'
import re, sys
d = sys.argv[1]
p = d + "/isac_common.hpp"; s = open(p).read()
a = "  ISAC_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));\n  ISAC_HIP(hipStreamSynchronize(ctx->stream));\n  return ISAC_OK;\n}"
assert a in s
s = s.replace(a, "  ISAC_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));\n  return ISAC_OK;\n}")
open(p, "w").write(s)
p = d + "/cqi.hip"; s = open(p).read()
a = "    ISAC_HIP(hipMemcpyAsync(ctx->pinned_csi, d_mean, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));\n    ISAC_HIP(hipStreamSynchronize(ctx->stream));\n    const double m = *(const double*)ctx->pinned_csi;"
assert a in s
s = s.replace(a, "    double m_stack = 0.0;\n    ISAC_HIP(hipMemcpyAsync(&m_stack, d_mean, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));\n    ISAC_HIP(hipStreamSynchronize(ctx->stream));\n    const double m = m_stack;")
open(p, "w").write(s)
PY
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function -Wno-unused-variable -Wno-unused-value -Wno-unused-result -ffp-contract=on"
cd $W/pkg/csrc
for f in capi echo rdm music cdl cdl_os cqi los; do /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o $f.o & done; wait
mkdir -p $ROOT/tools/_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/_ab/libisac_hip_prefix.so *.o -Wl,-soname,libisac_hip.so -Wl,--no-undefined
echo built tools/_ab/libisac_hip_prefix.so
