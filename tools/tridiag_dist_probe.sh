#!/bin/bash
# Development probe of eigh_tridiag_dist_kernel: correctness + host-call time of isac_eigh for orders above 64 under the development switches
# (ISAC_EIG_TRIDIAG_DIST: default = one XCD, L2-resident exchange; 0 = one-workgroup kernels; far = one XCD, write-through; s1 = all XCDs), and the
# kernel durations from a rocprofv3 kernel trace.   gpurun --timeout 600 -- 'bash tools/tridiag_dist_probe.sh'
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
{
for mode in "" 0 far s1; do
  echo "== ISAC_EIG_TRIDIAG_DIST=$mode"
  if [ -z "$mode" ]; then timeout 300 python $ROOT/tools/_tridiag_ab.py 2>&1 | grep -v amdgpu.ids
  else ISAC_EIG_TRIDIAG_DIST=$mode timeout 300 python $ROOT/tools/_tridiag_ab.py 2>&1 | grep -v amdgpu.ids; fi
done
for mode in 8 far s1 0; do
  rm -rf /tmp/pt; if [ $mode = 8 ]; then timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pt -- python $ROOT/tools/_tridiag_ab.py > /dev/null 2>&1
  else ISAC_EIG_TRIDIAG_DIST=$mode timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pt -- python $ROOT/tools/_tridiag_ab.py > /dev/null 2>&1; fi
  echo "== kernel trace (orders 65, 100, 128, 129, 200, 256, 320 x 4 calls), mode $mode"; python $ROOT/tools/prof_summary.py $(find /tmp/pt -name "*.db" | head -1) | grep -i "tridiag" 
done
ISAC_DEBUG=1 timeout 300 python $ROOT/tools/_tridiag_ab.py 2>&1 | grep "A=256 distributed" | tail -2
} > $OUT/tridiag_dist_probe.txt 2>&1
cat $OUT/tridiag_dist_probe.txt
