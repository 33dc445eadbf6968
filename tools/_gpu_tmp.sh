#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/exp_cov; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
for sy in 8 1 2 4 16 32 1024 8; do
ISAC_COV_SYNC=$sy ISAC_COV_WGTIMES=1 python bench.py --no-cpu-baseline --steps 12 --warmup 3 --inflight 1 --prime-ms 100 --trace-only 2> $OUT/covwgn.err > /dev/null
echo "sync $sy: $(grep COVWG $OUT/covwgn.err | tail -1 | cut -c1-70)"
done
