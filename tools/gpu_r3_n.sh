#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/exp_n; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
tl() { # label, extra args, env
  local label=$1; local extra=$2; shift; shift
  env ISAC_TIMELINE=1 "$@" python bench.py --no-cpu-baseline --steps 100 --warmup 10 --trace-only $extra 2> $OUT/tl_$label.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$label', d['value'], d['ms_per_step'], d.get('hw_queues'))"
  python tools/_tl_parse.py $OUT/tl_$label.err | head -8
}
{
tl chain_if2 "--pace-ms 0 --inflight 2" ISAC_BENCH_CHAIN=1
tl chain_if3 "--pace-ms 0 --inflight 3" ISAC_BENCH_CHAIN=1
tl chain_if4 "--pace-ms 0 --inflight 4" ISAC_BENCH_CHAIN=1
tl nochain_if4 "--pace-ms 0 --inflight 4" A=1
tl chain_if3_lds "--pace-ms 0 --inflight 3" ISAC_BENCH_CHAIN=1 ISAC_COV_LDS_KB=82 ISAC_COV_GX=256
} > $OUT/tl_all2.txt 2>&1
cat $OUT/tl_all2.txt
