#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3i; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
B="python $ROOT/bench.py --no-cpu-baseline"
val() { python -c "import json,sys; d=json.loads(sys.stdin.read()); h=d['host_timeline']; print(d['value'], d['ms_per_step'], 'collect', h['collect_wait_ms'], 'totals', h['enqueue_total_ms'], h['collect_total_ms'])"; }
for pm in 0 0.5 0.7 0.8 0.85 0.9; do for st in 100 20; do
  echo "pace $pm steps $st: $($B --pace-ms $pm --steps $st --warmup 5 2>/dev/null | tail -1 | val)"
done; done | tee $OUT/sweep.txt
for inf in 4 6; do echo "pace 0.8 inflight $inf: $($B --pace-ms 0.8 --inflight $inf 2>/dev/null | tail -1 | val)"; done | tee -a $OUT/sweep.txt
echo "pace 0.8 steps 1000: $($B --pace-ms 0.8 --steps 1000 2>/dev/null | tail -1 | val)" | tee -a $OUT/sweep.txt
