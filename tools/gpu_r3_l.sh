#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3l; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
B="python $ROOT/bench.py --no-cpu-baseline"
val() { python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['pacing']; print(d['value'], d['ms_per_step'], d['pipeline']['blocking_cpi_ms'], p['mode'], p['pace_ms'], p['unpaced_ms_per_cpi_during_priming'])"; }
echo "7 cells off: $($B --pace-ms 0 --cells-per-gpu 7 --steps 20 --warmup 3 2>/dev/null | tail -1 | val)" | tee $OUT/sweep.txt
echo "7 cells auto: $($B --cells-per-gpu 7 --steps 20 --warmup 3 2>/dev/null | tail -1 | val)" | tee -a $OUT/sweep.txt
echo "7 cells auto prime 600: $($B --cells-per-gpu 7 --steps 20 --warmup 3 --prime-ms 600 2>/dev/null | tail -1 | val)" | tee -a $OUT/sweep.txt
for i in 1 2 3; do echo "driver auto #$i: $($B --steps 20 --warmup 5 2>/dev/null | tail -1 | val)"; done | tee -a $OUT/sweep.txt
echo "bad fixed pace 2.0 (must self-correct) 100: $($B --pace-ms 2.0 2>/dev/null | tail -1 | val)" | tee -a $OUT/sweep.txt
echo "auto 100: $($B 2>/dev/null | tail -1 | val)" | tee -a $OUT/sweep.txt
echo "7 cells total --cells 7: $($B --cells 7 --steps 20 --warmup 3 2>/dev/null | tail -1 | val)" | tee -a $OUT/sweep.txt
