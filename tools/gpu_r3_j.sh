#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3k; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
B="python $ROOT/bench.py --no-cpu-baseline"
val() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['pipeline']['blocking_cpi_ms'], d['pacing'], [s['ms'] for s in d['roofline']['other_stages'] if 'covariance' in s['stage']])"; }
for i in 1 2 3 4; do echo "auto pace driver #$i: $($B --steps 20 --warmup 5 2>/dev/null | tail -1 | val)"; done | tee $OUT/sweep.txt
for i in 1 2; do echo "auto pace 100 #$i: $($B 2>/dev/null | tail -1 | val)"; done | tee -a $OUT/sweep.txt
echo "pace off driver: $($B --pace-ms 0 --steps 20 --warmup 5 2>/dev/null | tail -1 | val)" | tee -a $OUT/sweep.txt
echo "auto a256 inflight 3: $($B --ants 256 --inflight 3 --steps 12 --warmup 3 2>/dev/null | tail -1 | val)" | tee -a $OUT/sweep.txt
echo "off a256 inflight 3: $($B --pace-ms 0 --ants 256 --inflight 3 --steps 12 --warmup 3 2>/dev/null | tail -1 | val)" | tee -a $OUT/sweep.txt
echo "auto a256 inflight 4: $($B --ants 256 --inflight 4 --steps 12 --warmup 3 2>/dev/null | tail -1 | val)" | tee -a $OUT/sweep.txt
echo "auto a16: $($B --ants 16 2>/dev/null | tail -1 | val)" | tee -a $OUT/sweep.txt
echo "off a16: $($B --pace-ms 0 --ants 16 2>/dev/null | tail -1 | val)" | tee -a $OUT/sweep.txt
echo "auto 7 cells/gpu: $($B --cells-per-gpu 7 --steps 20 --warmup 3 2>/dev/null | tail -1 | val)" | tee -a $OUT/sweep.txt
echo "auto a128: $($B --ants 128 --inflight 4 --steps 20 --warmup 3 2>/dev/null | tail -1 | val)" | tee -a $OUT/sweep.txt
