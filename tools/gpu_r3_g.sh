#!/bin/bash
# round-3 GPU call G: host-side stall diagnosis (timeline stats), scheduler / interrupt-mode experiments
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3g; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
B="python $ROOT/bench.py --no-cpu-baseline"
val() { python -c "import json,sys; d=json.loads(sys.stdin.read()); h=d['host_timeline']; print(d['value'], d['ms_per_step'], d['pipeline']['blocking_cpi_ms'], 'collect', h['collect_wait_ms'], 'enqueue', h['enqueue_ms'], 'n>1ms', h['enqueue_over_1ms'], 'totals', h['enqueue_total_ms'], h['collect_total_ms'])"; }
uptime > $OUT/host.txt
for i in 1 2; do echo "plain 100 #$i: $($B 2>/dev/null | tail -1 | val)"; done | tee -a $OUT/sweep.txt
for i in 1 2; do echo "plain driver #$i: $($B --steps 20 --warmup 5 2>/dev/null | tail -1 | val)"; done | tee -a $OUT/sweep.txt
for i in 1 2; do echo "HSA_ENABLE_INTERRUPT=0 100 #$i: $(HSA_ENABLE_INTERRUPT=0 $B 2>/dev/null | tail -1 | val)"; done | tee -a $OUT/sweep.txt
echo "HSA_ENABLE_INTERRUPT=0 driver: $(HSA_ENABLE_INTERRUPT=0 $B --steps 20 --warmup 5 2>/dev/null | tail -1 | val)" | tee -a $OUT/sweep.txt
echo "nice -20 100: $(nice -n -20 $B 2>/dev/null | tail -1 | val)" | tee -a $OUT/sweep.txt
echo "chrt fifo 100: $(chrt -f 50 $B 2>/dev/null | tail -1 | val)" | tee -a $OUT/sweep.txt
echo "chrt fifo + noirq 100: $(HSA_ENABLE_INTERRUPT=0 chrt -f 50 $B 2>/dev/null | tail -1 | val)" | tee -a $OUT/sweep.txt
echo "taskset 0-7 100: $(taskset -c 0-7 $B 2>/dev/null | tail -1 | val)" | tee -a $OUT/sweep.txt
echo "blocking noirq: $(HSA_ENABLE_INTERRUPT=0 $B --inflight 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | val)" | tee -a $OUT/sweep.txt
echo "1000 steps: $($B --steps 1000 2>/dev/null | tail -1 | val)" | tee -a $OUT/sweep.txt
