#!/bin/bash
# round-3 GPU call C: fused tail + subspace fixes: tests, bench, gap attribution, queue / tail-stream sweep
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3c; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_music_subspace.py tests/test_gpu_tail_fusion.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -n 6 --timeout=600 -p no:cacheprovider -x > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/rc.txt
tail -8 $OUT/tests.log
B="python $ROOT/bench.py --no-cpu-baseline"
val() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['pipeline']['blocking_cpi_ms'])"; }
echo "driver: $($B --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | tee $OUT/bench_driver.json | val)"
echo "100 steps: $($B 2>/dev/null | tail -1 | tee $OUT/bench_100.json | val)"
echo "blocking: $($B --inflight 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | tee $OUT/bench_blocking.json | val)"
for ts in 0 1; do for q in 16 24 32; do for inf in 8; do
  echo "tail_stream=$ts queues=$q inflight=$inf: $(ISAC_TAIL_STREAM=$ts GPU_MAX_HW_QUEUES=$q $B --inflight $inf 2>/dev/null | tail -1 | val)"
done; done; done | tee $OUT/sweep.txt
echo "unfused tail: $(ISAC_TAIL_UNFUSED=1 $B 2>/dev/null | tail -1 | val)" | tee -a $OUT/sweep.txt
echo "inflight 12 q32 ts1: $(ISAC_TAIL_STREAM=1 GPU_MAX_HW_QUEUES=32 $B --inflight 10 2>/dev/null | tail -1 | val)" | tee -a $OUT/sweep.txt
cd /tmp
PS="python $ROOT/tools/prof_summary.py"
db() { find "$1" -name "*.db" | head -1; }
B2="python $ROOT/bench.py"
rm -rf /tmp/p1 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats -d /tmp/p1 -- $B2 --steps 20 --warmup 5 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
$PS $(db /tmp/p1) --csv $OUT/kernel_stats_single_stream.csv > $OUT/kernel_stats_single_stream.txt
for ts in 0 1; do
rm -rf /tmp/p2 && ISAC_TAIL_STREAM=$ts GPU_MAX_HW_QUEUES=$((16 + 8 * ts)) rocprofv3 --kernel-trace --stats -d /tmp/p2 -- $B2 --steps 60 --warmup 5 --no-cpu-baseline > $OUT/traced_bench_ts$ts.json 2>/dev/null
$PS $(db /tmp/p2) --csv $OUT/kernel_stats_pipelined_ts$ts.csv > $OUT/kernel_stats_pipelined_ts$ts.txt
$PS $(db /tmp/p2) --overlap > $OUT/pipeline_overlap_ts$ts.txt
$PS $(db /tmp/p2) --gaps > $OUT/pipeline_gaps_ts$ts.txt
done
head -22 $OUT/kernel_stats_single_stream.txt; cat $OUT/pipeline_overlap_ts0.txt | head -3; cat $OUT/pipeline_gaps_ts0.txt; cat $OUT/pipeline_overlap_ts1.txt | head -3; cat $OUT/pipeline_gaps_ts1.txt
for ts in 0 1; do tail -1 $OUT/traced_bench_ts$ts.json | val; done
