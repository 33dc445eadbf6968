#!/bin/bash
# round-3 GPU call F: fence-free two-kernel panel CFAR; cov block XCD mapping A/B
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3f; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_music_subspace.py tests/test_gpu_tail_fusion.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -n 6 --timeout=600 -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/rc.txt
tail -4 $OUT/tests.log
B="python $ROOT/bench.py --no-cpu-baseline"
val() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['pipeline']['blocking_cpi_ms'])"; }
cov() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], [s['ms'] for s in d['roofline']['other_stages'] if 'covariance' in s['stage']])"; }
echo "driver: $($B --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | tee $OUT/bench_driver.json | val)"
echo "driver again: $($B --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | val)"
echo "100 steps: $($B 2>/dev/null | tail -1 | tee $OUT/bench_100.json | val)"
echo "blocking: $($B --inflight 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | tee $OUT/bench_blocking.json | val)"
echo "old tail 100: $(ISAC_TAIL_UNFUSED=1 $B 2>/dev/null | tail -1 | val)" | tee -a $OUT/sweep.txt
echo "a256 inflight3 xcd0: $(ISAC_COV_XCD=0 $B --ants 256 --inflight 3 --steps 9 --warmup 3 2>/dev/null | tail -1 | cov)" | tee -a $OUT/sweep.txt
echo "a256 inflight3 xcd1: $(ISAC_COV_XCD=1 $B --ants 256 --inflight 3 --steps 9 --warmup 3 2>/dev/null | tail -1 | cov)" | tee -a $OUT/sweep.txt
echo "a128 xcd0: $(ISAC_COV_XCD=0 $B --ants 128 --inflight 4 --steps 20 --warmup 3 2>/dev/null | tail -1 | cov)" | tee -a $OUT/sweep.txt
echo "a128 xcd1: $(ISAC_COV_XCD=1 $B --ants 128 --inflight 4 --steps 20 --warmup 3 2>/dev/null | tail -1 | cov)" | tee -a $OUT/sweep.txt
cd /tmp
PS="python $ROOT/tools/prof_summary.py"
db() { find "$1" -name "*.db" | head -1; }
B2="python $ROOT/bench.py"
rm -rf /tmp/p1 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats -d /tmp/p1 -- $B2 --steps 20 --warmup 5 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
$PS $(db /tmp/p1) --csv $OUT/kernel_stats_single_stream.csv > $OUT/kernel_stats_single_stream.txt
rm -rf /tmp/p2 && rocprofv3 --kernel-trace --stats -d /tmp/p2 -- $B2 --steps 300 --warmup 5 --no-cpu-baseline --trace-only > $OUT/traced_bench.json 2>/dev/null
$PS $(db /tmp/p2) --csv $OUT/kernel_stats_pipelined.csv > $OUT/kernel_stats_pipelined.txt
$PS $(db /tmp/p2) --overlap > $OUT/pipeline_overlap.txt
$PS $(db /tmp/p2) --gaps > $OUT/pipeline_gaps.txt
head -20 $OUT/kernel_stats_single_stream.txt; head -3 $OUT/pipeline_overlap.txt; head -8 $OUT/pipeline_gaps.txt
tail -1 $OUT/traced_bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('traced', d['value'], d['ms_per_step'])"
