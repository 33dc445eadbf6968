#!/bin/bash
# round-3 GPU call B: eigh_top tests again, rocprofv3 single-stream stats + pipelined overlap with the subspace eigensolver
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3b; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_music_subspace.py -q -n 4 --timeout=300 -p no:cacheprovider > $OUT/subspace.log 2>&1; echo "subspace rc=$?" >> $OUT/rc.txt
tail -5 $OUT/subspace.log
cd /tmp
PS="python $ROOT/tools/prof_summary.py"
db() { find "$1" -name "*.db" | head -1; }
B="python $ROOT/bench.py"
rm -rf /tmp/p1 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats -d /tmp/p1 -- $B --steps 20 --warmup 5 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
$PS $(db /tmp/p1) --csv $OUT/kernel_stats_single_stream.csv > $OUT/kernel_stats_single_stream.txt
rm -rf /tmp/p2 && rocprofv3 --kernel-trace --stats -d /tmp/p2 -- $B --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
$PS $(db /tmp/p2) --csv $OUT/kernel_stats_pipelined.csv > $OUT/kernel_stats_pipelined.txt
$PS $(db /tmp/p2) --overlap > $OUT/pipeline_overlap.txt
rm -rf /tmp/p3 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats -d /tmp/p3 -- $B --ants 256 --steps 4 --warmup 1 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
$PS $(db /tmp/p3) > $OUT/kernel_stats_a256.txt
python $ROOT/tools/_host_overhead.py > $OUT/host_overhead.txt 2>&1
head -30 $OUT/kernel_stats_single_stream.txt; cat $OUT/pipeline_overlap.txt; head -12 $OUT/kernel_stats_a256.txt; tail -5 $OUT/host_overhead.txt
