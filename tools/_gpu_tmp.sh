#!/bin/bash
set -u
TAG=r03
ROOT=$(pwd); OUT=$ROOT/gpurun_out/profiles_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
PS="python $ROOT/tools/prof_summary.py"
db() { find "$1" -name "*.db" | head -1; }
B="python $ROOT/bench.py"
$B --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver_invocation.json 2> $OUT/bench.err
$B --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_driver_invocation_2.json
$B --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_default_100steps.json
$B --no-cpu-baseline --inflight 1 2>/dev/null | tail -1 > $OUT/${TAG}_bench_blocking.json
rm -rf /tmp/p1 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats -d /tmp/p1 -- $B --steps 20 --warmup 5 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
$PS $(db /tmp/p1) --csv $OUT/${TAG}_kernel_stats_single_stream.csv > $OUT/${TAG}_kernel_stats_single_stream.txt
pmc() { local name=$1; shift
  rm -rf /tmp/p3 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/p3 -- $B --steps 3 --warmup 1 --inflight 1 --prime-ms 0 --no-cpu-baseline > /dev/null 2>&1
  $PS $(db /tmp/p3) --pmc --csv $OUT/${TAG}_pmc_$name.csv > /dev/null
}
pmc wait_lds SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pmc valu_busy SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
head -4 $OUT/${TAG}_kernel_stats_single_stream.txt
