#!/bin/bash
# last GPU call of round 3: the A = 64 files of profiles/r03_* on the final tree (tools/refresh_profiles.sh minus the A = 256 / config-5 / seam legs,
# which the final kernels changes did not touch), then the whole GPU suite (serial, as the driver runs it)
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
$B --no-cpu-baseline --schedule ordered 2>/dev/null | tail -1 > $OUT/${TAG}_bench_default_100steps_ordered.json
$B --no-cpu-baseline --inflight 1 2>/dev/null | tail -1 > $OUT/${TAG}_bench_blocking.json
$B --no-cpu-baseline --cells-per-gpu 7 --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/${TAG}_bench_7cells_per_gpu.json
$B --no-cpu-baseline --ants 16 2>/dev/null | tail -1 > $OUT/${TAG}_bench_a16.json
rm -rf /tmp/p1 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats -d /tmp/p1 -- $B --steps 20 --warmup 5 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
$PS $(db /tmp/p1) --csv $OUT/${TAG}_kernel_stats_single_stream.csv > $OUT/${TAG}_kernel_stats_single_stream.txt
rm -rf /tmp/p2 && rocprofv3 --kernel-trace --stats -d /tmp/p2 -- $B --steps 300 --warmup 5 --no-cpu-baseline --trace-only > $OUT/${TAG}_bench_traced_pipelined.json 2>/dev/null
$PS $(db /tmp/p2) --csv $OUT/${TAG}_kernel_stats_pipelined.csv > $OUT/${TAG}_kernel_stats_pipelined.txt
$PS $(db /tmp/p2) --overlap > $OUT/${TAG}_pipeline_overlap.txt
$PS $(db /tmp/p2) --gaps > $OUT/${TAG}_pipeline_gaps.txt
pmc() { local name=$1; shift
  rm -rf /tmp/p3 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/p3 -- $B --steps 3 --warmup 1 --inflight 1 --prime-ms 0 --no-cpu-baseline > /dev/null 2>&1
  $PS $(db /tmp/p3) --pmc --csv $OUT/${TAG}_pmc_$name.csv > /dev/null
}
pmc fetch_size FETCH_SIZE
pmc write_size WRITE_SIZE
pmc mfma_busy SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
pmc valu_busy SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
pmc wait_lds SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
python $ROOT/tools/stage_times.py 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_stage_times_hip_events.txt
head -8 $OUT/${TAG}_kernel_stats_single_stream.txt
cd $ROOT
(time timeout 1200 python -m pytest tests -m gpu -q --timeout=1000 -p no:cacheprovider) > $OUT/suite_full.log 2>&1
grep "passed\|failed\|^real" $OUT/suite_full.log > $OUT/${TAG}_gpu_test_suite.txt
cat $OUT/${TAG}_gpu_test_suite.txt
