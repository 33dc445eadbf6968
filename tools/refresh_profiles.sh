#!/bin/bash
# Regenerates the round's files under profiles/ on a GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh r03'
# Outputs go to gpurun_out/profiles_<tag>/ ; copy them into profiles/ afterwards and run `python tools/profiles_readme.py <tag>`
# (profiles/README.md's headline numbers are generated from the CSV / JSON files, never typed).
set -u
TAG=${1:-r03}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
PS="python $ROOT/tools/prof_summary.py"
db() { find "$1" -name "*.db" | head -1; }
B="python $ROOT/bench.py"

# ---- bench lines: the driver's invocation, the longer run, and the variants
$B --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver_invocation.json 2> $OUT/bench.err
$B --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_driver_invocation_2.json
$B --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_default_100steps.json
$B --no-cpu-baseline --inflight 1 2>/dev/null | tail -1 > $OUT/${TAG}_bench_blocking.json
$B --no-cpu-baseline --cells-per-gpu 7 --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/${TAG}_bench_7cells_per_gpu.json
$B --no-cpu-baseline --ants 256 --inflight 1 --steps 5 --warmup 1 2>/dev/null | tail -1 > $OUT/${TAG}_bench_a256_blocking.json
$B --no-cpu-baseline --ants 256 --inflight 3 --steps 12 --warmup 3 2>/dev/null | tail -1 > $OUT/${TAG}_bench_a256.json
$B --no-cpu-baseline --ants 16 2>/dev/null | tail -1 > $OUT/${TAG}_bench_a16.json
ISAC_MUSIC_FULL_EIG=1 $B --no-cpu-baseline --inflight 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${TAG}_bench_blocking_full_eig.json
ISAC_TAIL_UNFUSED=1 $B --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_default_100steps_old_cfar.json

# ---- rocprofv3: kernel trace of the blocking call sequence on one stream, then of a long pipelined run with nothing but the timed loop
rm -rf /tmp/p1 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats -d /tmp/p1 -- $B --steps 20 --warmup 5 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
$PS $(db /tmp/p1) --csv $OUT/${TAG}_kernel_stats_single_stream.csv > $OUT/${TAG}_kernel_stats_single_stream.txt
rm -rf /tmp/p2 && rocprofv3 --kernel-trace --stats -d /tmp/p2 -- $B --steps 300 --warmup 5 --no-cpu-baseline --trace-only > $OUT/${TAG}_bench_traced_pipelined.json 2>/dev/null
$PS $(db /tmp/p2) --csv $OUT/${TAG}_kernel_stats_pipelined.csv > $OUT/${TAG}_kernel_stats_pipelined.txt
$PS $(db /tmp/p2) --overlap > $OUT/${TAG}_pipeline_overlap.txt
$PS $(db /tmp/p2) --gaps > $OUT/${TAG}_pipeline_gaps.txt
rm -rf /tmp/p5 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats -d /tmp/p5 -- $B --ants 256 --steps 4 --warmup 1 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
$PS $(db /tmp/p5) --csv $OUT/${TAG}_kernel_stats_single_stream_a256.csv > $OUT/${TAG}_kernel_stats_single_stream_a256.txt
# ---- PMC passes (each in its own run, --kernel-trace only)
pmc() { # name counters...
  local name=$1; shift
  rm -rf /tmp/p3 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/p3 -- $B --steps 3 --warmup 1 --inflight 1 --prime-ms 0 --no-cpu-baseline > /dev/null 2>&1
  $PS $(db /tmp/p3) --pmc --csv $OUT/${TAG}_pmc_$name.csv > /dev/null
}
pmc fetch_size FETCH_SIZE
pmc write_size WRITE_SIZE
pmc mfma_busy SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
pmc valu_busy SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
pmc wait_lds SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE

# ---- HIP-event probes
python $ROOT/tools/stage_times.py 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_stage_times_hip_events.txt
python $ROOT/tools/_comm_time.py 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_comm_seam_times.txt
$ROOT/tests/_build/abi_host time 64 3 2>/dev/null | tail -1 > $OUT/${TAG}_abi_host_timing.json
(time python $ROOT/examples/config5.py --cells 21 --ues 10 > $OUT/${TAG}_config5_21x10.json) 2> $OUT/${TAG}_config5_wall.txt
ls -la $OUT
