#!/bin/bash
# Regenerates every file under profiles/ on a GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh r02'
# Outputs go to gpurun_out/profiles_<tag>/ ; copy them into profiles/ afterwards.
set -u
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
PS="python $ROOT/tools/prof_summary.py"
db() { find "$1" -name "*.db" | head -1; }
B="python $ROOT/bench.py"

# ---- bench lines: the driver's invocation, the builder's longer run, and the variants
$B --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver_invocation.json 2> $OUT/bench.err
$B --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_default_100steps.json
$B --no-cpu-baseline --no-fuse 2>/dev/null | tail -1 > $OUT/${TAG}_bench_unfused.json
$B --no-cpu-baseline --no-fuse --noise-domain time 2>/dev/null | tail -1 > $OUT/${TAG}_bench_time_domain_noise.json
$B --no-cpu-baseline --inflight 1 2>/dev/null | tail -1 > $OUT/${TAG}_bench_blocking.json
$B --no-cpu-baseline --inflight 1 --no-fuse 2>/dev/null | tail -1 > $OUT/${TAG}_bench_blocking_unfused.json
$B --no-cpu-baseline --cells-per-gpu 7 --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/${TAG}_bench_7cells_per_gpu.json
$B --no-cpu-baseline --ants 256 --inflight 1 --steps 5 --warmup 1 2>/dev/null | tail -1 > $OUT/${TAG}_bench_a256.json
$B --no-cpu-baseline --ants 16 2>/dev/null | tail -1 > $OUT/${TAG}_bench_a16.json
(for pm in 0 300; do for w in 5 100; do echo "prime_ms=$pm warmup=$w steps=20: $($B --gpus 1 --steps 20 --warmup $w --prime-ms $pm --no-cpu-baseline 2>/dev/null | tail -1 | cut -c60-130)"; done; done) > $OUT/${TAG}_warmup_sensitivity.txt

# ---- rocprofv3: kernel trace of the blocking call sequence on one stream, then of the default pipelined run
# (default priming: the kernels are timed at the sustained clocks the bench's own HIP-event figure is taken at; an idle-started run reads ~8 % slower)
rm -rf /tmp/p1 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats -d /tmp/p1 -- $B --steps 20 --warmup 5 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
$PS $(db /tmp/p1) --csv $OUT/${TAG}_kernel_stats_single_stream.csv > $OUT/${TAG}_kernel_stats_single_stream.txt
rm -rf /tmp/p2 && rocprofv3 --kernel-trace --stats -d /tmp/p2 -- $B --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
$PS $(db /tmp/p2) --csv $OUT/${TAG}_kernel_stats_pipelined.csv > $OUT/${TAG}_kernel_stats_pipelined.txt
$PS $(db /tmp/p2) --overlap > $OUT/${TAG}_pipeline_overlap.txt
# ---- PMC passes (each in its own run, --kernel-trace only)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p3 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc $c -d /tmp/p3 -- $B --steps 3 --warmup 1 --inflight 1 --prime-ms 0 --no-cpu-baseline > /dev/null 2>&1
  n=$(echo $c | tr A-Z a-z)
  $PS $(db /tmp/p3) --pmc --csv $OUT/${TAG}_pmc_$n.csv > /dev/null
done
rm -rf /tmp/p4 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d /tmp/p4 -- $B --steps 3 --warmup 1 --inflight 1 --prime-ms 0 --no-cpu-baseline > /dev/null 2>&1
$PS $(db /tmp/p4) --pmc --csv $OUT/${TAG}_pmc_mfma_busy.csv > /dev/null
rm -rf /tmp/p6 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -d /tmp/p6 -- $B --steps 3 --warmup 1 --inflight 1 --prime-ms 0 --no-cpu-baseline > /dev/null 2>&1
$PS $(db /tmp/p6) --pmc --csv $OUT/${TAG}_pmc_valu_busy.csv > /dev/null

# ---- HIP-event probes
python $ROOT/tools/stage_times.py 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_stage_times_hip_events.txt
python $ROOT/tools/_comm_time.py 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_comm_seam_times.txt
$ROOT/tests/_build/abi_host time 64 3 2>/dev/null | tail -1 > $OUT/${TAG}_abi_host_timing.json
(time python $ROOT/examples/config5.py --cells 21 --ues 10 > $OUT/${TAG}_config5_21x10.json) 2> $OUT/${TAG}_config5_wall.txt
ISAC_CPU_DEBUG=1 $B --steps 2 --warmup 1 --prime-ms 0 2>&1 >/dev/null | grep isac_cpu | tail -12 > $OUT/${TAG}_cpu_port_stage_times.txt
ls -la $OUT
