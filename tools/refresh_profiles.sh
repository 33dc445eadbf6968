#!/bin/bash
# Regenerates the round's files under profiles/ on a GPU box (run through gpurun from the repo root):
#   gpurun --timeout 2400 -- 'bash tools/refresh_profiles.sh r05'
# Outputs go to gpurun_out/profiles_<tag>/ ; copy them into profiles/ afterwards and run `python tools/profiles_readme.py <tag>`
# (profiles/README.md's headline numbers are generated from the CSV / JSON files, never typed).
set -u
TAG=${1:-r05}
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
$B --no-cpu-baseline --ants 256 --inflight 6 --steps 24 --warmup 6 2>/dev/null | tail -1 > $OUT/${TAG}_bench_a256_inflight6.json
$B --no-cpu-baseline --ants 16 2>/dev/null | tail -1 > $OUT/${TAG}_bench_a16.json
ISAC_MUSIC_FULL_EIG=1 $B --no-cpu-baseline --inflight 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${TAG}_bench_blocking_full_eig.json
$B --no-cpu-baseline --schedule ordered 2>/dev/null | tail -1 > $OUT/${TAG}_bench_default_100steps_ordered.json
# warm-up sensitivity: the driver's invocation without the untimed priming phase
{ echo "# python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline [--prime-ms 0]: value (slots/s), ms per CPI, untimed priming steps"; for pm in 300 0 300 0; do
  $B --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --prime-ms $pm 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('--prime-ms $pm:', d['value'], d['ms_per_step'], d['priming']['untimed_steps_before_warmup'])"; done; } > $OUT/${TAG}_warmup_sensitivity.txt
# BASELINE configs[4]: 21 cells x 10 UE on this one GPU -- as the reference steps it (UL slots, precoded PDSCH input, per-occasion device CSI), then the round-4
# workload shape for comparison (no 'U' slots, CSI estimate evaluated once on the host), then that shape through the unfused CDL kernels (Z through HBM)
$B --workload config5 2>/dev/null | tail -1 > $OUT/${TAG}_bench_config5_21x10.json
ISAC_C5_NO_UL=1 $B --workload config5 2>/dev/null | tail -1 > $OUT/${TAG}_bench_config5_21x10_no_ul.json
ISAC_C5_NO_UL=1 ISAC_C5_HOST_CSI=1 $B --workload config5 2>/dev/null | tail -1 > $OUT/${TAG}_bench_config5_21x10_r04_workload.json
ISAC_C5_NO_UL=1 ISAC_C5_HOST_CSI=1 ISAC_CDL_UNFUSED=1 $B --workload config5 2>/dev/null | tail -1 > $OUT/${TAG}_bench_config5_21x10_r04_workload_unfused_cdl.json

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

pmc_a256() { local name=$1; shift
  rm -rf /tmp/p3 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/p3 -- $B --ants 256 --steps 2 --warmup 1 --inflight 1 --prime-ms 0 --no-cpu-baseline > /dev/null 2>&1
  $PS $(db /tmp/p3) --pmc --csv $OUT/${TAG}_pmc_a256_$name.csv > /dev/null
}
pmc_a256 fetch_size FETCH_SIZE
pmc_a256 mfma_busy SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
pmc_a256 wait_lds SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE

# ---- HIP-event probes
python $ROOT/tools/stage_times.py 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_stage_times_hip_events.txt
python $ROOT/tools/comm_probe.py 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_comm_seam_times.txt
{ echo "# the same probe through the unfused kernels (ISAC_CDL_UNFUSED=1: cdl_pack + cdl_gemm + cdl_fir4, Z through HBM), same box"; ISAC_CDL_UNFUSED=1 python $ROOT/tools/comm_probe.py --batch-only 2>/dev/null | grep -v amdgpu.ids; } >> $OUT/${TAG}_comm_seam_times.txt
# PMC counters of the CDL kernels (fused and unfused), batch of 10 DL jobs: MFMA busy, waits, bytes
cdlpmc() { # tag env counters...
  local tag=$1 envs=$2; shift 2
  rm -rf /tmp/p8 && env $envs rocprofv3 --kernel-trace --pmc "$@" -d /tmp/p8 -- python $ROOT/tools/comm_probe.py --batch-only --reps 3 > /dev/null 2>&1
  { echo "## $tag: $*"; $PS $(db /tmp/p8) --pmc | grep -E "cdl_|kernel,|^kernel" ; } >> $OUT/${TAG}_pmc_cdl_kernels.txt
}
: > $OUT/${TAG}_pmc_cdl_kernels.txt
cdlpmc fused X=1 FETCH_SIZE
cdlpmc fused X=1 WRITE_SIZE
cdlpmc fused X=1 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
cdlpmc fused X=1 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
cdlpmc unfused ISAC_CDL_UNFUSED=1 FETCH_SIZE
cdlpmc unfused ISAC_CDL_UNFUSED=1 WRITE_SIZE
rm -rf /tmp/p8 && rocprofv3 --kernel-trace --stats -d /tmp/p8 -- python $ROOT/tools/comm_probe.py --batch-only --reps 5 > /dev/null 2>&1
$PS $(db /tmp/p8) > $OUT/${TAG}_kernel_stats_cdl_batch.txt
# cold start: what a one-call-per-cell host sees (fresh processes)
{ for m in noreserve reserve; do $B --cold-probe $m 2>/dev/null | tail -1; done; $B --cold-probe reserve --cold-warm-ms 0 2>/dev/null | tail -1; } > $OUT/${TAG}_cold_probe_lines.json
$ROOT/tests/_build/abi_host time 64 3 2>/dev/null | tail -1 > $OUT/${TAG}_abi_host_timing.json
# config 5 under rocprofv3: per-kernel stats of one frame of 21 cells (CDL / CSI kernels beside the sensing kernels)
rm -rf /tmp/p9 && rocprofv3 --kernel-trace --stats -d /tmp/p9 -- $B --workload config5 --steps 1 --warmup 1 > /dev/null 2>&1
$PS $(db /tmp/p9) --csv $OUT/${TAG}_kernel_stats_config5.csv > $OUT/${TAG}_kernel_stats_config5.txt
python $ROOT/tools/cov_probe.py 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_cov_probe.txt; python $ROOT/tools/cov_probe.py --ants 256 --reps 10 2>/dev/null | grep -v amdgpu.ids >> $OUT/${TAG}_cov_probe.txt
(cd $ROOT && bash tools/tridiag_dist_probe.sh > /dev/null 2>&1; cp gpurun_out/tridiag_dist_probe.txt $OUT/${TAG}_tridiag_dist_probe.txt)
ls -la $OUT
