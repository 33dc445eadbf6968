#!/bin/bash
# Regenerates every file under profiles/ on a GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh r01'
# Outputs go to gpurun_out/profiles_<tag>/ ; copy them into profiles/ afterwards.
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
PS="python $ROOT/tools/prof_summary.py"
db() { find "$1" -name "*.db" | head -1; }

python $ROOT/bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/bench.err
python $ROOT/bench.py --no-cpu-baseline --fuse 2>/dev/null | tail -1 > $OUT/${TAG}_bench_fused.json
python $ROOT/bench.py --no-cpu-baseline --inflight 1 2>/dev/null | tail -1 > $OUT/${TAG}_bench_blocking.json
python $ROOT/bench.py --no-cpu-baseline --cells-per-gpu 7 --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/${TAG}_bench_7cells.json
python $ROOT/bench.py --no-cpu-baseline --ants 256 --inflight 1 --steps 5 --warmup 1 2>/dev/null | tail -1 > $OUT/${TAG}_bench_a256.json

rm -rf /tmp/p1 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace -d /tmp/p1 -- python $ROOT/bench.py --steps 10 --warmup 2 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
$PS $(db /tmp/p1) --csv $OUT/${TAG}_kernel_stats_single_stream.csv > $OUT/${TAG}_kernel_stats_single_stream.txt
rm -rf /tmp/p2 && rocprofv3 --kernel-trace -d /tmp/p2 -- python $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
$PS $(db /tmp/p2) --csv $OUT/${TAG}_kernel_stats_pipelined.csv > /dev/null
$PS $(db /tmp/p2) --overlap > $OUT/${TAG}_pipeline_overlap.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p3 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc $c -d /tmp/p3 -- python $ROOT/bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
  n=$(echo $c | tr A-Z a-z)
  $PS $(db /tmp/p3) --pmc --csv $OUT/${TAG}_pmc_$n.csv > /dev/null
done
rm -rf /tmp/p4 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d /tmp/p4 -- python $ROOT/bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
$PS $(db /tmp/p4) --pmc --csv $OUT/${TAG}_pmc_mfma_busy.csv > /dev/null

rm -rf /tmp/p6 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -d /tmp/p6 -- python $ROOT/bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
$PS $(db /tmp/p6) --pmc --csv $OUT/${TAG}_pmc_valu_busy.csv > /dev/null

python $ROOT/tools/stage_times.py 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_stage_times_hip_events.txt
for t in kbench dbench nbench latbench mbench; do
  [ -x $ROOT/tools/$t ] && $ROOT/tools/$t > $OUT/${TAG}_${t}.txt 2>&1
done
python $ROOT/tools/_overlap_probe.py 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_stage_pair_overlap.txt
cd $ROOT; ISAC_DEBUG=1 python $ROOT/tools/_eig_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_eig_probe.txt
ISAC_DEBUG=1 python $ROOT/tools/_eig_big_probe.py 2>&1 | grep -v amdgpu.ids >> $OUT/${TAG}_eig_probe.txt
ls -la $OUT
