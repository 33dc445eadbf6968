#!/bin/bash
# refresh of the A = 256 files of profiles/r03_* only (after a change that touches nothing but cov_mfma_block_kernel)
set -u
TAG=r03
ROOT=$(pwd); OUT=$ROOT/gpurun_out/profiles_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
PS="python $ROOT/tools/prof_summary.py"
db() { find "$1" -name "*.db" | head -1; }
B="python $ROOT/bench.py"
$B --no-cpu-baseline --ants 256 --inflight 1 --steps 5 --warmup 1 2>/dev/null | tail -1 > $OUT/${TAG}_bench_a256_blocking.json
$B --no-cpu-baseline --ants 256 --inflight 3 --steps 12 --warmup 3 2>/dev/null | tail -1 > $OUT/${TAG}_bench_a256.json
rm -rf /tmp/p5 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats -d /tmp/p5 -- $B --ants 256 --steps 4 --warmup 1 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
$PS $(db /tmp/p5) --csv $OUT/${TAG}_kernel_stats_single_stream_a256.csv > $OUT/${TAG}_kernel_stats_single_stream_a256.txt
pmc() { local name=$1; shift
  rm -rf /tmp/p3 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/p3 -- $B --ants 256 --steps 2 --warmup 1 --inflight 1 --prime-ms 0 --no-cpu-baseline > /dev/null 2>&1
  $PS $(db /tmp/p3) --pmc --csv $OUT/${TAG}_pmc_a256_$name.csv > /dev/null
}
if [ "${A256_PMC:-1}" = "1" ]; then
pmc fetch_size FETCH_SIZE
pmc mfma_busy SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
pmc wait_lds SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
fi
head -4 $OUT/${TAG}_kernel_stats_single_stream_a256.txt
