#!/bin/bash
# Quick profile of the bench's blocking call sequence (one stream):  gpurun -- 'bash tools/prof_quick.sh <tag> [bench args...]'
# kernel trace, then separate --pmc passes (HBM traffic, VALU occupancy); summaries under gpurun_out/prof_<tag>/
set -u
TAG=${1:-q}; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
PS="python $ROOT/tools/prof_summary.py"
db() { find "$1" -name "*.db" | head -1; }
B="python $ROOT/bench.py --no-cpu-baseline --prime-ms 0 --inflight 1 $*"
rm -rf /tmp/q1 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace -d /tmp/q1 -- $B --steps 10 --warmup 2 > /dev/null 2>&1
$PS $(db /tmp/q1) --csv $OUT/${TAG}_kernel_stats_single_stream.csv > $OUT/${TAG}_kernel_stats_single_stream.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/q3 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc $c -d /tmp/q3 -- $B --steps 3 --warmup 1 > /dev/null 2>&1
  $PS $(db /tmp/q3) --pmc --csv $OUT/${TAG}_pmc_$(echo $c | tr A-Z a-z).csv > /dev/null
done
rm -rf /tmp/q6 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -d /tmp/q6 -- $B --steps 3 --warmup 1 > /dev/null 2>&1
$PS $(db /tmp/q6) --pmc --csv $OUT/${TAG}_pmc_valu_busy.csv > /dev/null
rm -rf /tmp/q7 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS -d /tmp/q7 -- $B --steps 3 --warmup 1 > /dev/null 2>&1
$PS $(db /tmp/q7) --pmc --csv $OUT/${TAG}_pmc_lds.csv > /dev/null
rm -rf /tmp/q8 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d /tmp/q8 -- $B --steps 3 --warmup 1 > /dev/null 2>&1
$PS $(db /tmp/q8) --pmc --csv $OUT/${TAG}_pmc_wait.csv > /dev/null
grep -E "echo_range|range_kernel" $OUT/${TAG}_pmc_lds.csv $OUT/${TAG}_pmc_wait.csv | sed 's/_ZN4isac//; s/ILi[^"]*kd"/"/; s/INS_[^"]*kd"/"/' | cut -c1-160
head -14 $OUT/${TAG}_kernel_stats_single_stream.txt
grep -E "echo_range|range_kernel|cov_mfma|beamsum" $OUT/${TAG}_pmc_fetch_size.csv $OUT/${TAG}_pmc_write_size.csv | cut -c1-220
grep -E "echo_range" $OUT/${TAG}_pmc_valu_busy.csv | cut -c1-220
