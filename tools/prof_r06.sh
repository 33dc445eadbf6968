#!/bin/bash
# Round-6 profile of the bench's blocking call sequence (one stream), lazy and array echo grid:  gpurun -- 'bash tools/prof_r06.sh [tag] [extra bench args]'
# kernel trace + separate --pmc passes (HBM traffic, MFMA busy, VALU / wait split) -> gpurun_out/prof_<tag>/<tag>_<mode>_*.{txt,csv}
set -u
TAG=${1:-r06}; shift || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
PS="python $ROOT/tools/prof_summary.py"
db() { find "$1" -name "*.db" | head -1; }
for MODE in ${MODES:-lazy array}; do
  B="python $ROOT/bench.py --no-cpu-baseline --no-cold --prime-ms 0 --inflight 1 --echo $MODE $*"
  rm -rf /tmp/q1 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace -d /tmp/q1 -- $B --steps 20 --warmup 3 > /dev/null 2>&1
  $PS $(db /tmp/q1) --csv $OUT/${TAG}_${MODE}_kernel_stats_single_stream.csv > $OUT/${TAG}_${MODE}_kernel_stats_single_stream.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/q3 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc $c -d /tmp/q3 -- $B --steps 3 --warmup 1 > /dev/null 2>&1
    $PS $(db /tmp/q3) --pmc --csv $OUT/${TAG}_${MODE}_pmc_$(echo $c | tr A-Z a-z).csv > /dev/null
  done
  rm -rf /tmp/q5 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d /tmp/q5 -- $B --steps 3 --warmup 1 > /dev/null 2>&1
  $PS $(db /tmp/q5) --pmc --csv $OUT/${TAG}_${MODE}_pmc_mfma.csv > /dev/null
  rm -rf /tmp/q6 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_MFMA -d /tmp/q6 -- $B --steps 3 --warmup 1 > /dev/null 2>&1
  $PS $(db /tmp/q6) --pmc --csv $OUT/${TAG}_${MODE}_pmc_valu.csv > /dev/null
  rm -rf /tmp/q8 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES -d /tmp/q8 -- $B --steps 3 --warmup 1 > /dev/null 2>&1
  $PS $(db /tmp/q8) --pmc --csv $OUT/${TAG}_${MODE}_pmc_wait.csv > /dev/null
  echo "== $MODE"; head -16 $OUT/${TAG}_${MODE}_kernel_stats_single_stream.txt | cut -c1-200
  grep -hE "echo_range|cov_|beamsum" $OUT/${TAG}_${MODE}_pmc_fetch_size.csv $OUT/${TAG}_${MODE}_pmc_write_size.csv $OUT/${TAG}_${MODE}_pmc_mfma.csv $OUT/${TAG}_${MODE}_pmc_valu.csv $OUT/${TAG}_${MODE}_pmc_wait.csv | sed 's/_ZN4isac[0-9]*//; s/EEv[^"]*"/"/' | cut -c1-170
done
