#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/exp_a256; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
PS="python $ROOT/tools/prof_summary.py"
db() { find "$1" -name "*.db" | head -1; }
B="python $ROOT/bench.py"
pmc() { local name=$1; shift
  rm -rf /tmp/p3 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/p3 -- $B --ants 256 --steps 2 --warmup 1 --inflight 1 --prime-ms 0 --no-cpu-baseline > /dev/null 2>&1
  $PS $(db /tmp/p3) --pmc --csv $OUT/a256_pmc_$name.csv > /dev/null
  grep "cov_mfma_block\|eigh_tridiag_kernel" $OUT/a256_pmc_$name.csv
}
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
pmc wait SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pmc valu SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY
