#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/exp_n; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
PS="python tools/prof_summary.py"
db() { find $1 -name "*.db" | head -1; }
(ISAC_COV_RR=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -n 4 --timeout=500 -p no:cacheprovider -k "cov or covariance" | tail -3) > $OUT/cov_rr_tests.txt 2>&1
for mode in "0 0" "1 0" "0 1" "1 1"; do set -- $mode
  export ISAC_COV_RR=$1 ISAC_ER_PLAIN_STORE=$2
  rm -rf /tmp/pp && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats -d /tmp/pp -- python bench.py --steps 20 --warmup 5 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
  echo "== RR=$1 PLAIN=$2 single stream"; $PS $(db /tmp/pp) | head -4
  echo "== RR=$1 PLAIN=$2 pipelined 100 steps"; python bench.py --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done > $OUT/cov_rr.txt 2>&1
cat $OUT/cov_rr_tests.txt $OUT/cov_rr.txt
