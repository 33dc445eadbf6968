#!/bin/bash
# Development A/B of the covariance kernel's switches:  gpurun -- 'bash tools/cov_ab.sh'
# per variant: pipelined rate (100 steps), isolated covariance stage time, FETCH_SIZE of the covariance kernel (separate --pmc pass)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/cov_ab; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
[ $# -eq 0 ] && set -- 1,-1 0,-1 1,7 1,1 1,0
for v in "$@"; do
  IFS=, read lm bm <<< "$v"
  export ISAC_COV_LINEMAP=$lm ISAC_COV_BARMASK=$bm
  for i in 1 2; do
    python $ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/b.json 2>/dev/null
    python - <<PY
import json
d=json.load(open("$OUT/b.json"))
print("linemap=$lm barmask=$bm", d["value"], "blocking", d["pipeline"]["blocking_cpi_ms"], "cov stage ms", d["roofline"]["other_stages"][-1]["ms"])
PY
  done
  rm -rf /tmp/q3 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/q3 -- python $ROOT/bench.py --no-cpu-baseline --prime-ms 0 --inflight 1 --steps 3 --warmup 1 > /dev/null 2>&1
  python $ROOT/tools/prof_summary.py $(find /tmp/q3 -name "*.db" | head -1) --pmc --csv $OUT/fetch_${lm}_${bm}.csv > /dev/null
  grep -E "cov_mfma" $OUT/fetch_${lm}_${bm}.csv | cut -c1-200
done
