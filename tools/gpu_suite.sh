#!/bin/bash
# the whole GPU suite (serial, as the driver runs it) + smoke + the driver's bench invocation on the current tree -> gpurun_out/suite_<tag>/
set -u
TAG=${1:-r04}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/suite_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
(time timeout 1700 python -m pytest tests -m gpu -q --timeout=1200 -p no:cacheprovider) > $OUT/suite_full.log 2>&1
tail -5 $OUT/suite_full.log > $OUT/${TAG}_gpu_test_suite.txt
cat $OUT/${TAG}_gpu_test_suite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${TAG}_bench_driver_invocation.json
python - <<PY
import json
d=[json.loads(l) for l in open("$OUT/${TAG}_bench_driver_invocation.json") if l.startswith("{")][-1]; print(d["value"], d["ms_per_step"], d["pipeline"]["blocking_cpi_ms"], d["roofline"].get("frac"), d["cpu_baseline"]["value"])
PY
