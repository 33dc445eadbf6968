#!/bin/bash
# the whole GPU suite (serial, as the driver runs it) + the 240-seed fuzz campaign on the current tree -> gpurun_out/profiles_r03/
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/profiles_r03; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
(time timeout 1500 python -m pytest tests -m gpu -q --timeout=1200 -p no:cacheprovider) > $OUT/suite_full.log 2>&1
tail -4 $OUT/suite_full.log > $OUT/r03_gpu_test_suite.txt
(ISAC_FUZZ_N=240 timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 6 --timeout=600 -p no:cacheprovider | tail -3) > $OUT/r03_fuzz_campaigns.txt 2>&1
cat $OUT/r03_gpu_test_suite.txt $OUT/r03_fuzz_campaigns.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/r03_bench_driver_invocation.json
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r03_bench_driver_invocation_2.json
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r03_bench_default_100steps.json
python bench.py --no-cpu-baseline --schedule ordered 2>/dev/null | tail -1 > $OUT/r03_bench_default_100steps_ordered.json
python bench.py --no-cpu-baseline --inflight 1 2>/dev/null | tail -1 > $OUT/r03_bench_blocking.json
python - <<PY
import json
for f in ("bench_driver_invocation","bench_driver_invocation_2","bench_default_100steps","bench_default_100steps_ordered","bench_blocking"):
    try:
        d=[json.loads(l) for l in open("$OUT/r03_%s.json"%f) if l.startswith("{")][-1]; print(f, d["value"], d["ms_per_step"], d["pipeline"]["blocking_cpi_ms"], d["roofline"].get("frac"))
    except Exception as e: print(f, "FAILED", e)
PY
