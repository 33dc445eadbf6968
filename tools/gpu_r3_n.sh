#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/exp_n; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
(timeout 900 python -m pytest tests/test_gpu_spectral.py tests/test_gpu_parity.py tests/test_gpu_tail_fusion.py -m gpu -q -x -n 4 --timeout=800 -p no:cacheprovider | tail -5) > $OUT/tests4.txt 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_driver.json
python bench.py --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | tail -1 > $OUT/bench_100.json
python bench.py --no-cpu-baseline --steps 20 --warmup 5 --inflight 1 2>/dev/null | tail -1 > $OUT/bench_blocking.json
cat $OUT/tests4.txt
python - <<PY
import json
for f in ("bench_driver","bench_100","bench_blocking"):
    try:
        d=json.loads(open("$OUT/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["pipeline"].get("blocking_cpi_ms"), d["roofline"].get("frac"), d["roofline"].get("avg_launch_ms"), d["roofline"].get("whole_cpi"))
    except Exception as e: print(f, "FAILED", e)
PY
