#!/bin/bash
# experiment call: echo_range_kernel straight-line variants + the reworked covariance block kernel
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/exp_n; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -n 4 --timeout=500 -p no:cacheprovider -k "cov or covariance or a256 or 256" | tail -5) > $OUT/cov_tests.txt 2>&1
for v in 0 1 2 3 4 5 0; do ISAC_ER_VARIANT=$v timeout 300 python tools/_er_variant_probe.py; done > $OUT/er_variants.txt 2>&1
for v in 0 3 5; do ER_TARGETS=2 ISAC_ER_VARIANT=$v timeout 300 python tools/_er_variant_probe.py; done >> $OUT/er_variants.txt 2>&1
(timeout 300 python bench.py --no-cpu-baseline --ants 256 --inflight 1 --steps 5 --warmup 1 2>/dev/null | tail -1) > $OUT/bench_a256_blocking.json
(timeout 300 python bench.py --no-cpu-baseline --ants 256 --inflight 3 --steps 12 --warmup 3 2>/dev/null | tail -1) > $OUT/bench_a256.json
cat $OUT/cov_tests.txt $OUT/er_variants.txt
python - <<PY
import json
for f in ("bench_a256_blocking","bench_a256"):
    try:
        d=json.loads(open("$OUT/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["pipeline"].get("blocking_cpi_ms"), d.get("stage_ms"))
    except Exception as e: print(f, "FAILED", e)
PY
