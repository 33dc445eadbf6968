#!/bin/bash
# round-3 final GPU call: profiles/r03_* (tools/refresh_profiles.sh), the whole GPU suite (serial, as the driver runs it), a fuzz campaign
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/profiles_r03; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
bash tools/refresh_profiles.sh r03 > $OUT/refresh.log 2>&1
cd $ROOT
(time timeout 1500 python -m pytest tests -m gpu -q --timeout=1200 -p no:cacheprovider) > $OUT/suite_full.log 2>&1
tail -4 $OUT/suite_full.log > $OUT/r03_gpu_test_suite.txt
(ISAC_FUZZ_N=240 timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 6 --timeout=600 -p no:cacheprovider | tail -3) > $OUT/r03_fuzz_campaigns.txt 2>&1
cat $OUT/r03_gpu_test_suite.txt $OUT/r03_fuzz_campaigns.txt
python - <<PY
import json
for f in ("bench_driver_invocation","bench_driver_invocation_2","bench_default_100steps","bench_blocking","bench_a256","bench_a256_blocking","bench_7cells_per_gpu"):
    try:
        d=[json.loads(l) for l in open("$OUT/r03_%s.json"%f) if l.startswith("{")][-1]; print(f, d["value"], d["ms_per_step"], d["pipeline"]["blocking_cpi_ms"], d["roofline"].get("frac"))
    except Exception as e: print(f, "FAILED", e)
PY
head -22 $OUT/r03_kernel_stats_single_stream.txt
