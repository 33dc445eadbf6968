#!/bin/bash
# round-3 GPU call A: new eigensolver route tests, the whole GPU suite, bench lines (driver invocation, blocking, A=256)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3a; mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/host.txt; free -g >> $OUT/host.txt
timeout 600 python -m pytest tests/test_gpu_music_subspace.py -q -x --timeout=300 > $OUT/subspace.log 2>&1; echo "subspace rc=$?" >> $OUT/rc.txt
tail -15 $OUT/subspace.log
B="python bench.py --no-cpu-baseline"
timeout 300 $B --gpus 1 --steps 20 --warmup 5 2>$OUT/bench_driver.err | tail -1 > $OUT/bench_driver.json
timeout 300 $B --inflight 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_blocking.json
timeout 300 $B --ants 256 --inflight 1 --steps 5 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_a256.json
timeout 300 $B --ants 256 --inflight 3 --steps 9 --warmup 3 2>/dev/null | tail -1 > $OUT/bench_a256_inflight3.json
for f in bench_driver bench_blocking bench_a256 bench_a256_inflight3; do python - <<PY
import json
try:
    d=json.load(open("$OUT/$f.json")); print("$f", d["value"], d["ms_per_step"], d["pipeline"]["blocking_cpi_ms"], d["roofline"].get("frac"))
except Exception as e: print("$f FAILED", e)
PY
done
timeout 1500 python -m pytest tests -m gpu -q -n 4 --timeout=1200 -p no:cacheprovider > $OUT/suite.log 2>&1; echo "suite rc=$?" >> $OUT/rc.txt
tail -40 $OUT/suite.log
