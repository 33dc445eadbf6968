#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/exp_n; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
(timeout 900 python -m pytest tests/test_gpu_music_subspace.py tests/test_gpu_parity.py -m gpu -q -x -n 4 --timeout=800 -p no:cacheprovider | tail -3) 2>&1
for i in 1 2; do python bench.py --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['pipeline'].get('blocking_cpi_ms'))"; done
