#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3m; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
B="python $ROOT/bench.py --no-cpu-baseline"
val() { python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['pacing']; print(d['value'], d['ms_per_step'], d['pipeline']['blocking_cpi_ms'], p['mode'], p['pace_ms'])"; }
echo "a256 blocking: $($B --ants 256 --inflight 1 --steps 5 --warmup 1 2>/dev/null | tail -1 | val)" | tee $OUT/sweep.txt
echo "a256 inflight 3: $($B --ants 256 --inflight 3 --steps 12 --warmup 3 2>/dev/null | tail -1 | val)" | tee -a $OUT/sweep.txt
cd /tmp
rm -rf /tmp/p5 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats -d /tmp/p5 -- python $ROOT/bench.py --ants 256 --steps 4 --warmup 1 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
python $ROOT/tools/prof_summary.py $(find /tmp/p5 -name "*.db" | head -1) > $OUT/kernel_stats_a256.txt
head -12 $OUT/kernel_stats_a256.txt
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -n 6 --timeout=1200 -p no:cacheprovider > $OUT/suite.log 2>&1; echo "suite rc=$?" >> $OUT/rc.txt
tail -8 $OUT/suite.log
