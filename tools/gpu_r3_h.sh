#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3h; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
B="python $ROOT/bench.py --no-cpu-baseline"
val() { python -c "import json,sys; d=json.loads(sys.stdin.read()); h=d['host_timeline']; print(d['value'], d['ms_per_step'], d['pipeline']['blocking_cpi_ms'], 'collect', h['collect_wait_ms'], 'totals', h['enqueue_total_ms'], h['collect_total_ms'])"; }
python -c "
import ctypes
hip=ctypes.CDLL('/opt/rocm/lib/libamdhip64.so'); lo=ctypes.c_int(); hi=ctypes.c_int(); hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)); print('priority range least', lo.value, 'greatest', hi.value)" | tee $OUT/sweep.txt
for m in 0 1 2 3; do for st in 100 20; do
  echo "prio mode $m steps $st: $(ISAC_CTX_PRIO_MODE=$m $B --steps $st --warmup 5 2>/dev/null | tail -1 | val)"
done; done | tee -a $OUT/sweep.txt
for m in 1 3; do echo "prio mode $m inflight 6: $(ISAC_CTX_PRIO_MODE=$m $B --inflight 6 2>/dev/null | tail -1 | val)"; done | tee -a $OUT/sweep.txt
echo "prio mode 1 steps 1000: $(ISAC_CTX_PRIO_MODE=1 $B --steps 1000 2>/dev/null | tail -1 | val)" | tee -a $OUT/sweep.txt
