#!/bin/bash
# Round-6 final refresh: gpurun -- 'bash tools/prof_r06_final.sh'  -> gpurun_out/prof_r06/ (copied into profiles/ by hand)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_r06; mkdir -p $OUT
export TMPDIR=/tmp
PS="python $ROOT/tools/prof_summary.py"
db() { find "$1" -name "*.db" | head -1; }
# 1. kernel trace + PMC passes of the bench's blocking call sequence, lazy and array echo grid
bash tools/prof_r06.sh r06 > $OUT/r06_prof_console.txt 2>&1
# 2. driver-style bench lines
cd $ROOT
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/r06_bench_default_lazy.json
python bench.py --gpus 1 --steps 20 --warmup 5 --echo array --no-cold 2>/dev/null | tail -1 > $OUT/r06_bench_array.json
python bench.py --gpus 1 --steps 100 --warmup 5 --no-cold --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r06_bench_default_lazy_100steps.json
python bench.py --targets 4 --no-cold --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r06_bench_q4.json
python bench.py --targets 2 --no-cold --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r06_bench_q2.json
python bench.py --ants 256 --no-cold --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r06_bench_a256.json
python bench.py --ants 16 --no-cold --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r06_bench_a16.json
python bench.py --workload config5 --steps 3 --warmup 1 2>/dev/null | tail -1 > $OUT/r06_bench_config5_21x10.json
ISAC_CDL_TIME_DOMAIN=1 python bench.py --workload config5 --steps 3 --warmup 1 2>/dev/null | tail -1 > $OUT/r06_bench_config5_21x10_time_domain_cdl.json
python tools/c5_parts_probe.py > $OUT/r06_config5_parts.txt 2>&1
# 3. config 5 kernel stats and idle gaps (one traced frame after a warm-up frame)
cd /tmp
rm -rf /tmp/q5 && rocprofv3 --kernel-trace -d /tmp/q5 -- python $ROOT/bench.py --workload config5 --steps 2 --warmup 1 > /dev/null 2>&1
$PS $(db /tmp/q5) --csv $OUT/r06_kernel_stats_config5.csv > $OUT/r06_kernel_stats_config5.txt
python $ROOT/tools/c5_gaps.py $(db /tmp/q5) 30 > $OUT/r06_config5_idle_gaps.txt 2>&1
# 4. the overlap-save launches alone (40-job DL batches, 20-job UL batch)
for p in CDL-A CDL-D; do rm -rf /tmp/q6; rocprofv3 --kernel-trace -d /tmp/q6 -- python $ROOT/tools/cdl_os_probe.py $p 5 > /dev/null 2>&1; echo "== $p downlink, 40 jobs"; $PS $(db /tmp/q6) | head -5; done > $OUT/r06_cdl_os_kernels.txt 2>&1
rm -rf /tmp/q7; rocprofv3 --kernel-trace -d /tmp/q7 -- python $ROOT/tools/ul_probe.py > /dev/null 2>&1; { echo "== uplink, 20 jobs"; $PS $(db /tmp/q7) | head -5; } >> $OUT/r06_cdl_os_kernels.txt 2>&1
# 5. A = 256 kernel stats
rm -rf /tmp/q8 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace -d /tmp/q8 -- python $ROOT/bench.py --ants 256 --no-cpu-baseline --no-cold --prime-ms 0 --inflight 1 --steps 10 --warmup 2 > /dev/null 2>&1
$PS $(db /tmp/q8) --csv $OUT/r06_kernel_stats_single_stream_a256.csv > $OUT/r06_kernel_stats_single_stream_a256.txt
# 6. plain-C host batch mode
cd $ROOT
{ for q in 4 8; do for a in 16 64; do echo "GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q tests/_build/abi_host batch $a $([ $a = 16 ] && echo 16 || echo 8) 200; done; done; } > $OUT/r06_abi_host_batch.txt 2>&1
ls $OUT
for f in $OUT/r06_bench_*.json; do python - "$f" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip() or "{}")
r = d.get("roofline", {})
print(sys.argv[1].split("/")[-1], d.get("value"), d.get("ms_per_step"), (d.get("pipeline") or {}).get("blocking_cpi_ms"), r.get("bound"), r.get("frac"), r.get("avg_launch_ms"))
PY
done
