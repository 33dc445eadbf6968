# development sweep: GPU_MAX_HW_QUEUES x CPIs in flight (bench.py, 100 steps)
for cfg in "16 8" "20 10" "24 10" "24 12" "16 7" "12 6"; do set -- $cfg; q=$1; inf=$2
export GPU_MAX_HW_QUEUES=$q
for i in 1 2; do
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --inflight $inf > gpurun_out/b.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/b.json"))
print("queues=$q inflight=$inf", d["value"], d["ms_per_step"])
PY
done; done
