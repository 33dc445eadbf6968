for cfg in "16 8" "16 6" "16 4"; do set -- $cfg; q=$1; inf=$2
export GPU_MAX_HW_QUEUES=$q
for i in 1 2 3; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --inflight $inf > gpurun_out/b.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/b.json"))
print("queues=$q inflight=$inf", d["value"], d["ms_per_step"], d["pipeline"]["blocking_cpi_ms"], d["roofline"]["frac"])
PY
done; done
