# development A/B of one environment switch on the same box:  bash tools/_ab_env.sh VAR v1 v2 [bench args]
VAR=$1; A=$2; B=$3; shift 3
for rep in 1 2 3; do for v in $A $B; do
env $VAR=$v python bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" > gpurun_out/b.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/b.json"))
print("$VAR=$v", d["value"], d["ms_per_step"], "blocking", d["pipeline"]["blocking_cpi_ms"], "cov", d["roofline"]["other_stages"][-1]["ms"], "fused", d["roofline"]["avg_launch_ms"])
PY
done; done
