#!/bin/bash
# Development probe: P independent processes (like MATLAB parallel workers, one cell each), every one issuing BLOCKING CPIs
# (--inflight 1) on the same GPU at the same time; prints the per-process and the aggregate sensing rate.
P=${1:-4}
STEPS=${2:-150}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
pids=()
for i in $(seq 1 $P); do
  python $ROOT/bench.py --no-cpu-baseline --inflight 1 --steps $STEPS --warmup 20 > /tmp/mp_$i.json 2>/dev/null &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
python - <<PY
import json
tot = 0.0
for i in range(1, $P + 1):
    d = json.loads(open(f"/tmp/mp_{i}.json").read().strip().splitlines()[-1])
    tot += d["value"]
    print(f"process {i}: {d['value']:.0f} slots/s ({d['ms_per_step']:.2f} ms per blocking CPI)")
print(f"$P processes, blocking calls: aggregate {tot:.0f} slots/s")
PY
