# development A/B: fused synthesis + range (default) vs fused synthesis + covariance
ISAC_FUSE_MODE=cov python -m pytest tests/test_gpu_spectral.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -5
for m in "range 2" "cov 2" "cov 0" "range 2" "cov 2"; do set -- $m
ISAC_FUSE_MODE=$1 ISAC_EC_SETSHIFT=$2 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/b.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/b.json"))
print("mode=$1 setshift=$2", d["value"], d["ms_per_step"], "blocking", d["pipeline"]["blocking_cpi_ms"], "fused-kernel ms", d["roofline"]["avg_launch_ms"])
PY
done
