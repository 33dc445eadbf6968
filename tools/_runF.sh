cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4h; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "cdl" 2>&1 | tail -2
timeout 900 python bench.py --workload config5 2>gpurun_out/r4h/c5.err | tail -1 > gpurun_out/r4h/c5.json; tail -1 gpurun_out/r4h/c5.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4h/c5.json")); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["whole_batch_call_ms"], d["comm_seams"])
PY
