cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4h; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -p no:cacheprovider -k "cdl" 2>&1 | tail -2
for n in 1 2; do timeout 900 python bench.py --workload config5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['paths'], d['comm_seams'])"; done
