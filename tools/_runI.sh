cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4j; ROOT=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_music_subspace.py tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "eig or music or subspace or chain or fft2d" 2>&1 | tail -4
cd /tmp
for v in reg lds; do
  if [ $v = lds ]; then export ISAC_EIG_TRIDIAG_LDS=1; else unset ISAC_EIG_TRIDIAG_LDS; fi
  rm -rf /tmp/p5 && ISAC_SINGLE_STREAM=1 rocprofv3 --kernel-trace -d /tmp/p5 -- python $ROOT/bench.py --steps 20 --warmup 3 --inflight 1 --no-cpu-baseline --prime-ms 0 > /dev/null 2>&1
  echo "== $v"; python $ROOT/tools/prof_summary.py $(find /tmp/p5 -name "*.db" | head -1) | grep -E "tridiag|subspace|bisect" | cut -c1-150
  python $ROOT/bench.py --no-cpu-baseline --inflight 1 --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('blocking', d['value'], d['ms_per_step'], d['pipeline']['blocking_cpi_ms'])"
done 2>&1 | tee $ROOT/gpurun_out/r4j/tridiag_reg.txt
