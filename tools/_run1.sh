cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4a; export TMPDIR=/tmp
bash tools/ab_libs.sh "tree r3 philox7 tree r3" python tools/_er_variant_probe.py > gpurun_out/r4a/ab_fused.txt 2>&1
cat gpurun_out/r4a/ab_fused.txt
timeout 900 python -m pytest tests/test_gpu_spectral.py tests/test_gpu_fullsize.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/r4a/tests.txt
python - <<'PY' 2>&1 | tee gpurun_out/r4a/gen_accuracy.txt
import sys, numpy as np
sys.path.insert(0, "tests")
import oracle as O
from conftest import load_pkg, make_scene
pkg = load_pkg()
sc = make_scene(n_ants=4, n_slots=2, nrb=273, with_noise=False)
rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
got = pkg.sensing.monoStaticSensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, seed=12345, noise_domain="spectral", nfft=4096)
clean = pkg.sensing.monoStaticSensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, nfft=4096)
sig = np.sqrt(sc.rp.N0 / 2.0) * 64.0
nz = (got - clean) / sig
w = O.philox_spectral_noise(sc.K, sc.L, sc.A, 12345)
d = np.abs(nz - w)
print("device vs float32 restatement: max abs", d.max(), "max rel", (d / np.maximum(np.abs(w), 1e-3)).max(), "rms", np.sqrt((d**2).mean()), "n", w.size, "max|w|", np.abs(nz).max())
PY
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4a/bench_driver_$i.json; done
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4a/bench_100.json
python bench.py --no-cpu-baseline --inflight 1 2>/dev/null | tail -1 > gpurun_out/r4a/bench_blocking.json
python - <<'PY'
import json
for f in ("bench_driver_1","bench_driver_2","bench_100","bench_blocking"):
    d=[json.loads(l) for l in open("gpurun_out/r4a/%s.json"%f) if l.startswith("{")][-1]; print(f, d["value"], d["ms_per_step"], d["pipeline"]["blocking_cpi_ms"], d["roofline"].get("frac"))
PY
