#!/bin/bash
# The first multi-GPU run as ONE command: the N = 1 / 2 / 4 / 8 lines of BASELINE configs[2] (7 cells, strong; 7 cells per GPU, weak) and
# configs[4] (21 cells x 10 UEs) with the in-run N = 1 leg, + the efficiencies bench.py computes from it.  Needs an 8-GPU node; on fewer GPUs the larger N are skipped.
#   tools/scale_curve.sh [steps] [out_dir]
# NOTE: no scaling curve has been measured on hardware yet (one-GPU build boxes): this script is what produces it.
set -u
STEPS=${1:-20}; OUT=${2:-gpurun_out/scale_curve}; mkdir -p "$OUT"
NGPU=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
echo "# $(date -u +%FT%TZ) scale curve on $NGPU GPU(s), $STEPS steps" | tee "$OUT/summary.txt"
run() {   # name, n, args...
  local name=$1 n=$2; shift 2
  if [ "$n" -gt "$NGPU" ]; then echo "$name N=$n: skipped ($NGPU GPU(s))" | tee -a "$OUT/summary.txt"; return; fi
  python bench.py --gpus "$n" --steps "$STEPS" --warmup 5 --no-cpu-baseline --no-cold $( [ "$n" -gt 1 ] && echo --n1-leg ) "$@" 2> "$OUT/${name}_n$n.err" | tail -1 > "$OUT/${name}_n$n.json"
  python - "$OUT/${name}_n$n.json" "$name" "$n" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    sp = d.get("scaling_point") or {}
    print(f"{sys.argv[2]} N={sys.argv[3]}: {d['value']} {d['unit']} ({d['ms_per_step']} ms per step, scaling {d['scaling']}); N=1 in run {sp.get('n1_value')}, efficiency {sp.get('efficiency_vs_n1')}; cells gathered {len(d.get('cells', []))}")
except Exception as e:
    print(f"{sys.argv[2]} N={sys.argv[3]}: FAILED ({e!r})")
PY
}
for n in 1 2 4 8; do run config3_7cells_strong "$n" --cells 7; done
for n in 1 2 4 8; do run config3_7cells_per_gpu_weak "$n" --cells-per-gpu 7; done
for n in 1 2 4 8; do run config5_21x10 "$n" --workload config5 --cells 21 --ues 10 --steps $(( STEPS > 5 ? 5 : STEPS )); done
