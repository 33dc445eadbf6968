"""The driver's contract for bench.py: `python bench.py --gpus 1 --steps K --warmup W` prints exactly one JSON line on stdout with the
fields the driver and the judge read -- metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling /
vs_baseline / dtype / data / config.workload, `roofline` {bound, achieved, peak, unit, frac, traffic} for the dominant kernel and
`cpu_baseline` {value, unit, cores, kind, sample} from the CPU port on the host cores."""
from __future__ import annotations

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_carries_the_contract_fields():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "6", "--warmup", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]                       # ONE line on stdout, nothing else
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"].startswith("sensing slots/sec") and d["unit"] == "sensing slots/sec"
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "64-antenna" in d["config"]["workload"] and "K=3276 L=224" in d["config"]["workload"]
    assert abs(d["value"] - 16.0 * 6 / (d["ms_per_step"] * 6 / 1e3)) <= 1e-3 * d["value"]        # slots = 16 per CPI
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and 0.0 < rf["frac"] < 1.0                             # (no lower bar: under pytest-xdist other tests share the GPU)
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    if rf["bound"] == "mfma":
        # round 6, lazy echo grid (the default at this shape): the longest launch of the CPI is the covariance -- priced on its issued fp64 MFMA flops; the fused
        # echo + range kernel follows as `second_kernel`, priced on the bytes it still moves (txGrid read; no echoGrid store)
        assert "LAZY" in d["config"]["workload"] and rf["unit"] == "TFLOP/s" and rf["peak"] == 78.6 and "cov_lazy_kernel" in rf["kernel"]
        assert abs(rf["achieved"] - rf["issued_flops_per_launch"] / 1e12 / (rf["avg_launch_ms"] / 1e3)) <= 1e-2 * rf["achieved"]
        sk = rf["second_kernel"]
        assert sk["bound"] == "hbm" and sk["algorithmic_bytes_per_launch"] == 3276 * 224 * 64 * 16 and 0.0 < sk["frac"] < 1.0
        assert rf["avg_launch_ms"] > sk["avg_launch_ms"]
        assert rf["traffic"] is None or rf["traffic"] < 0.1 * sk["algorithmic_bytes_per_launch"]    # the covariance reads D and writes partial tiles: tens of MB
    else:
        assert rf["unit"] == "GB/s" and rf["peak"] == 8000.0
        assert rf["traffic"] is None or rf["traffic"] >= rf["algorithmic_bytes_per_launch"]
        assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / 1e9 / (rf["avg_launch_ms"] / 1e3)) <= 1e-2 * rf["achieved"]
    assert rf["whole_cpi"]["algorithmic_bytes"] == 2 * 3276 * 224 * 64 * 16 + (983040 + 3276 * 224) * 64 * 16    # SURVEY 8d's 3.261 GB per CPI, whatever the echo-grid mode
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == d["unit"] and cb["sample"]
    assert d["value"] > 10 * cb["value"]                           # north_star: >= 10x the CPU path on the same host
    assert "un-tuned" in cb["tuning"] and cb["cpi_s"]["min"] <= cb["cpi_s"]["median"] <= cb["cpi_s"]["max"]
    # what the committed profiles say, read from the CSVs (not typed into bench.py)
    top = rf["top_by_time"]
    assert top and top["source"].startswith("profiles/r") and 0 < top["share"] < 1 and top["kernel"].endswith("_kernel")
    assert rf["traffic_source"] is None or "profiles/r" in rf["traffic_source"]
    # host-side bookkeeping of the pipelined run
    assert d["pacing"]["mode"] in ("auto", "fixed", "off") and d["hw_queues"]["GPU_MAX_HW_QUEUES"] == "16" and d["host_timeline"]["enqueue_ms"]["p50"] > 0
    assert d["per_rank_ms"] and abs(d["per_rank_ms"][0] - d["ms_per_step"] * d["steps"]) <= 1e-2 * d["per_rank_ms"][0]


def test_config5_workload_line():
    """`bench.py --workload config5` (BASELINE configs[4]): per frame and cell one sensing CPI + every DL slot through every UE's CDL channel + every
    UE's CSI reports; one JSON line with slots/s, the MFMA roofline entry of the CDL contraction kernel and the per-job / per-UE seam times."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "bench.py", "--workload", "config5", "--cells", "2", "--ues", "4", "--steps", "1", "--warmup", "0", "--inflight", "2"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["dtype"] == "f64" and d["data"] == "synthetic" and "configs[4]" in d["config"]["workload"]
    assert d["per_frame_and_rank"] == {"cells": 2, "cdl_applies": 2 * 4 * 16, "csi_reports": 2 * 4 * 4, "sensing_cpis": 2, "ul_applies": 2 * 4 * 4, "precoded": True,
                                       "csi_h": "device, per occasion", "rank_selection": True, "srs_reports": 2 * 4}   # (round 5: 'U' slots, precoded PDSCH input, per-occasion device CSI; round 6: riSelect, SRS)
    assert abs(d["value"] - 2 * 20 / (d["ms_per_step"] / 1e3)) <= 1e-3 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 78.6 and 0.0 < rf["frac"] < 1.0
    assert abs(rf["achieved"] - rf["issued_flops_per_launch"] / 1e12 / (rf["avg_launch_ms"] / 1e3)) <= 1e-2 * rf["achieved"]
    assert d["comm_seams"]["cdl_apply_ms_per_job"] > 0 and d["comm_seams"]["csi_report_ms_per_ue"] > 0
    # the gathered per-cell records: whole estimate lists + every UE's last CSI report (wideband CQI, 18 subband CQIs at 273 PRB / 16-PRB subbands, PMI)
    assert len(d["cells"]) == 2
    for c in d["cells"]:
        assert len(c["ues"]) == 4 and all(u["cqi"] is None or 0 <= u["cqi"] <= 15 for u in c["ues"])
        assert all(len(u["sbCQI"]) == 18 and len(u["sbI2"]) == 18 and len(u["i1"]) == 3 for u in c["ues"])
        # round 6: the rank of each report (two-antenna UEs: 1 or 2) and the gNB's SRS measurement (ceil(273 / 16) = 18 subbands: TPMI 0..5 of the two-port codebook, CQI 1..15)
        assert all(u["ri"] in (1, 2) and len(u["ulTPMI"]) == 18 and len(u["ulCQI"]) == 18 for u in c["ues"])
        assert all(0 <= t <= 5 for u in c["ues"] for t in u["ulTPMI"]) and all(1 <= q <= 15 for u in c["ues"] for q in u["ulCQI"])
        if c["valid"]:
            assert len(c["rngEst"]) == c["nRng"] >= 1 and len(c["velEst"]) == c["nVel"] and len(c["aziEst"]) == c["nAzi"]
