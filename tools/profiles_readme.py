#!/usr/bin/env python3
"""Prints the "headline readings" block of profiles/README.md for a round tag FROM the committed CSV / JSON files (the prose
must not drift from the data):   python tools/profiles_readme.py r03 > /tmp/block.md"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"


def rows(name):
    f = os.path.join(P, f"{tag}_{name}")
    return list(csv.DictReader(open(f))) if os.path.exists(f) else []


def jline(name):
    f = os.path.join(P, f"{tag}_{name}")
    if not os.path.exists(f):
        return None
    for ln in reversed(open(f).read().splitlines()):
        if ln.startswith("{"):
            return json.loads(ln)
    return None


def short(k):
    import re
    m = re.search(r"\d+([a-z][a-z0-9_]*_kernel)", k.replace("isac", ""))
    return m.group(1) if m else k.strip('"')[:40]


def kern(rs, name):
    for r in rs:
        if name in r["kernel"]:
            return r
    return None


out = []
ss = rows("kernel_stats_single_stream.csv")
if ss:
    out.append(f"* Single-stream kernel trace (`{tag}_kernel_stats_single_stream.csv`, blocking call order): " +
               "; ".join(f"`{short(r['kernel'])}` {float(r['avg_us']):.1f} us x {r['calls']} ({float(r['pct']):.1f} %)" for r in ss[:9]) + ".")
    top = max(ss, key=lambda r: float(r["total_us"]))
    out.append(f"* Largest share of GPU time: `{short(top['kernel'])}` ({float(top['pct']):.1f} %, {float(top['avg_us']):.1f} us avg).")
fe, wr = rows("pmc_fetch_size.csv"), rows("pmc_write_size.csv")
for name, alg in (("echo_range_", 1.5029), ("cov_mfma_lds_kernel", 0.7514), ("cov_mfma_small_kernel", 0.7514), ("beamsum_kernel", 1.0066)):
    a, b = kern(fe, name), kern(wr, name)
    if a and b:
        f_kb, w_kb = float(a["avg_value"]), float(b["avg_value"])
        out.append(f"* `{short(a['kernel'])}`: FETCH_SIZE 2 x {f_kb:,.0f} KB + WRITE_SIZE {w_kb:,.0f} KB = {(2 * f_kb + w_kb) * 1024 / 1e9:.3f} GB per launch "
                   f"(algorithmic {alg:.3f} GB); {float(a['avg_duration_us']):.1f} us in the FETCH pass.")
mf = rows("pmc_mfma_busy.csv")
cov = [r for r in mf if "cov_mfma_lds" in r["kernel"] or "cov_mfma_small" in r["kernel"]]
if cov:
    d = {r["counter"]: float(r["avg_value"]) for r in cov}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
        out.append(f"* `{short(cov[0]['kernel'])}`: SQ_VALU_MFMA_BUSY_CYCLES {d['SQ_VALU_MFMA_BUSY_CYCLES']:,.0f} (per counter instance, x 32 instances) over GRBM_GUI_ACTIVE "
                   f"{d['GRBM_GUI_ACTIVE']:,.0f} cycles x 1024 SIMDs -> MfmaUtil {32 * d['SQ_VALU_MFMA_BUSY_CYCLES'] / (d['GRBM_GUI_ACTIVE'] * 1024):.2f}.")
wl = [r for r in rows("pmc_wait_lds.csv") if "echo_range_" in r["kernel"]]
if wl:
    d = {r["counter"]: float(r["avg_value"]) for r in wl}
    if "SQ_WAIT_ANY" in d and "SQ_WAVE_CYCLES" in d:
        s = f"* `echo_range_sl_kernel` (final kernel of the round): SQ_WAIT_ANY / SQ_WAVE_CYCLES = {d['SQ_WAIT_ANY'] / d['SQ_WAVE_CYCLES']:.2f}"
        if "SQ_LDS_BANK_CONFLICT" in d and "SQ_LDS_IDX_ACTIVE" in d:
            s += f"; SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = {d['SQ_LDS_BANK_CONFLICT']:,.0f} / {d['SQ_LDS_IDX_ACTIVE']:,.0f} = {d['SQ_LDS_BANK_CONFLICT'] / d['SQ_LDS_IDX_ACTIVE']:.2f}"
        out.append(s + f" (`{tag}_pmc_wait_lds.csv`).")
a_f, a_m, a_w = rows("pmc_a256_fetch_size.csv"), rows("pmc_a256_mfma_busy.csv"), rows("pmc_a256_wait_lds.csv")
blk = [r for r in a_f + a_m + a_w if "cov_mfma_block" in r["kernel"]]
if blk:
    d = {r["counter"]: float(r["avg_value"]) for r in blk}
    dur = {r["counter"]: float(r["avg_duration_us"]) for r in blk}
    s = f"* `{short(blk[0]['kernel'])}` (A = 256, PMC passes of a cold 3-CPI run):"
    if "FETCH_SIZE" in d:
        s += f" FETCH_SIZE 2 x {d['FETCH_SIZE']:,.0f} KB = {2 * d['FETCH_SIZE'] * 1024 / 1e9:.2f} GB per launch ({dur['FETCH_SIZE']:.0f} us);"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
        s += (f" MfmaUtil {32 * d['SQ_VALU_MFMA_BUSY_CYCLES'] / (d['GRBM_GUI_ACTIVE'] * 1024):.2f} at {d['GRBM_GUI_ACTIVE'] / dur['GRBM_GUI_ACTIVE'] / 1e3:.2f} GHz"
              f" (GRBM_GUI_ACTIVE / duration);")
    if "SQ_WAIT_ANY" in d and "SQ_WAVE_CYCLES" in d:
        s += f" SQ_WAIT_ANY / SQ_WAVE_CYCLES {d['SQ_WAIT_ANY'] / d['SQ_WAVE_CYCLES']:.2f}; SQ_LDS_BANK_CONFLICT {d.get('SQ_LDS_BANK_CONFLICT', 0):,.0f} of {d.get('SQ_LDS_IDX_ACTIVE', 0):,.0f} LDS cycles"
    out.append(s + ".")
for name, label in (("bench_driver_invocation.json", "driver invocation (`--gpus 1 --steps 20 --warmup 5`)"), ("bench_driver_invocation_2.json", "the same again"),
                    ("bench_default_100steps.json", "100 steps"), ("bench_default_100steps_ordered.json", "100 steps, `--schedule ordered` (all contexts on one pair of streams)"), ("bench_blocking.json", "`--inflight 1` (blocking CPIs)"),
                    ("bench_blocking_full_eig.json", "`--inflight 1`, full eigendecomposition route (`ISAC_MUSIC_FULL_EIG=1`: the round-2 eigensolver)"),
                    ("bench_default_100steps_old_cfar.json", "100 steps, per-antenna CFAR + memset + count (`ISAC_TAIL_UNFUSED=1`)"),
                    ("bench_7cells_per_gpu.json", "7 cells per GPU"), ("bench_a256.json", "`--ants 256 --inflight 3`"),
                    ("bench_a256_inflight6.json", "`--ants 256 --inflight 6`"),
                    ("bench_a256_blocking.json", "`--ants 256 --inflight 1`"), ("bench_a16.json", "`--ants 16`"), ("bench_traced_pipelined.json", "under rocprofv3, 300 steps, `--trace-only`")):
    d = jline(name)
    if d:
        rf = d.get("roofline", {})
        out.append(f"* bench, {label}: **{d['value']:,.0f} slots/s**, {d['ms_per_step']:.3f} ms per step" +
                   (f", blocking CPI {d['pipeline']['blocking_cpi_ms']} ms" if d.get("pipeline", {}).get("blocking_cpi_ms") else "") +
                   (f", `roofline.frac` {rf.get('frac')} ({rf.get('avg_launch_ms')} ms per launch), whole-CPI frac {rf.get('whole_cpi', {}).get('frac')}" if rf.get("frac") else "") + ".")
c5 = jline("bench_config5_21x10.json")
if c5:
    rf = c5["roofline"]
    out.append(f"* bench, `--workload config5` (21 cells x 10 UE on ONE GPU; per 20-slot frame: {c5['per_frame_and_rank']['sensing_cpis']} sensing CPIs + {c5['per_frame_and_rank']['cdl_applies']} CDL applies + "
               f"{c5['per_frame_and_rank']['csi_reports']} CSI reports): **{c5['value']:,.0f} slots/s**, {c5['ms_per_step']:.1f} ms per frame; `{rf.get('kernel', 'cdl_gemm_kernel').split('<')[0].split(' ')[0]}` {rf['avg_launch_ms']} ms per {rf['jobs_per_launch']}-job launch = "
               f"{rf['achieved']} TFLOP/s issued = `roofline.frac` {rf['frac']} of the fp64 MFMA peak; CDL apply {c5['comm_seams']['cdl_apply_ms_per_job']} ms per job, CSI report {c5['comm_seams']['csi_report_ms_per_ue']} ms per UE.")
for name, label in (("bench_config5_21x10_no_ul.json", "without the 'U' slots' uplink applies (`ISAC_C5_NO_UL=1`)"),
                    ("bench_config5_21x10_r04_workload.json", "the round-4 workload shape (`ISAC_C5_NO_UL=1 ISAC_C5_HOST_CSI=1`: no uplink, CSI estimate evaluated once on the host)"),
                    ("bench_config5_21x10_r04_workload_unfused_cdl.json", "the round-4 workload shape through the UNFUSED CDL kernels (`+ ISAC_CDL_UNFUSED=1`: Z through HBM)")):
    d = jline(name)
    if d:
        rf = d["roofline"]
        out.append(f"* config 5, {label}: **{d['ms_per_step']:.1f} ms per frame** ({d['value']:,.0f} slots/s); priced launch {rf['avg_launch_ms']} ms per {rf['jobs_per_launch']} jobs = `roofline.frac` {rf['frac']}; "
                   f"per frame and rank {d['per_frame_and_rank']}.")
cold = os.path.join(P, f"{tag}_cold_probe_lines.json")
if os.path.exists(cold):
    for ln in open(cold):
        if ln.startswith("{"):
            d = json.loads(ln)
            out.append(f"* cold probe, mode `{d['mode']}`" + (f" (isac_ctx_reserve {d.get('reserve_ms')} ms, warm_ms {d.get('reserve_warm_ms_requested')})" if d['mode'] == 'reserve' else "") +
                       f": first blocking CPI {d['first_cpi_ms']} ms, CPIs 2..21 median {d['cpi_2_21_ms']['median']} / max {d['cpi_2_21_ms']['max']} ms, steady blocking CPI {d['steady_blocking_cpi_ms']} ms (fresh process, inputs resident, device idle 1 s).")
d0c = jline("bench_driver_invocation.json")
if d0c and d0c.get("cold"):
    c = d0c["cold"]
    if "first_cpi_ms" in c.get("reserve", {}) and "first_cpi_ms" in c.get("noreserve", {}):
        out.append(f"* `cold` block of the driver's line: first CPI {c['noreserve']['first_cpi_ms']} ms straight in, {c['reserve']['first_cpi_ms']} ms after isac_ctx_reserve ({c['reserve'].get('reserve_ms')} ms).")
c5k = rows("kernel_stats_config5.csv")
if c5k:
    out.append(f"* config 5 under rocprofv3 (`{tag}_kernel_stats_config5.csv`, warm-up + one frame): " + "; ".join(f"`{short(r['kernel'])}` {float(r['avg_us']):.1f} us x {r['calls']} ({float(r['pct']):.1f} %)" for r in c5k[:8]) + ".")
d0 = jline("bench_driver_invocation.json")
if d0 and d0.get("cpu_baseline"):
    cb = d0["cpu_baseline"]
    impl = cb.get("implementations", {})
    out.append(f"* `cpu_baseline` of the driver's line: {cb['value']} slots/s ({cb.get('language')}, {cb['cores']} threads)" +
               (f"; both implementations: C++/OpenMP port {impl['cpp_openmp_port']['value']} slots/s, NumPy / SciPy oracle {impl['numpy_scipy_oracle']['value']} slots/s ({impl['numpy_scipy_oracle']['cpi_s']} s per CPI)" if impl else "") +
               f"; GPU / CPU = {d0.get('gpu_vs_cpu_baseline')}.")
for name in ("pipeline_overlap.txt", "pipeline_gaps.txt"):
    f = os.path.join(P, f"{tag}_{name}")
    if os.path.exists(f):
        lines = open(f).read().splitlines()
        out.append(f"* `{tag}_{name}`: " + " | ".join(lines[:2]))
a256 = rows("kernel_stats_single_stream_a256.csv")
if a256:
    out.append(f"* A = 256 single-stream kernel trace (`{tag}_kernel_stats_single_stream_a256.csv`): " + "; ".join(f"`{short(r['kernel'])}` {float(r['avg_us']):.1f} us x {r['calls']} ({float(r['pct']):.1f} %)" for r in a256[:6]) + ".")
f = os.path.join(P, f"{tag}_tridiag_dist_probe.txt")
if os.path.exists(f):
    ln = [l for l in open(f).read().splitlines() if "distributed tridiagonalisation" in l]
    if ln:
        out.append(f"* `{tag}_tridiag_dist_probe.txt` (n = 256, in-kernel clocks of the last wavefront): " + ln[-1].split("phases(x64 clk): ")[-1] + ".")
print("\n".join(out))
# the generated block of profiles/README.md is replaced in place (from its title line to the next "---")
readme = os.path.join(P, "README.md")
if os.path.exists(readme) and "--no-write" not in sys.argv:
    txt = open(readme).read()
    title = f"Headline readings (generated: `python tools/profiles_readme.py {tag}`)"
    if title in txt:
        a = txt.index(title)
        b = txt.index("\n---", a)
        open(readme, "w").write(txt[:a] + title + "\n" + "\n".join(out) + "\n" + txt[b:])
