#!/usr/bin/env python3
"""Sensing hot-path benchmark:  python bench.py --gpus N --steps K --warmup W

One "step" = one CPI (coherent processing interval) of the sensing chain on one cell:
    sensing.monoStaticSensing  ->  sensing.estimation.fft2D
(echo synthesis + OFDM demodulation, range-Doppler map, 2D CA-CFAR, covariance, MUSIC DoA)
on the 100 MHz / 30 kHz / 273-PRB shape with 64 antennas (BASELINE.json configs[1]):
K = 3276 subcarriers, L = 224 symbols (16 grid slots), T = 983 040 samples, nIFFT 4096, nFFT 256.
metric = sensing slots/sec = 16 slots per CPI x CPIs/sec, summed over ranks (cells shard across
GPUs with no data-path collective; one RCCL all-gather of the per-cell result records at the end).

Inputs are synthetic and generated ON THE DEVICE before the timed region (QPSK txGrid, CP-OFDM
txWaveform, Philox AWGN inside the kernels).  Prints one JSON line on rank 0.

`--gpus N` without a torchrun environment launches N ranks itself (one per GPU, RCCL); under torchrun the
launcher's RANK / LOCAL_RANK / WORLD_SIZE are used.  `roofline` quotes the dominant HBM-bound kernel (the fused
echo-synthesis + range-stage kernel: the largest share of the wide kernels' time and of the bytes), timed with HIP
events inside the run; `roofline.top_by_time` names whichever kernel has the largest time share in the committed
rocprofv3 single-stream summary (profiles/rNN_kernel_stats_single_stream.csv), and `roofline.traffic` is read from the
committed PMC passes (profiles/rNN_pmc_*.csv).  `cpu_baseline` is the un-tuned C++/OpenMP port of the same chain
(oracle/cpu_port) on every host core at full size.
"""
from __future__ import annotations

import argparse
import gc
import ctypes as C
import importlib
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np

# 8 CPIs in flight x 2 HIP streams each: give every stream its own hardware queue (streams that share a queue serialise: a
# 1 ms few-CU eigensolver kernel at the head of a shared queue stalls the wide kernels behind it).  Measured at 100 steps
# (profiles/r02_queue_sweep.txt): 8 queues / 4 CPIs 15.6-16.6 k slots/s, 16 / 6 17.3 k, 16 / 8 17.6-17.7 k, 32 / 8 the same, more CPIs
# than queues / 2 collapses (16 / 12: 16.0 k, 32 / 12: 8.5 k).  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "5g_based_system_level_integrated_sensing_and_communication_simulator_amd"

COPY_RATE_GBS = 6260.0        # flat float4 copy of the fused kernel's 2 x 0.75 GB on this part (the guide's form: 6.26-6.32 TB/s, profiles/r05_cbench_copy_rate.txt)
COPY_RATE_COLUMNS_GBS = 5730.0  # the same bytes copied in the fused kernel's own shape: one workgroup per 52 416-byte column
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured-achievable)
FP64_MFMA_PEAK_TFLOPS = 78.6     # MI355X fp64 matrix (= vector) dense peak
DOMINANT_KERNEL = "echo_range_"          # echo_range_sl_kernel<Q, NZ> (one / two LoS targets) or echo_range_kernel<Q, NZ>


def profile_facts(lazy=False):
    """What the committed rocprofv3 summaries (profiles/, newest round) say -- read, not hard-coded:
      top_by_time  the kernel with the largest share of GPU time in rNN[_mode]_kernel_stats_single_stream.csv (blocking call order, one stream);
      traffic      HBM bytes per launch of the fused kernel (and of the covariance launch: traffic_cov) at the A = 64 / 224-symbol shape from the separate --pmc passes:
                   2 x FETCH_SIZE (gfx950 counts wide coalesced reads at half, MI355X_MICROARCH.md) + WRITE_SIZE, KB -> B.
    Since round 6 the summaries exist per echo-grid mode (rNN_lazy_* / rNN_array_*, tools/prof_r06.sh); the mode of this run is read, older un-moded files otherwise.
    Not measurable from inside this process; quoted "from profile" and only for the shape it was collected at."""
    import csv
    import glob
    import re
    out = {"top_by_time": None, "traffic": None, "traffic_source": None, "traffic_cov": None}
    pdir = os.path.join(ROOT, "profiles")
    mode = "lazy" if lazy else "array"
    store_tag = "Lb0E" if lazy else "Lb1E"                    # echo_range_sl_kernel<Q, NZ, STORE, GROUP>: the STORE = false instantiation is the lazy one

    def newest(suffix):
        best = None
        for f in glob.glob(os.path.join(pdir, "r[0-9][0-9]*_" + mode + "_" + suffix)) + glob.glob(os.path.join(pdir, "r[0-9][0-9]_" + suffix)):
            m = re.match(r"r(\d\d)([a-z]?)_", os.path.basename(f))
            key = (int(m.group(1)), m.group(2)) if m else None
            if key and (best is None or key > best[0]):
                best = (key, f)
        return best[1] if best else None

    def is_dom(name):
        return DOMINANT_KERNEL in name and ("echo_range_sl_kernel" not in name or store_tag in name or "Lb" not in name)

    def short(name):
        m = re.match(r"_ZN4isac\d+([A-Za-z0-9_]+?)(?:I|E)", name.strip('"'))
        return m.group(1) if m else name.strip('"')[:48]
    try:
        f = newest("kernel_stats_single_stream.csv")
        if f:
            rows = list(csv.DictReader(open(f)))
            top = max(rows, key=lambda r: float(r["total_us"]))
            dom = [r for r in rows if is_dom(r["kernel"])]
            cov = [r for r in rows if ("cov_lazy_kernel" if lazy else "cov_mfma_") in r["kernel"]]
            out["top_by_time"] = {"kernel": short(top["kernel"]), "share": round(float(top["pct"]) / 100.0, 4), "avg_us": float(top["avg_us"]),
                                  "dominant_hbm_kernel_share": round(float(dom[0]["pct"]) / 100.0, 4) if dom else None,
                                  "dominant_hbm_kernel_avg_us": float(dom[0]["avg_us"]) if dom else None,
                                  "covariance_kernel_avg_us": float(cov[0]["avg_us"]) if cov else None,
                                  "source": "profiles/" + os.path.basename(f), "from_committed_profile": True}
        ff, fw = newest("pmc_fetch_size.csv"), newest("pmc_write_size.csv")
        if ff and fw:
            def kb(path, counter, pred):
                for r in csv.DictReader(open(path)):
                    if pred(r["kernel"]) and r["counter"] == counter:
                        return float(r["avg_value"])
                return None
            fe, wr = kb(ff, "FETCH_SIZE", is_dom), kb(fw, "WRITE_SIZE", is_dom)
            if fe is not None and wr is not None:
                out["traffic"] = int((2.0 * fe + wr) * 1024)
                out["traffic_source"] = (f"from profile: profiles/{os.path.basename(ff)} (x2, gfx950) + profiles/{os.path.basename(fw)}, "
                                         "separate --pmc passes, A = 64 / 224-symbol shape")
            is_cov = lambda n: ("cov_lazy_kernel" if lazy else "cov_mfma_") in n          # noqa: E731
            fe, wr = kb(ff, "FETCH_SIZE", is_cov), kb(fw, "WRITE_SIZE", is_cov)
            if fe is not None and wr is not None:
                out["traffic_cov"] = int((2.0 * fe + wr) * 1024)
    except Exception as e:                                     # a malformed summary must not break the benchmark
        out["error"] = repr(e)
    return out


def cell_params(n_ants, targets, velocity):
    """Flat per-cell struct with the defaults of scenarios/openStreetMapCity.m + gNBParameters.m."""
    p = SimpleNamespace()
    t = np.atleast_2d(np.asarray(targets, dtype=np.float64))
    p.numTargets = t.shape[0]
    p.targetPosition = t
    p.gNBPosition = np.array([0.0, 0.0, 30.0])
    p.tddPattern = "DDDSU"
    p.numDLSlots = 3
    p.numSlots = 20
    p.gNBTxAnts = n_ants
    p.dlCarrierFreq = 3.5e9
    p.gNBNoiseFigure = 6.0
    p.gNBTemperature = 290.0
    p.gNBTxPower = 46.0
    p.gNBRxGain = 25.5
    p.rcs = np.ones(t.shape[0])
    p.velocity = np.asarray(velocity, dtype=np.float64)
    p.gNBSenAntenna = SimpleNamespace(kind="ula", numElements=n_ants, d=0.5, nV=n_ants // 2, p=2)
    p.Pfa = 1e-9
    p.detectionArea = np.array([[50.0, 500.0], [-50.0, 50.0]])
    return p


class SlotPool:
    """`n` execution slots of one GPU (context = HIP streams + scratch) shared by all the cells it hosts: at most `n`
    CPIs are in flight on the GPU however many cells there are (two HIP streams per context: keep 2 n within
    GPU_MAX_HW_QUEUES -- 24 streams collapse to half the rate, profiles/r02_queue_sweep.txt).  Results are collected in submission order."""

    def __init__(self, pkg, device, n, ordered=False):
        self.ctxs = [pkg.Context(device) for _ in range(max(1, n))]
        self.ordered = bool(ordered) and len(self.ctxs) > 1
        if self.ordered:
            # every context enqueues on the first one's two streams, the covariance stays on the main stream: the device runs the wide
            # kernels of consecutive CPIs (beam-sum, fused echo + range, covariance) back to back, each alone, and the narrow MUSIC /
            # CFAR chains underneath on the second stream (include/isac.h: isac_ctx_share_streams, ISAC_OPT_WIDE_ORDER)
            for c in self.ctxs:
                c.set_wide_order(True)
            for c in self.ctxs[1:]:
                c.share_streams(self.ctxs[0])
        self.owner = [None] * len(self.ctxs)
        self.k = 0
        self.timeline = None          # list of (collect_s, enqueue_s) per submit while recording (host-side stalls show up here)
        self.pace_s = 0.0             # minimum spacing of consecutive submissions (0 = none): see --pace-ms
        self.next_t = 0.0
        self.spin_run = 0             # consecutive submissions that had to wait for their pace slot (beyond a pipeline fill: the pace throttles)

    def collect(self, slot):
        cell, self.owner[slot] = self.owner[slot], None
        return cell.finish(self.ctxs[slot]) if cell is not None else None

    def submit(self, cell):
        slot = self.k % len(self.ctxs)
        t0 = time.perf_counter()
        self.collect(slot)
        t1 = time.perf_counter()
        if self.pace_s > 0.0:
            spun = t1 < self.next_t
            while t1 < self.next_t:
                t1 = time.perf_counter()
            self.next_t = max(t1, self.next_t) + self.pace_s
            # A pace just below the GPU-bound period never makes the host wait here (the wait for the oldest CPI above is longer); waiting on
            # every submission well past a pipeline fill means the pace itself has become the bottleneck: back off
            self.spin_run = self.spin_run + 1 if spun else 0
            if self.spin_run >= 2 * len(self.ctxs) + 4:
                self.pace_s *= 0.85
                self.spin_run = len(self.ctxs) // 2
        cell.enqueue(self.ctxs[slot])
        if self.timeline is not None:
            self.timeline.append((t1 - t0, time.perf_counter() - t1))
        self.owner[slot] = cell
        self.k += 1

    def drain(self):
        for s in range(len(self.ctxs)):
            self.collect((self.k + s) % len(self.ctxs))

    def sync(self):
        for c in self.ctxs:
            c.sync()


class Cell:
    """One cell's device-resident inputs + the per-CPI call chain.  CPIs run on the slots of a SlotPool (its own
    `inflight` slots unless a shared pool is given): consecutive CPIs overlap on the GPU -- the MUSIC branch of
    CPI i (covariance -> eig -> scan, latency-bound on one CU) runs under the echo / range kernels of CPI i+1.
    `n_buf` buffer sets (transmit grid + waveform + echo grid) = the number of CPIs of THIS cell that can be in
    flight at once."""

    def __init__(self, pkg, device, cell_id, n_ants, n_slots, n_targets, inflight=1, fuse=True, pool=None, n_buf=None, noise_domain="spectral", lazy=False):
        L = pkg._lib
        self.fuse = fuse
        self.lazy = bool(lazy) and fuse           # echo grid kept inside the context (isac_mono_static_sensing_fused_dev with d_echo_grid == NULL): --echo lazy
        self.noise_domain = noise_domain
        self.pkg, self.L = pkg, L
        # device arrays are device-global, so a cell's inputs can be consumed on any context of the same device
        self.pool = pool or SlotPool(pkg, device, inflight)
        self.ctxs = self.pool.ctxs
        n_buf = n_buf or len(self.ctxs)
        ctx = self.ctx = self.ctxs[0]
        rng = np.random.default_rng(0x5EED0003 + cell_id)
        r = rng.uniform(50.0, 350.0, n_targets)
        az = np.deg2rad(rng.uniform(-60.0, 60.0, n_targets))
        targets = np.stack([r * np.cos(az), r * np.sin(az), np.full(n_targets, 1.5)], axis=1)
        vel = rng.integers(-10, 11, n_targets).astype(np.float64)
        self.carrier = SimpleNamespace(NRBsDL=273, SubcarrierSpacing=30)
        self.wave = SimpleNamespace(Nfft=4096, SampleRate=122.88e6, SymbolsPerSlot=14)
        self.cellp = cell_params(n_ants, targets, vel)
        self.rp = pkg.sensing.radarParams(self.cellp, self.carrier, self.wave)
        self.cfar = pkg.sensing.detection.cfar2D(self.rp)
        self.K, self.Lsym, self.A = 3276, 14 * n_slots, n_ants
        car = L.Carrier(self.K, 4096, 30, 0)
        t = C.c_int64(0)
        ctx.lib.isac_ofdm_waveform_length(C.byref(car), C.c_int32(self.Lsym), C.byref(t))
        self.T = int(t.value)
        # every in-flight CPI has its OWN transmit grid / waveform (a new frame of PDSCH data per CPI, as in the
        # simulator): CPIs that shared one input buffer would hit each other's lines in the Infinity Cache and
        # overstate the rate by ~13 % (measured)
        amp = 10.0 ** ((46.0 - 30.0) / 20.0) * np.sqrt(4096.0 ** 2 / (self.K * self.A))      # gNBPhy.m:592
        self.tx_grids, self.tx_waves = [], []
        for s in range(n_buf):
            g = ctx.empty((self.K, self.Lsym, self.A))
            w = ctx.empty((self.T, self.A))
            ctx.check(ctx.lib.isac_synth_qpsk_grid_dev(ctx.handle, C.c_void_p(g.ptr), self.K, self.Lsym, self.A,
                                                       C.c_uint64(0x5EED0001 + 1000 * s + cell_id), 1))
            ctx.check(ctx.lib.isac_ofdm_modulate_dev(ctx.handle, C.c_void_p(g.ptr), self.Lsym, self.A, C.byref(car),
                                                     C.c_double(amp), C.c_void_p(w.ptr), C.c_int64(self.T)))
            self.tx_grids.append(g)
            self.tx_waves.append(w)
        self.tx_grid, self.tx_wave = self.tx_grids[0], self.tx_waves[0]
        self.los = np.ones(n_targets, dtype=np.uint8)
        self.echo = [ctx.empty((self.K, self.Lsym, self.A)) for _ in range(n_buf)]  # one echo grid per in-flight CPI
        self.seed = 0x5EED0002 + cell_id
        self.n_sub = 0
        self.last = None
        self.profile_sink = None      # list: per-launch durations of the fused kernel (HIP events recorded by the library)
        ctx.sync()

    def finish(self, ctx):
        """Collect the CPI this cell enqueued on `ctx`."""
        est = None
        if self.profile_sink is not None and self.fuse:
            ms = C.c_double(0.0)
            if ctx.lib.isac_profile_last_kernel_ms(ctx.handle, C.byref(ms)) == 0:
                self.profile_sink.append(ms.value)
        try:
            est = self.pkg.sensing.estimation.fft2D_collect(ctx)
        except self.pkg.IsacError as e:           # reference: try/catch -> senResults = NaN (cellSimulation.m:196-202)
            if e.name != "NO_DETECTION":
                raise
        self.last = est
        return est

    def enqueue(self, c):
        """Enqueue one CPI (monoStaticSensing -> fft2D) on context `c`."""
        b = self.n_sub % len(self.echo)
        tx_wave, tx_grid = self.tx_waves[b], self.tx_grids[b]
        echo = self.pkg.sensing.monoStaticSensing(tx_wave, (self.K, self.Lsym, self.A), self.carrier, self.rp, self.los,
                                                  seed=self.seed + self.n_sub, noise_domain=self.noise_domain, nfft=4096, out=None if self.lazy else self.echo[b], ctx=c,
                                                  fuse_fft2d=(self.rp, self.cfar, tx_grid) if self.fuse else None, lazy=self.lazy)
        self.pkg.sensing.estimation.fft2D_submit(self.rp, self.cfar, echo, tx_grid, ctx=c, reuse_range=self.fuse)
        self.n_sub += 1

    def submit(self):
        self.pool.submit(self)

    def drain(self):
        self.pool.drain()
        return self.last

    def step(self):
        """Blocking CPI (submit + collect), the reference's call order."""
        self.submit()
        return self.drain()

    def sync(self):
        self.pool.sync()

    def algorithmic_bytes(self):
        """echo+demod reads txWaveform and writes echoGrid; RDM+CFAR reads rxGrid + txGrid."""
        b = 16
        echo = self.T * self.A * b + self.K * self.Lsym * self.A * b
        rdm = 2 * self.K * self.Lsym * self.A * b
        return echo, rdm

    def time_dominant_kernel_isolated(self, reps=10):
        """Average duration of the dominant kernel (fused echo synthesis + range stage) with nothing else on the GPU: HIP events
        recorded by the library around exactly that launch (isac_profile_*), on the stream it is launched on."""
        c = self.ctx
        c.check(c.lib.isac_profile_enable(c.handle, 1))
        out = []
        for i in range(reps + 1):
            self.pkg.sensing.monoStaticSensing(self.tx_wave, (self.K, self.Lsym, self.A), self.carrier, self.rp, self.los, seed=self.seed + i,
                                               noise_domain=self.noise_domain, nfft=4096, out=None if self.lazy else self.echo[0], ctx=c, fuse_fft2d=(self.rp, self.cfar, self.tx_grid),
                                               lazy=self.lazy)
            ms = C.c_double(0.0)
            c.check(c.lib.isac_profile_last_kernel_ms(c.handle, C.byref(ms)))
            out.append(ms.value)
        c.sync()
        return float(np.mean(out[1:]))

    def time_covariance_kernel_isolated(self, reps=10):
        """Average duration of the wide covariance launch of fft2D (cov_lazy_kernel with a lazy echo grid, cov_mfma_lds_kernel / cov_mfma_block_pl_kernel on an array) with
        nothing else on the GPU: blocking CPIs on the cell's first context, the library's event pair around exactly that launch (isac_profile_enable(ctx, 2))."""
        c = self.ctx
        c.check(c.lib.isac_profile_enable(c.handle, 2))
        out = []
        for i in range(reps + 1):
            b = 0
            echo = self.pkg.sensing.monoStaticSensing(self.tx_wave, (self.K, self.Lsym, self.A), self.carrier, self.rp, self.los, seed=self.seed + i,
                                                      noise_domain=self.noise_domain, nfft=4096, out=None if self.lazy else self.echo[b], ctx=c,
                                                      fuse_fft2d=(self.rp, self.cfar, self.tx_grid) if self.fuse else None, lazy=self.lazy)
            self.pkg.sensing.estimation.fft2D_submit(self.rp, self.cfar, echo, self.tx_grid, ctx=c, reuse_range=self.fuse)
            try:
                self.pkg.sensing.estimation.fft2D_collect(c)
            except self.pkg.IsacError as e:
                if e.name != "NO_DETECTION":
                    raise
            ms = C.c_double(0.0)
            if c.lib.isac_profile_last_kernel_ms(c.handle, C.byref(ms)) == 0:
                out.append(ms.value)
        c.check(c.lib.isac_profile_enable(c.handle, 0))
        return float(np.mean(out[1:])) if len(out) > 1 else None

    def covariance_issued_flops(self):
        """fp64 MFMA flops the covariance launch issues on useful samples: 8 A^2 K L nominal x the share of 16 x 16 tiles computed (the upper triangle: of the 64 x 64
        block pairs and, inside a diagonal pair, of its tiles) x 3/4 (three real MFMAs per complex tile step).  Padding (the lazy form's masked partner rows, +6.5 % at
        K = 3276; antennas padded to a multiple of 16) is NOT counted."""
        nominal = 8.0 * self.A * self.A * self.K * self.Lsym
        nt = (self.A + 15) // 16
        if self.A <= 64:
            share = (nt * (nt + 1) / 2) / (nt * nt)
        else:
            nb = (self.A + 63) // 64
            share = (nb * 10 + (nb * (nb - 1) // 2) * 16) / (nb * nb * 16)
        return nominal * share * 0.75, nominal

    def dominant_kernel_bytes(self):
        """Algorithmic HBM bytes of one launch of the fused kernel.  Array form: txGrid read once + echoGrid written once = 2 K L A 16 B (SURVEY 8d's RDM+CFAR read of
        txGrid + monoStaticSensing's write of echoGrid; rxGrid is never re-read).  LAZY form: the kernel does not store echoGrid -- txGrid read once = K L A 16 B (its
        range-stage rows, 84 MB by the counters, are not counted)."""
        return (1 if self.lazy else 2) * self.K * self.Lsym * self.A * 16


def csirs_positions(nrb):
    """First-port CSI-RS resource elements of the reference's row-5 / density-1 / symbol-0 configuration (setupCSIRS.m:8-11): two adjacent
    subcarriers per RB on the slot's first symbol -- 1-based (k, l) subscripts as dlPMISelect.m:354-362 uses them."""
    k = np.concatenate([[12 * r + 1, 12 * r + 2] for r in range(nrb)])
    return k, np.ones_like(k)


def freq_response(ch, k_sub, n_sc, scs_hz, ports):
    """Perfect CSI-RS channel estimate at the subcarriers k_sub (1-based): H[i, u, p] = sum_n h[n, p, u] exp(-2 pi j f_i tau_n)."""
    h = ch.path_gains(ch.time)[:, :ports, :]                # [n, p, u]
    tau = ch.path_delays()
    f = ((np.asarray(k_sub) - 1) - n_sc / 2) * scs_hz
    e = np.exp(-2j * np.pi * f[:, None] * tau[None, :])     # [i, n]
    return np.asfortranarray(np.einsum("kn,npu->kup", e, h))


class CommCell:
    """BASELINE configs[4] ("full ISAC: MIMO PDSCH beamforming + SINR->CQI + mono-static sensing, 21 cells x 10 UE"): the communication seams of
    ONE cell over ONE 20-slot frame, as the reference steps them --
      * every slot that carries downlink symbols (12 'D' + 4 'S' of DDDSU x 4): the slot waveform (+ MaxChannelDelay zero rows) through EVERY UE's
        nrCDLChannel (uePhy.m:724-731: applyChannelModel on each gNB packet; CDL-D for LoS UEs, CDL-A otherwise, updateCDLModels.m:9-14);
      * every CSI-RS occasion (setupCSIRS.m:11: period 5 slots -> 4 per frame): every UE's Type-I PMI search + subband CQI (uePhy.m:901-908).
      * every 'U' slot (4 per frame): every UE's uplink packet [T x 2] through its UL channel into the 64-element gNB array (gNBPhy.m:833-864: applyChannelModel
        on each UE packet; cdl.m:78-85), one batched call per slot and delay-profile group   (round 5; ISAC_C5_NO_UL=1 leaves it out: the round-4 workload).
    Inputs are synthetic and resident in HBM before the timed region: the 16 DL slot waveforms -- since round 5 the PRECODED PDSCH: QPSK layers [K x 14 x nu]
    through prgPrecode (a rank-nu precoder per 4-PRB PRG, prgPrecode.m:107-134, isac_prg_precode_dev) and the OFDM modulator -- and one UL waveform per UE.
    The CSI-RS channel estimate of every occasion is formed ON THE DEVICE from that occasion's path gains (isac_cdl_path_gains_dev +
    isac_cdl_freq_response_dev; the estimator itself is out of scope; ISAC_C5_HOST_CSI=1: one host evaluation at set-up as in round 4).  Channel time advances
    from slot to slot and from frame to frame (path gains formed on the device per gain block)."""
    DL_SLOTS, UL_SLOTS, CSI_OCCASIONS, SLOT_T, LAYERS, PRG_PRBS = 16, 4, 4, 61440, 2, 4
    WITH_UL = os.environ.get("ISAC_C5_NO_UL") is None
    WITH_RI = os.environ.get("ISAC_C5_NO_RI") is None            # round 6: rank selection per CSI report (uePhy.m:900)
    WITH_SRS = os.environ.get("ISAC_C5_NO_SRS") is None          # round 6: the gNB's SRS measurement of every UE (gNBPhy.m:1023-1060)
    SRS_BAND = 16
    BATCH_OCCASIONS = os.environ.get("ISAC_C5_CSI_PER_OCCASION") is None   # round 6: a cell's four CSI-RS occasions of the frame as one batch
    SHARE = os.environ.get("ISAC_C5_NO_SHARE") is None                     # round 6: the two delay-profile groups of a cell share the forward transforms of their slot waveforms
    DEVICE_CSI = os.environ.get("ISAC_C5_HOST_CSI") is None

    def __init__(self, pkg, ctxs_cdl, ctx_csi, cell_id, n_ants, n_ues):
        CM, self.PL, L = pkg.communication.channelModels, pkg.communication.phyLayer, pkg._lib
        ctx_cdl = ctxs_cdl[0]
        self.CM, self.ctx, self.ctxs, self.ctx_csi, self.n_ues, self.A = CM, ctx_cdl, list(ctxs_cdl), ctx_csi, n_ues, n_ants
        self.cell_id = cell_id
        rng = np.random.default_rng(0xC5000 + cell_id)
        self.los = rng.random(n_ues) < 0.5
        nt_shape = (n_ants // 16, 8, 2, 1, 1) if n_ants >= 16 else (1, n_ants // 2, 2, 1, 1)
        self.chans = [CM.CDLChannel(DelayProfile="CDL-D" if lo else "CDL-A", TransmitAntennaArraySize=nt_shape, Seed=73) for lo in self.los]   # cdl.m:57-64
        self.T = self.SLOT_T + max(ch.info().MaxChannelDelay for ch in self.chans)            # uePhy.m:729: zero rows appended
        K = 3276
        car = L.Carrier(K, 4096, 30, 0)
        self.waves, self.precoders, self.layer_seeds = [], [], []          # (precoders / layer seeds kept for the oracle probe of tests/test_gpu_cdl_config5.py)
        grid, layers = ctx_cdl.empty((K, 14, n_ants)), ctx_cdl.empty((K, 14, self.LAYERS))
        n_prg = -(-273 // self.PRG_PRBS)
        dft = np.exp(-2j * np.pi * np.outer(np.arange(n_ants), np.arange(n_ants)) / n_ants) / np.sqrt(n_ants)
        for s_ in range(self.DL_SLOTS):
            w = ctx_cdl.empty((self.T, n_ants))
            # PDSCH layers (QPSK) x the slot's precoders: one rank-nu matrix per PRG, rows of the DFT beam set picked per (cell, slot, PRG) -- shape-true stand-in
            # for the W of the scheduled UE's PMI (gNBPhy.m:803,822); the channel input is then nu beams per PRG on 64 antennas, not i.i.d. QPSK per antenna
            ctx_cdl.check(ctx_cdl.lib.isac_synth_qpsk_grid_dev(ctx_cdl.handle, C.c_void_p(layers.ptr), K, 14, self.LAYERS, C.c_uint64(0xD100 + 64 * cell_id + s_), 0))
            beams = rng.integers(0, n_ants, (n_prg, self.LAYERS))
            F = np.stack([dft[b, :] for b in beams], axis=2) * np.sqrt(n_ants / self.LAYERS)          # [nu x A x n_prg], unit power per antenna on average
            self.precoders.append(F); self.layer_seeds.append(0xD100 + 64 * cell_id + s_)
            self.PL.prgPrecodeGrid(layers, F, 0, ctx=ctx_cdl, out=grid)
            ctx_cdl.check(ctx_cdl.lib.isac_ofdm_modulate_dev(ctx_cdl.handle, C.c_void_p(grid.ptr), 14, n_ants, C.byref(car), C.c_double(1.0), C.c_void_p(w.ptr), C.c_int64(self.T)))
            self.waves.append(w)
        self.groups = [[u for u in range(n_ues) if self.los[u]], [u for u in range(n_ues) if not self.los[u]]]
        self.groups = [g for g in self.groups if g]
        self.rx = [[ctx_cdl.empty((self.T, 2)) for _ in range(len(g) * self.SLOTS_PER_CALL)] for g in self.groups]
        self.gains = []
        for g in self.groups:
            st = self.chans[g[0]]._static()
            self.gains.append(ctx_cdl.empty((len(g) * self.SLOTS_PER_CALL * 4 * st.base.shape[0] * st.base.shape[2] * st.base.shape[3],)))    # up to 4 gain blocks per job
        # CSI inputs (setupCSIRS.m:5-23): 4-port row-5 CSI-RS on 273 PRBs, Type-I single panel (2, 1), subband PMI / CQI, 16-PRB subbands
        self.csi_k, self.csi_l = csirs_positions(273)
        self.report = SimpleNamespace(NSizeBWP=273, NStartBWP=0, PanelDimensions=(2, 1), CodebookMode=1, PMIMode="Subband", CQIMode="Subband", SubbandSize=16)
        self.carrier = SimpleNamespace(NSizeGrid=273, NStartGrid=0, SymbolsPerSlot=14)
        self.codebook = self.PL.type1SinglePanelCodebook(self.report, 1, 4)
        r = rng.uniform(30.0, 350.0, n_ues)
        pl_db = 32.4 + 20.0 * np.log10(3.5) + 30.0 * np.log10(np.maximum(r, 10.0))
        self.nvar = 10.0 ** (-(46.0 - pl_db - (-174.0 + 10.0 * np.log10(100e6) + 7.0)) / 10.0)
        self.h_est = [ctx_csi.to_device(freq_response(ch, self.csi_k, K, 30e3, 4)) for ch in self.chans]
        self.h_est_occ = [[ctx_csi.empty(tuple(self.h_est[u].shape)) for u in range(n_ues)] for _ in range(self.CSI_OCCASIONS)]
        # uplink: every UE's packet of a 'U' slot through its UL channel (UE 2 elements -> gNB array), cdl.m:78-85
        self.ul_chans = [CM.CDLChannel(DelayProfile="CDL-D" if lo else "CDL-A", TransmitAntennaArraySize=(1, 1, 2, 1, 1), ReceiveAntennaArraySize=nt_shape, Seed=73) for lo in self.los]
        self.ul_waves, self.ul_rx, self.ul_gains = [], [], []
        if self.WITH_UL:
            g2 = ctx_cdl.empty((K, 14, 2))
            for u in range(n_ues):
                w = ctx_cdl.empty((self.T, 2))
                ctx_cdl.check(ctx_cdl.lib.isac_synth_qpsk_grid_dev(ctx_cdl.handle, C.c_void_p(g2.ptr), K, 14, 2, C.c_uint64(0xE100 + 64 * cell_id + u), 0))
                ctx_cdl.check(ctx_cdl.lib.isac_ofdm_modulate_dev(ctx_cdl.handle, C.c_void_p(g2.ptr), 14, 2, C.byref(car), C.c_double(1.0), C.c_void_p(w.ptr), C.c_int64(self.T)))
                self.ul_waves.append(w)
            for g in self.groups:                         # (one output per (U slot, UE): the frame's four U slots of a group go out in ONE library call)
                st = self.ul_chans[g[0]]._static()
                self.ul_rx.append([ctx_cdl.empty((self.T, n_ants)) for _ in range(self.UL_SLOTS * len(g))])
                self.ul_gains.append(ctx_cdl.empty((self.UL_SLOTS * len(g) * 4 * st.base.shape[0] * st.base.shape[2] * st.base.shape[3],)))
        self.last_cqi = None
        self.last_reports = None
        self.last_ranks = self.last_srs = None
        if self.WITH_UL and self.WITH_SRS:                    # setupSRS.m:8-24: two SRS ports, full bandwidth, 16-PRB measurement subbands (subbandSize.m: 16 or 32 at 273 PRBs)
            self.srs_k1 = np.arange(1, K + 1)
            self.h_srs = [ctx_csi.empty((K, n_ants, 2)) for _ in range(n_ues)]
            self.nvar_ul = 10.0 ** (-(23.0 - pl_db - (-174.0 + 10.0 * np.log10(100e6) + 5.0)) / 10.0)   # 23 dBm UE, 5 dB gNB noise figure
        for c_ in self.ctxs:
            c_.sync()
        ctx_csi.sync()

    SLOTS_PER_CALL = int(os.environ.get("ISAC_C5_SLOTS_PER_CALL", "8"))   # slots of the frame per library call (1 / 2 / 4 / 8 / 16: 137 / 131.6 / 125-127 / 121-123 / 122-123 ms per frame)

    def enqueue_frame(self):
        """All downlink slots of the frame through every UE's channel: one library call per delay profile and SLOTS_PER_CALL consecutive slots (every UE of
        the group appears once per slot, its channel time advancing from slot to slot) -- asynchronous, one contraction + one filter launch per call."""
        for s0 in range(0, self.DL_SLOTS, self.SLOTS_PER_CALL):
            for gi, (g, rx, gn) in enumerate(zip(self.groups, self.rx, self.gains)):
                slots = range(s0, min(s0 + self.SLOTS_PER_CALL, self.DL_SLOTS))
                # (one context per delay-profile group: the filter launch of one group runs beside the contraction launch of the other)
                # (round 6: both delay-profile groups of a cell on ONE context -- with ISAC_OPT_CDL_SHARE_SPECTRA the second group's call reuses the forward transforms of the
                #  first: the UEs of a cell receive the same slot waveforms; ISAC_C5_NO_SHARE=1: one context per group as in rounds 4-5)
                c_ = self.ctxs[(self.cell_id if self.SHARE else gi) % len(self.ctxs)]
                self.CM.applyCDLBatch([self.chans[u] for s_ in slots for u in g], [self.waves[s_] for s_ in slots for u in g], ctx=c_,
                                      outs=[rx[(s_ - s0) * len(g) + i] for s_ in slots for i in range(len(g))], gains=gn)
        if self.WITH_UL:                                      # the frame's four 'U' slots: every UE's packet into the gNB array, one call per group (a channel that
            for gi, g in enumerate(self.groups):              # appears four times advances its time from slot to slot)
                self.CM.applyCDLBatch([self.ul_chans[u] for _ in range(self.UL_SLOTS) for u in g], [self.ul_waves[u] for _ in range(self.UL_SLOTS) for u in g],
                                      ctx=self.ctxs[gi % len(self.ctxs)], outs=self.ul_rx[gi], gains=self.ul_gains[gi])
        # channel times the frame's reports refer to, fixed HERE (the reports may run on another host thread while the next frame's applies advance the channels):
        # the frame's first DL slot (CSI-RS occasions count from it) and the frame's last 'U' slot (SRS)
        self.t_frame = [ch.time - self.DL_SLOTS * self.T / ch.SampleRate for ch in self.chans]
        self.t_srs = [ch.time - self.T / ch.SampleRate for ch in self.ul_chans] if self.WITH_UL else None

    def reports(self, t_frame=None, t_srs=None):
        self.csi_reports(t_frame)
        self.srs_reports(t_srs)

    def csi_reports(self, t_frame=None):
        """The frame's CSI-RS occasions: every UE's report, one batched call (one synchronisation of the CSI context) per occasion."""
        t_frame = t_frame or self.t_frame                    # channel time at the frame's first DL slot (enqueue_frame has advanced the channels past the frame)
        csirs = SimpleNamespace(k=self.csi_k, l=self.csi_l)
        if self.BATCH_OCCASIONS and self.DEVICE_CSI:
            # round 6: the frame's four occasions of the cell in ONE pass -- every (occasion, UE) is an entry of the batched calls (its own channel time, its own estimate):
            # one estimate call per delay-profile group, one report call per rank: 2 synchronisations per cell and frame instead of 8 (each is a host round trip of ~0.2 ms
            # that the GPU does not hide: the applies queued on the other streams do not run ahead of a blocked host)
            for g in self.groups:
                self.CM.csiEstimateBatch([self.chans[u] for o in range(self.CSI_OCCASIONS) for u in g], self.csi_k, 3276, 30e3, 4, ctx=self.ctx_csi,
                                         times=[t_frame[u] + o * 5 * self.T / self.chans[u].SampleRate for o in range(self.CSI_OCCASIONS) for u in g],
                                         outs=[self.h_est_occ[o][u] for o in range(self.CSI_OCCASIONS) for u in g])
            h_all = [self.h_est_occ[o][u] for o in range(self.CSI_OCCASIONS) for u in range(self.n_ues)]
            nvar_all = np.tile(self.nvar, self.CSI_OCCASIONS)
            if self.WITH_RI:
                sel = self.PL.riSelectBatch(self.carrier, csirs, self.report, h_all, nvar_all, DOWNLINK_SINR90PC, ctx=self.ctx_csi)[-self.n_ues:]
                rep, self.last_ranks = [r_[1:] for r_ in sel], [r_[0] for r_ in sel]
            else:
                rep = self.PL.cqiSelectBatch(self.carrier, csirs, self.report, 1, h_all, nvar_all, DOWNLINK_SINR90PC, ctx=self.ctx_csi, codebook=self.codebook)[-self.n_ues:]
        else:
          for o in range(self.CSI_OCCASIONS):
            if self.DEVICE_CSI:                               # this occasion's channel estimates, formed on the device from this occasion's path gains (CSI-RS period
                for g in self.groups:                         # 5 slots): one library call per delay-profile group
                    self.CM.csiEstimateBatch([self.chans[u] for u in g], self.csi_k, 3276, 30e3, 4, ctx=self.ctx_csi,
                                             times=[t_frame[u] + o * 5 * self.T / self.chans[u].SampleRate for u in g], outs=[self.h_est[u] for u in g])
            if self.WITH_RI:                                  # uePhy.m:900-908: riSelect (the PMI search at every rank the two-antenna UE supports), the report at that rank
                sel = self.PL.riSelectBatch(self.carrier, csirs, self.report, self.h_est, self.nvar, DOWNLINK_SINR90PC, ctx=self.ctx_csi)
                rep, self.last_ranks = [r_[1:] for r_ in sel], [r_[0] for r_ in sel]
            else:
                rep = self.PL.cqiSelectBatch(self.carrier, csirs, self.report, 1, self.h_est, self.nvar, DOWNLINK_SINR90PC,
                                             ctx=self.ctx_csi, codebook=self.codebook)
        self.last_cqi = [None if np.isnan(c[0][0]) else int(c[0][0]) for c in rep]
        self.last_reports = rep                              # (cqi, pmi, info) per UE of the frame's last occasion: what the per-cell record carries to rank 0

    def srs_reports(self, t_srs=None):
        """The gNB's uplink measurement (gNBPhy.m:1023-1060): per UE the channel estimate at the SRS symbol of the frame's last 'U' slot -- the perfect estimate,
        formed on the device from that slot's UL path gains (the estimator is toolbox code; as nrChannelEstimate's output it covers every subcarrier) -- through
        pmiSelect (6 TPMIs of the two-port codebook x 3 276 REs x the 64-element array) to TPMI per subband and CQI per RB.  The reference's SRS period is 8 slots
        on a DDDSU pattern (setupSRS.m:13: a UE's SRS meets a 'U' slot once per 40 slots); here once per frame and UE: twice that rate."""
        if not (self.WITH_UL and self.WITH_SRS):
            return
        for g in self.groups:
            self.CM.csiEstimateBatch([self.ul_chans[u] for u in g], self.srs_k1, 3276, 30e3, 2, ctx=self.ctx_csi,
                                     times=[(t_srs or self.t_srs)[u] for u in g], outs=[self.h_srs[u] for u in g])
        self.last_srs = self.PL.srsReportBatch(1, self.h_srs, self.srs_k1 - 1, self.nvar_ul, self.SRS_BAND, 273, UPLINK_SINR90PC, ctx=self.ctx_csi)

    def gemm_launch(self):
        """(jobs, issued 3M flops, bytes) of one contraction launch of the larger delay-profile group: what `roofline` prices."""
        gi = int(np.argmax([len(g) for g in self.groups]))
        n_paths = self.chans[self.groups[gi][0]].path_delays().size
        cols = -(-(2 * n_paths) // 16) * 16
        return gi, len(self.groups[gi]), 6.0 * self.T * cols * self.A * len(self.groups[gi]), n_paths

    def os_mix_flops(self, gi, n_jobs):
        """fp64 flops of one cdl_os_mix_mfma_kernel launch over n_jobs jobs of group gi, counted with ONE gain block per job (a job that crosses a gain refresh forms C(f)
        twice: the count is a lower bound): phase A on the matrix pipe -- per job 2 u x 4 row tiles x ceil(paths / 4) k-steps x 256 bin tiles x 3 MFMAs of 2 x 16 x 16 x 4 flops --,
        phase B on the VALU -- windows x 4096 bins x 128 complex multiply-adds x 8 flops."""
        ch = self.chans[self.groups[gi][0]]
        g, shift = ch.filter_taps()
        mpad = -(-(int(np.max(shift)) + g.shape[1] - 1) // 8) * 8
        n_seg = -(-self.T // (4096 - mpad))
        ksteps = -(-g.shape[0] // 4)
        fa = n_jobs * 2 * 4 * ksteps * 256 * 3 * 2.0 * 16 * 16 * 4
        fb = n_jobs * n_seg * 4096 * 128 * 8.0
        return fa, fb, n_seg


DOWNLINK_SINR90PC = np.array([-3.46, 1.54, 6.54, 11.05, 13.54, 16.04, 17.54, 20.04, 22.04, 24.43, 26.93, 27.43, 29.43, 32.43, 35.43])   # setupSINRtoCQIMappingTable.m:7-11
UPLINK_SINR90PC = np.array([-5.46, -0.46, 4.54, 9.05, 11.54, 14.04, 15.54, 18.04, 20.04, 22.43, 24.93, 25.43, 27.43, 30.43, 33.43])


def run_config5(args, pkg, rank, world, local_rank, dist, torch):
    """`--workload config5`: cells sharded cell c -> rank c mod world; per frame and cell one sensing CPI (16 slots, pipelined over the SlotPool),
    the frame's CDL applies and CSI reports.  A step = one 20-slot frame of every cell of the rank.  value = cells x 20 slots x frames / time."""
    d = importlib.import_module(PKG + "._dist")
    n_cells = args.cells if args.cells > 0 else 21
    mine = d.shard_cells(n_cells, rank, world)
    pool = SlotPool(pkg, local_rank, args.inflight)
    n_cdl_ctx = int(os.environ.get("ISAC_C5_CDL_CONTEXTS", "2"))
    ctxs_cdl, ctx_csi = [pkg.Context(local_rank) for _ in range(max(1, n_cdl_ctx))], pkg.Context(local_rank)
    ctx_cdl = ctxs_cdl[0]
    if CommCell.SHARE:
        for c_ in ctxs_cdl:
            c_.set_cdl_share_spectra(True)
    n_buf = -(-args.inflight // max(len(mine), 1))
    lazy_native = 48 < args.ants <= 64 and args.targets <= 2 and args.echo != "array"      # the sensing CPIs keep their echo grids lazy where that is native (bench --echo)
    sense = [Cell(pkg, local_rank, c, args.ants, args.slots, args.targets, pool=pool, n_buf=min(n_buf, 2), lazy=lazy_native) for c in mine]
    comm = [CommCell(pkg, ctxs_cdl, ctx_csi, c, args.ants, args.ues) for c in mine]

    def barrier():
        pool.sync(); ctx_csi.sync()
        for c_ in ctxs_cdl:
            c_.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # The reports of a cell (CSI: riSelect + cqiSelect of its four occasions; SRS) are host round trips -- uploads, small launches, a copy back, a synchronisation, the
    # host half of the report -- that a GPU busy with the applies does not hide while the issuing thread is blocked in them.  They run on a second HOST thread
    # (the reference's cells are parallel workers, networkSimulation.m:47-60; ctypes drops the GIL inside the library), one cell at a time, on their own context;
    # the main thread keeps enqueueing sensing CPIs and applies.  A frame's reports are all waited for before the timed region ends.  ISAC_C5_REPORT_THREAD=0: inline.
    from concurrent.futures import ThreadPoolExecutor
    threaded = os.environ.get("ISAC_C5_REPORT_THREAD", "1") != "0"
    reporter = ThreadPoolExecutor(max_workers=1) if threaded else None
    pending = []

    def frame():
        for sc, cc in zip(sense, comm):
            pool.submit(sc)
            cc.enqueue_frame()
            if reporter is not None:
                pending.append(reporter.submit(cc.reports, cc.t_frame, cc.t_srs))      # (the frame's time stamps travel with the job: the next frame's enqueue overwrites them)
            else:
                cc.reports()

    def reports_done():
        while pending:
            pending.pop().result()

    for _ in range(args.warmup):
        frame()
    pool.drain()
    reports_done()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame()
    pool.drain()
    reports_done()
    barrier()
    dt = time.perf_counter() - t0
    if reporter is not None:
        reporter.shutdown(wait=True)
    os.write(2, f"bench.py: rank {rank}/{world}: device {local_rank}, {len(mine)} cell(s) {mine}, timed region {1e3 * dt:.3f} ms for {args.steps} frame(s), config5\n".encode())
    recs = np.array([d.make_record(cid, sc.last, dt, ue_reports=cc.last_reports, ue_ranks=cc.last_ranks, ue_srs=cc.last_srs, srs_band=CommCell.SRS_BAND) for cid, sc, cc in zip(mine, sense, comm)]).reshape(-1, d.RECORD_LEN)
    on_gpu = dist is not None and dist.get_backend() == "nccl"
    allr = d.gather_records(recs, dist, torch.device("cuda", local_rank) if on_gpu else None)
    dt_max = float(np.nanmax(allr[:, 6])) if allr.size else dt
    if rank != 0:
        return
    # ---- the contraction kernel with the device to itself: 10 launches of the larger group's batch, HIP events around exactly that launch
    cc = comm[0]
    gi, n_jobs, flops, n_paths = cc.gemm_launch()
    ctx_cdl.check(ctx_cdl.lib.isac_profile_enable(ctx_cdl.handle, 1))
    ms_k, ms_call = [], []
    # (the batch the frame loop issues: SLOTS_PER_CALL consecutive slots x the UEs of the group -- the persistent grid of the fused kernel is priced on the
    #  launch shape it actually runs with; a one-slot batch would spend a quarter of its tiles on warm-up)
    n_slots = min(CommCell.SLOTS_PER_CALL, CommCell.DL_SLOTS)
    n_jobs, flops = n_jobs * n_slots, flops * n_slots
    for i in range(11):
        ctx_cdl.sync(); ctx_cdl.timer_start()
        cc.CM.applyCDLBatch([cc.chans[u] for s_ in range(n_slots) for u in cc.groups[gi]], [cc.waves[(i + s_) % cc.DL_SLOTS] for s_ in range(n_slots) for u in cc.groups[gi]],
                            ctx=ctx_cdl, outs=cc.rx[gi][:n_jobs], gains=cc.gains[gi])
        v = C.c_double(0.0)
        ctx_cdl.check(ctx_cdl.lib.isac_profile_last_kernel_ms(ctx_cdl.handle, C.byref(v)))
        ms_call.append(ctx_cdl.timer_stop_ms()); ms_k.append(v.value)
    ms_k, ms_call = float(np.mean(ms_k[1:])), float(np.mean(ms_call[1:]))
    t1 = time.perf_counter()
    for _ in range(3):
        cc.csi_reports()
    ms_csi = 1e3 * (time.perf_counter() - t1) / (3 * cc.CSI_OCCASIONS * cc.n_ues)
    n_applies = sum(c_.n_ues for c_ in comm) * CommCell.DL_SLOTS
    time_domain = bool(os.environ.get("ISAC_CDL_TIME_DOMAIN") or os.environ.get("ISAC_CDL_UNFUSED")) or args.ants != 64
    if time_domain:
        roof = {"bound": "mfma", "kernel": "cdl_fused_kernel<NCT,NSLOT> (DL apply of a batch in one persistent launch: contraction X [T x Nt] against the path gains of every job, 3M form on "
                                           "v_mfma_f64_16x16x4_f64, + 16-tap delay filters + integer delays on the CU; Z never in HBM)" if not os.environ.get("ISAC_CDL_UNFUSED") else
                                           "cdl_gemm_kernel<NCT,false> (DL contraction of a batch; ISAC_CDL_UNFUSED: the delay filter is a second launch, Z through HBM)",
                "achieved": round(flops / 1e12 / (ms_k / 1e3), 2), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(flops / 1e12 / (ms_k / 1e3) / FP64_MFMA_PEAK_TFLOPS, 4),
                "traffic": None, "avg_launch_ms": round(ms_k, 4), "launches_averaged": 10, "jobs_per_launch": n_jobs, "paths": n_paths,
                "issued_flops_per_launch": flops, "flops_note": "contraction only (the launch also runs the delay filters on the VALU: + 16 % flops, not counted): 3 real MFMAs x 2 flops per complex multiply-add, padded to 16-column tiles: 6 T cols Nt per job",
                "whole_batch_call_ms": round(ms_call, 4), "whole_batch_note": "path gains + apply of the same batch, HIP events around the call",
                "timing": "HIP events recorded by the library around the apply launch (isac_profile_*), device otherwise idle"}
    else:
        fa, fb, n_seg = cc.os_mix_flops(gi, n_jobs)
        roof = {"bound": "mfma", "kernel": "cdl_os_mix_mfma_kernel (overlap-save DL apply, the arithmetic of the path: per 16-bin tile C(f) = sum_n H_n E_n(f) on v_mfma_f64_16x16x4_f64 -- 3M complex form --, "
                                           "then Y(f) = C(f) X(f) for every 4096-sample window on the fp64 VALU; forward / inverse transforms are separate HBM-bound launches)",
                "achieved": round((fa + fb) / 1e12 / (ms_k / 1e3), 2), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round((fa + fb) / 1e12 / (ms_k / 1e3) / FP64_MFMA_PEAK_TFLOPS, 4),
                "traffic": None, "avg_launch_ms": round(ms_k, 4), "launches_averaged": 10, "jobs_per_launch": n_jobs, "paths": n_paths, "windows": n_seg,
                "issued_flops_per_launch": fa + fb, "mfma_flops": fa, "valu_flops": fb,
                "flops_note": "matrix-pipe flops of C(f) (three real MFMAs per complex product, paths padded to a multiple of four) + vector flops of Y(f) (8 per complex multiply-add), one gain "
                              "block per job (a lower bound); the part's fp64 vector and matrix peaks are the same 78.6 TFLOP/s.  The time-domain formulation of rounds 4-5 issued 1.09 GF per "
                              "job on the matrix pipe for the same result: this launch needs %.2f GF per job" % ((fa + fb) / n_jobs / 1e9),
                "whole_batch_call_ms": round(ms_call, 4), "whole_batch_note": "path gains + forward transforms + mix + inverse transforms of the same batch, HIP events around the call",
                "timing": "HIP events recorded by the library around the mix launch (isac_profile_*), device otherwise idle"}
    res = {"metric": "sensing slots/sec (CDL echo->2D-FFT->2D-CFAR)", "value": round(n_cells * 20 * args.steps / dt_max, 2), "unit": "slots/sec (whole cells: sensing CPI + CDL applies + CSI reports)",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt_max / args.steps, 3), "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"BASELINE configs[4]: {n_cells} cells x {args.ues} UE, {args.ants}-antenna gNB; per 20-slot frame and cell: one sensing CPI ({args.slots} slots, "
                                  f"echo -> 2D-FFT -> 2D-CFAR -> MUSIC), {CommCell.DL_SLOTS} DL slot waveforms [{cc.T} x {args.ants}] (precoded PDSCH: {CommCell.LAYERS} layers, one precoder per "
                                  f"{CommCell.PRG_PRBS}-PRB PRG) through every UE's CDL-D / CDL-A channel ({CommCell.DL_SLOTS * args.ues} applies), "
                                  + (f"{CommCell.UL_SLOTS} UL slots x every UE's packet [{cc.T} x 2] into the {args.ants}-element array ({CommCell.UL_SLOTS * args.ues} applies), " if CommCell.WITH_UL else "")
                                  + f"{CommCell.CSI_OCCASIONS} CSI reports per UE ("
                                  + ("riSelect over ranks 1-2 + " if CommCell.WITH_RI else "") + "Type-I PMI search + subband CQI, 546 CSI-RS REs x 32 entries; channel estimate of each occasion "
                                  f"{'formed on the device from its path gains' if CommCell.DEVICE_CSI else 'evaluated once on the host'})"
                                  + (f", one SRS measurement per UE (pmiSelect: 6 TPMIs x 3276 REs x {args.ants} receive elements -> TPMI per {CommCell.SRS_BAND}-PRB subband + CQI per RB)" if CommCell.WITH_UL and CommCell.WITH_SRS else ""),
                      "parallelism": f"cells sharded over {world} GPU(s)"},
           "per_frame_and_rank": {"cells": len(mine), "cdl_applies": n_applies, "csi_reports": len(mine) * args.ues * CommCell.CSI_OCCASIONS, "sensing_cpis": len(mine),
                                  "ul_applies": sum(c_.n_ues for c_ in comm) * CommCell.UL_SLOTS if CommCell.WITH_UL else 0, "precoded": True,
                                  "csi_h": "device, per occasion" if CommCell.DEVICE_CSI else "host, once at set-up",
                                  "rank_selection": CommCell.WITH_RI, "srs_reports": sum(c_.n_ues for c_ in comm) if CommCell.WITH_UL and CommCell.WITH_SRS else 0},
           "roofline": roof,
           "comm_seams": {"cdl_apply_ms_per_job": round(ms_call / n_jobs, 4), "csi_report_ms_per_ue": round(ms_csi, 4),
                          "csi_note": "host wall per UE of the batched report (one synchronisation per cell and occasion), device otherwise idle"},
           "cells": [d.record_json(r) for r in allr[:64]],
           "cells_note": "gathered from every rank (one all_gather of fixed-size records, _dist.py): every cell's whole estResults (fft2D.m:102,114-115) and every UE's last "
                         "CSI report of the frame (wideband CQI, subband CQIs, PMI i1 / i2: networkSimulation.m:173-232)"}
    print(json.dumps(res))


def stage_table(cell, reps=5):
    """Isolated per-stage durations (HIP events on the context stream, one CPI resident, nothing else running) with each
    stage's own roofline -- so that the JSON line shows every large kernel, not only the one quoted in `roofline`."""
    c = cell.ctx
    b = 16
    K, L, A, T = cell.K, cell.Lsym, cell.A, cell.T

    def timed(fn):
        fn(); c.sync()
        c.timer_start()
        for _ in range(reps):
            fn()
        return c.timer_stop_ms() / reps

    def mono(noise, fuse):
        kw = dict(seed=cell.seed if noise else None, noise_domain=cell.noise_domain, nfft=4096, out=cell.echo[0], ctx=c,
                  fuse_fft2d=(cell.rp, cell.cfar, cell.tx_grid) if fuse else None)
        return lambda: cell.pkg.sensing.monoStaticSensing(cell.tx_wave, (K, L, A), cell.carrier, cell.rp, cell.los, **kw)

    ra = c.empty((A, A))
    cov = lambda: c.check(c.lib.isac_covariance_dev(c.handle, C.c_void_p(cell.echo[0].ptr), C.c_int64(K * L), C.c_int32(A), C.c_void_p(ra.ptr)))
    out = []
    echo_b = T * A * b + K * L * A * b
    fused_b = echo_b + K * L * A * b
    for name, fn, nb, note in (
            ("monoStaticSensing + fft2D range stage (beam-sum, coefficient vectors, per-target demodulation, fused synthesis + range kernel), Philox AWGN",
             mono(True, True), fused_b, "reads txWaveform + txGrid once, writes echoGrid once; the kernel in `roofline` is the last launch of this stage"),
            ("monoStaticSensing alone (unfused), Philox AWGN", mono(True, False), echo_b, "read txWaveform once, write echoGrid once"),
            ("monoStaticSensing alone, noise off", mono(False, False), echo_b, "same traffic without the generator")):
        ms = timed(fn)
        out.append({"stage": name, "ms": round(ms, 4), "bound": "hbm", "algorithmic_bytes": nb,
                    "achieved_GBps": round(nb / 1e9 / (ms / 1e3), 1), "frac": round(nb / 1e9 / (ms / 1e3) / HBM_PEAK_GBS, 4), "note": note})
    ms = timed(cov)
    fl = 8.0 * A * A * K * L
    nbk = (A + 15) // 16
    issued = fl * (nbk * (nbk + 1) / 2) / (nbk * nbk) * 0.75    # Hermitian: upper-triangular 16 x 16 tiles only; 3 real MFMAs per complex tile step (3M form)
    out.append({"stage": "covariance Ra = X X^H / N (fp64 MFMA)", "ms": round(ms, 4), "bound": "mfma",
                "issued_flops": issued, "achieved_TFLOPs": round(issued / 1e12 / (ms / 1e3), 2), "peak_TFLOPs": FP64_MFMA_PEAK_TFLOPS,
                "frac": round(issued / 1e12 / (ms / 1e3) / FP64_MFMA_PEAK_TFLOPS, 4), "nominal_flops": fl,
                "note": "frac counts ISSUED MFMA flops (upper-triangular tiles: %d of %d, three real products per complex tile step instead of four); the nominal 8 A^2 K L count would credit the skipped work" % (nbk * (nbk + 1) // 2, nbk * nbk)})
    return out


def cpu_baseline(cell, budget_s=12.0):
    """The reference's chain on the host cores: the C++17 / OpenMP port under oracle/cpu_port (kind "port": the MATLAB reference cannot
    run here -- no MATLAB, no toolboxes).  Same workload as the GPU line at FULL size (all antennas, all 224 symbols): the transmit
    waveform and grid are copied back from the device, the port draws its own AWGN (inside the timed call, as randn is inside
    basicRadarChannel), every OpenMP thread the host offers.  Bounded sample: whole CPIs until `budget_s` seconds have elapsed
    (at least 3, at most 16), median per-CPI time."""
    from oracle import cpu_port as P
    tx_wave, tx_grid = cell.tx_wave.numpy(), cell.tx_grid.numpy()
    cf = SimpleNamespace(CUTIdx=cell.cfar.CUTIdx, Pfa=cell.cfar.cfarDetector2D.ProbabilityFalseAlarm,
                         GuardBandSize=tuple(cell.cfar.cfarDetector2D.GuardBandSize), TrainingBandSize=tuple(cell.cfar.cfarDetector2D.TrainingBandSize))
    times, t_start, est = [], time.perf_counter(), None
    echo = np.empty(tx_grid.shape, dtype=np.complex128, order="F")          # the CPU implementation keeps its echo grid from CPI to CPI
    while len(times) < 3 or (time.perf_counter() - t_start < budget_s and len(times) < 16):
        t0 = time.perf_counter()
        echo = P.mono_static_sensing(tx_wave, tx_grid.shape, cell.carrier, cell.rp, cell.los, None, nfft=4096, seed=0x5EED0002 + len(times), out=echo)
        try:
            est = P.fft2d(cell.rp, cf, echo, tx_grid)
        except ValueError:
            est = None
        times.append(time.perf_counter() - t0)
    cpi_s = float(np.median(times))
    n_slots = cell.Lsym // 14
    port = {"value": round(n_slots / cpi_s, 3), "unit": "sensing slots/sec", "cores": P.threads(), "kind": "port", "language": "C++17 + OpenMP (oracle/cpu_port)",
            "tuning": "un-tuned port (plain OpenMP loops, own radix-4 FFT, no BLAS): a reported baseline, not evidence of kernel quality",
            "cpi_s": {"median": round(cpi_s, 4), "min": round(float(np.min(times)), 4), "max": round(float(np.max(times)), 4), "n": len(times)},
            "sample": f"{len(times)} whole CPIs of the bench workload ({cell.A} antennas, K={cell.K}, L={cell.Lsym}, T={cell.T}) through oracle/cpu_port "
                      f"(C++17 + OpenMP, own radix-4 FFT, fp64, AWGN drawn inside the timed call), median {cpi_s:.3f} s per CPI, "
                      f"{P.threads()} OpenMP threads; first estimates rng {None if est is None else np.round(est.rngEst[:2], 3).tolist()} "
                      f"azi {None if est is None else est.aziEst[:2].tolist()}",
            "note": "the MATLAB reference itself cannot be timed (no MATLAB / toolboxes on this host)"}
    # BASELINE.md section 2 names two CPU implementations: the NumPy / SciPy oracle (pocketfft with every worker thread, OpenBLAS for the covariance and
    # the eigendecomposition -- the libraries closest to MATLAB's FFTW / MKL) is timed as well, on ONE whole CPI of the same workload, AWGN draw included
    # (standard_normal, as randn sits inside basicRadarChannel.m:67-69).  The faster of the two is `value`: the baseline is not the weaker implementation.
    import oracle as O
    rp_o = O.radar_params(cell.cellp, cell.carrier, cell.wave) if hasattr(O, "radar_params") else cell.rp
    cf_o = O.cfar2d_config(rp_o)
    t0 = time.perf_counter()
    rng = np.random.default_rng(0x5EED0002)
    noise = np.empty((cell.T, cell.A), dtype=np.complex128, order="F")
    noise.real = rng.standard_normal((cell.A, cell.T)).T
    noise.imag = rng.standard_normal((cell.A, cell.T)).T
    echo_o = O.mono_static_sensing(tx_wave, tx_grid.shape, cell.carrier, rp_o, cell.los, noise, nfft=4096)
    del noise
    try:
        est_o = O.fft2d(rp_o, cf_o, echo_o, tx_grid, rdm_fn=O.rdm_explicit)
    except ValueError:
        est_o = None
    np_s = time.perf_counter() - t0
    oracle_leg = {"value": round(n_slots / np_s, 3), "unit": "sensing slots/sec", "kind": "port", "language": "NumPy / SciPy oracle (oracle/*.py: scipy.fft with workers = all cores, OpenBLAS)",
                  "cpi_s": round(np_s, 3), "n": 1, "cores": os.cpu_count(),   # (`cores` everywhere in this object = the THREADS the leg actually ran on)
                  "first_estimates": None if est_o is None else {"rng": np.round(est_o.rngEst[:2], 3).tolist(), "azi": est_o.aziEst[:2].tolist()}}
    best = port if port["value"] >= oracle_leg["value"] else dict(oracle_leg, tuning="NumPy / SciPy restatement", sample=f"one whole CPI through the NumPy / SciPy oracle, {np_s:.2f} s", note=port["note"])
    best = dict(best)
    try:
        import psutil
        phys = psutil.cpu_count(logical=False)
    except Exception:                                        # noqa: BLE001
        phys = None
    best["host"] = {"logical_cpus": os.cpu_count(), "physical_cores": phys,
                    "cores_convention": "`cores` = threads the timed leg used: the OpenMP port runs one thread per PHYSICAL core (OMP default under the container's "
                                        "affinity mask), the NumPy / SciPy leg hands every LOGICAL cpu to scipy.fft workers / OpenBLAS"}
    best["implementations"] = {"cpp_openmp_port": {k: port[k] for k in ("value", "cores", "cpi_s", "language")}, "numpy_scipy_oracle": oracle_leg,
                               "note": "BASELINE.md section 2: both CPU implementations timed on this host in this run; `value` is the faster one"}
    return best


def roofline_entry(cell, args, ms_timed, ms_iso, n_launches, stages, cpi_bytes, per_cpi_ms, ms_cov_iso=None):
    """`roofline` for the dominant HBM-bound kernel: the fused echo-synthesis + range-stage kernel (the largest share of the wide kernels'
    time and of the bytes moved).  achieved = algorithmic bytes per launch / average launch duration measured with HIP events inside
    this run; `top_by_time` and `traffic` come from the committed rocprofv3 summaries (profile_facts)."""
    facts = profile_facts(args.lazy)
    whole = {"algorithmic_bytes": cpi_bytes, "ms": round(per_cpi_ms, 4), "achieved_GBps": round(cpi_bytes / 1e9 / (per_cpi_ms / 1e3), 1),
             "frac": round(cpi_bytes / 1e9 / (per_cpi_ms / 1e3) / HBM_PEAK_GBS, 4),
             "note": "SURVEY 8d bytes per CPI (txWaveform + echoGrid + rxGrid + txGrid) / driver-visible time per CPI"}
    if not args.fuse or ms_iso is None:
        return {"bound": "hbm", "kernel": None, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                "top_by_time": facts["top_by_time"],
                "note": "--no-fuse: per-kernel event timing is wired to the fused kernel only", "other_stages": stages, "whole_cpi": whole}
    nb = cell.dominant_kernel_bytes()
    # The launch duration that defines the kernel's own roofline fraction is the one with the device to itself: in the timed region up to
    # `--inflight` CPIs share the GPU, so an event pair around one launch also spans the time the device spends on the other CPIs' kernels
    # (reported separately below; the end-to-end figure under concurrency is `whole_cpi`).
    ms = ms_iso
    at_shape = (args.ants == 64 and args.slots == 16 and args.targets == 1)
    cov_entry = None
    if ms_cov_iso:
        issued, nominal = cell.covariance_issued_flops()
        cov_name = ("cov_lazy_kernel<Q,2> (operand tiles re-formed from D, a, seed: Philox rounds / Box-Muller / synthesis in the MFMA gaps)" if args.lazy else
                    "cov_mfma_lds_kernel<4>" if 48 < args.ants <= 64 else "cov_mfma_small_kernel" if args.ants <= 64 else "cov_mfma_block_pl_kernel (64 x 64 block pairs)")
        cov_entry = {"bound": "mfma", "kernel": cov_name + ": Ra = X X^H / N on v_mfma_f64_16x16x4_f64, 3M form, upper-triangular tiles (fft2D.m:106-107)",
                     "achieved": round(issued / 1e12 / (ms_cov_iso / 1e3), 2), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(issued / 1e12 / (ms_cov_iso / 1e3) / FP64_MFMA_PEAK_TFLOPS, 4), "traffic": facts.get("traffic_cov") if at_shape else None,
                     "traffic_source": facts["traffic_source"] if at_shape else None,
                     "avg_launch_ms": round(ms_cov_iso, 4), "launches_averaged": 10,
                     "avg_launch_ms_rocprofv3": None if not (at_shape and facts["top_by_time"] and facts["top_by_time"].get("covariance_kernel_avg_us")) else round(facts["top_by_time"]["covariance_kernel_avg_us"] / 1e3, 4),
                     "duration_note": "two clocks for the same launch: the library's HIP event pair in THIS run (`avg_launch_ms`, what `achieved` / `frac` use) and the kernel duration of the committed "
                                      "rocprofv3 trace (`avg_launch_ms_rocprofv3`, another box); the event pair reads 10-15 % longer (it includes the launch gaps either side of the kernel)",
                     "issued_flops_per_launch": issued, "nominal_flops_per_launch": nominal,
                     "flops_note": "issued = 8 A^2 K L x (upper-triangular 16 x 16 tiles / all tiles) x 3/4 (three real MFMAs per complex tile step); masked padding rows not counted",
                     "timing": "HIP events recorded by the library around exactly this launch (isac_profile_enable(ctx, 2)), blocking CPIs, device otherwise idle"}
    if cov_entry and ms_cov_iso > ms:
        # the LONGEST launch of the CPI is the covariance (lazy echo grid: the fused kernel no longer stores echoGrid, the covariance pays for the generator; more than 64
        # antennas: the block-pair kernel): `roofline` describes that kernel, the fused kernel follows as `second_kernel`
        facts_cov = dict(facts["top_by_time"]) if isinstance(facts.get("top_by_time"), dict) else facts.get("top_by_time")
        cov_entry.update({"top_by_time": facts_cov, "other_stages": stages, "whole_cpi": whole,
                          "second_kernel": {"bound": "hbm", "kernel": "echo_range_sl_kernel (fused synthesis + range stage" + (", echoGrid NOT stored" if args.lazy else "") + ")",
                                            "avg_launch_ms": round(ms, 4), "algorithmic_bytes_per_launch": nb, "achieved_GBps": round(nb / 1e9 / (ms / 1e3), 1),
                                            "frac": round(nb / 1e9 / (ms / 1e3) / HBM_PEAK_GBS, 4),
                                            "algorithmic_bytes_note": "txGrid read once (lazy: no echoGrid store)" if args.lazy else "txGrid read once + echoGrid written once"}})
        return cov_entry
    return {"bound": "hbm", "kernel": ("echo_range_sl_kernel" if args.targets <= 2 else "echo_range_kernel") + f"<{args.targets if args.targets <= 4 else 0},1> (fused: per-target rank-1 echo synthesis + Philox AWGN on the demodulated grid -> echoGrid; "
                                      "rx.*conj(tx), Kaiser, 4096-pt range IFFT, CUT rows)",
            "achieved": round(nb / 1e9 / (ms / 1e3), 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(nb / 1e9 / (ms / 1e3) / HBM_PEAK_GBS, 4),
            "traffic": facts["traffic"] if at_shape else None,
            "traffic_source": facts["traffic_source"] if at_shape else None,
            "top_by_time": facts["top_by_time"],
            "avg_launch_ms": round(ms, 4), "launches_averaged": 10,
            "avg_launch_ms_rocprofv3": None if not (at_shape and facts["top_by_time"] and facts["top_by_time"].get("dominant_hbm_kernel_avg_us")) else round(facts["top_by_time"]["dominant_hbm_kernel_avg_us"] / 1e3, 4),
            "frac_on_rocprofv3_duration": None if not (at_shape and facts["top_by_time"] and facts["top_by_time"].get("dominant_hbm_kernel_avg_us")) else round(nb / 1e9 / (facts["top_by_time"]["dominant_hbm_kernel_avg_us"] / 1e6) / HBM_PEAK_GBS, 4),
            "duration_note": "two clocks for the same launch: the library's HIP event pair in THIS run (`avg_launch_ms`, what `achieved` / `frac` use) and the kernel duration of the committed "
                             "rocprofv3 trace (`avg_launch_ms_rocprofv3`, another box); the event pair reads 10-15 % longer (it includes the launch gaps either side of the kernel)",
            "avg_launch_ms_in_timed_region": None if not ms_timed else round(ms_timed, 4), "launches_in_timed_region": n_launches,
            "frac_in_timed_region": None if not ms_timed else round(nb / 1e9 / (ms_timed / 1e3) / HBM_PEAK_GBS, 4),
            "timing": "HIP events recorded by the library around every launch of this kernel, on the stream of the launch: `avg_launch_ms` with the "
                      "device to itself (10 launches right after the timed region; this is what rocprofv3 reports for the single-stream run, "
                      "profiles/rNN_kernel_stats_single_stream.csv), `..._in_timed_region` with the other in-flight CPIs' kernels sharing the GPU",
            "algorithmic_bytes_per_launch": nb, "algorithmic_bytes_note": ("LAZY echo grid: txGrid read once = K L A 16 B (no echoGrid store)" if args.lazy else "txGrid read once + echoGrid written once = 2 K L A 16 B; echoGrid is not re-read by the range stage"),
            "covariance_kernel": cov_entry,
            # the same bytes moved by plain copy kernels on this part (1 : 1 read / write mix), tools/cbench.hip: the guide's flat form (one float4 per thread) and the
            # kernel's own column shape.  Round 4 quoted a persistent grid-stride probe (5.35-5.53 TB/s) -- that was the probe's shape, not the part's limit.
            "copy_rate_reference": {"GBps": COPY_RATE_GBS, "frac_of_copy_rate": round(nb / 1e9 / (ms / 1e3) / COPY_RATE_GBS, 4),
                                    "frac_of_copy_rate_on_counter_traffic": round(facts["traffic"] / 1e9 / (ms / 1e3) / COPY_RATE_GBS, 4) if (at_shape and facts["traffic"]) else None,
                                    "column_shaped_copy_GBps": COPY_RATE_COLUMNS_GBS,
                                    "frac_of_column_shaped_copy_on_counter_traffic": round(facts["traffic"] / 1e9 / (ms / 1e3) / COPY_RATE_COLUMNS_GBS, 4) if (at_shape and facts["traffic"]) else None,
                                    "source": "profiles/r05_cbench_copy_rate.txt: a flat float4 copy (n / 256 short-lived workgroups: MI355X_MICROARCH.md's 6.29 TB/s form) moves the "
                                              "kernel's 0.75 GB -> 0.75 GB at 6.26 TB/s; one workgroup per 52 416-byte column -- the shape a per-column transform must have -- at 5.73; "
                                              "persistent grid-stride copies and hipMemcpyAsync at 5.0-5.9.  `frac` above prices the same launch against the 8 TB/s specification"},
            "other_stages": stages, "whole_cpi": whole}


def cold_probe(args, pkg):
    """`--cold-probe MODE` (internal; run by the main bench in a FRESH process): what the drop-in's real caller sees.  The reference runs the sensing chain
    once per cell and simulation (cellSimulation.m:189-202, one worker per cell: networkSimulation.m:47-60), so the first calls of a process matter.
    Inputs are made resident (the host would upload senTxWave), the device is left idle for a second, then
      MODE = reserve:   isac_ctx_reserve(warm_ms) and 21 BLOCKING CPIs (config 5's one sensing pass per cell), each timed on the host;
      MODE = noreserve: the 21 blocking CPIs straight away (first call loads code objects, sizes scratch, builds tables, at idle clocks).
    Prints one JSON object."""
    t_proc = time.perf_counter()
    cell = Cell(pkg, 0, 0, args.ants, args.slots, args.targets, inflight=1, fuse=True, n_buf=1)
    cell.sync()
    t_inputs = 1e3 * (time.perf_counter() - t_proc)
    time.sleep(1.0)                                           # back to an idle device
    out = {"mode": args.cold_probe, "inputs_resident_ms": round(t_inputs, 1)}
    if args.cold_probe == "reserve":
        t0 = time.perf_counter()
        lib_ms = pkg.sensing.reserve(cell.T, (cell.K, cell.Lsym, cell.A), cell.carrier, cell.rp, cell.cfar, nfft=4096, warm_ms=args.cold_warm_ms, ctx=cell.ctx)
        out["reserve_ms"] = round(1e3 * (time.perf_counter() - t0), 2)
        out["reserve_warm_ms_requested"] = args.cold_warm_ms
        out["reserve_ms_library_clock"] = round(lib_ms, 2)
    per = []
    gc.collect(); gc.disable()                                # (the caller of the drop-in is MATLAB: no Python collector pauses in its figures)
    for _ in range(21):
        t0 = time.perf_counter()
        cell.step()
        per.append(1e3 * (time.perf_counter() - t0))
    gc.enable()
    out.update({"first_cpi_ms": round(per[0], 3), "cpi_2_21_ms": {"median": round(float(np.median(per[1:])), 3), "max": round(float(np.max(per[1:])), 3)},
                "wall_21_blocking_cpis_ms": round(float(np.sum(per)), 2), "per_cpi_ms": [round(v, 3) for v in per]})
    # the steady blocking CPI of the same process (clocks up): the yardstick for the figures above
    for _ in range(200):
        cell.step()
    t0 = time.perf_counter()
    for _ in range(20):
        cell.step()
    out["steady_blocking_cpi_ms"] = round(1e3 * (time.perf_counter() - t0) / 20, 3)
    print(json.dumps(out))


def cold_block(args):
    """Run the cold probe in fresh processes (this process's device state must not help them) and return the `cold` object of the JSON line."""
    import subprocess
    base = [sys.executable, os.path.abspath(__file__), "--ants", str(args.ants), "--slots", str(args.slots), "--targets", str(args.targets), "--cold-warm-ms", str(args.cold_warm_ms)]
    res = {}
    for mode in ("reserve", "noreserve"):
        try:
            r = subprocess.run(base + ["--cold-probe", mode], capture_output=True, text=True, timeout=180)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            res[mode] = json.loads(line[-1]) if (r.returncode == 0 and line) else {"error": (r.stderr or r.stdout)[-300:]}
        except Exception as e:                                # noqa: BLE001 -- a probe that cannot run must not take the bench line down
            res[mode] = {"error": repr(e)}
    res["note"] = ("fresh process each, inputs resident, device idle for 1 s, then 21 blocking CPIs (submit + collect, the reference's call order; config 5 runs one "
                   "sensing pass per cell): `reserve` = after isac_ctx_reserve(warm_ms) -- what a host that calls it during scenario set-up gets; `noreserve` = the first "
                   "call pays for code objects, scratch, tables and idle clocks")
    return res


def respawn_under_torchrun(n_gpus):
    """`python bench.py --gpus N` without a torchrun environment launches its own N ranks (one per GPU, RCCL) on this node."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--ants", type=int, default=64)
    ap.add_argument("--slots", type=int, default=16)
    ap.add_argument("--targets", type=int, default=1)
    ap.add_argument("--cells-per-gpu", type=int, default=1)
    ap.add_argument("--cells", type=int, default=0, help="total number of cells, sharded cell c -> rank c mod world (BASELINE configs[2]: 7 cells on "
                                                         "2/4/8 GPUs, inherently imbalanced); 0 = --cells-per-gpu cells on every rank (weak scaling)")
    ap.add_argument("--inflight", type=int, default=8, help="CPIs in flight per cell (contexts)")
    ap.add_argument("--no-fuse", dest="fuse", action="store_false", help="separate monoStaticSensing and fft2D range kernels (the range stage re-reads echoGrid)")
    ap.add_argument("--noise-domain", choices=("spectral", "time"), default="spectral",
                    help="Philox AWGN drawn on the demodulated grid (default; same distribution, include/isac.h isac_noise_mode) or per time sample")
    ap.add_argument("--echo", choices=("lazy", "array", "auto"), default="auto",
                    help="lazy: the echo grid stays inside the context as a descriptor -- the fused kernel skips its store, the covariance kernel re-forms its operands "
                         "(include/isac.h 'LAZY echo grid': spectral Philox noise, 49..64 antennas, <= 2 targets; other shapes fall back to a context-owned buffer); "
                         "array: monoStaticSensing writes echoGrid, fft2D's covariance reads it back (rounds 1-5); auto (default): lazy where it is native")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prime-ms", type=float, default=300.0, help="untimed device priming (hot-path steps) before the warm-up steps; 0 = none")
    ap.add_argument("--pace-ms", type=float, default=-1.0, help="minimum host-side spacing of consecutive CPI submissions; 0 = none, < 0 (default) = 0.93 x the "
                                                                 "un-paced time per CPI measured during the priming phase.  CPIs submitted in a burst advance in lockstep on the GPU "
                                                                 "(round-robin dispatch among their queues) and finish together, so their narrow MUSIC tails leave the GPU nearly idle "
                                                                 "once per batch; staggered arrivals keep the in-flight CPIs at different phases (profiles/r03_pacing_sweep.txt)")
    ap.add_argument("--schedule", choices=("ordered", "free"), default="free",
                    help="ordered: all in-flight contexts enqueue on ONE pair of streams (isac_ctx_share_streams + ISAC_OPT_WIDE_ORDER) -- the wide kernels of "
                         "consecutive CPIs run back to back, never side by side, no pacing; free: two streams per context, the device interleaves the CPIs")
    ap.add_argument("--trace-only", action="store_true", help="profiling aid: nothing after the timed region (no isolated-kernel / blocking-CPI / stage / CPU legs), "
                                                              "so that a rocprofv3 trace holds priming + warm-up + the timed steps only")
    ap.add_argument("--workload", choices=("config2", "config5"), default="config2",
                    help="config2 (default): the sensing CPI BASELINE's metric is quoted on; config5: BASELINE configs[4] -- per frame and cell one sensing CPI + every "
                         "DL slot through every UE's CDL channel + every UE's CSI reports (--cells, default 21; --ues)")
    ap.add_argument("--ues", type=int, default=10, help="config5: UEs per cell")
    ap.add_argument("--n1-value", type=float, default=None, help="N > 1: the N = 1 value of the same per-GPU workload (slots/s) -> `efficiency_vs_n1` in the line")
    ap.add_argument("--n1-leg", action="store_true", help="N > 1: before the timed region rank 0 times the same per-GPU workload ALONE (the other ranks idle at a "
                                                          "barrier) and the line carries `n1_in_run` + `efficiency_vs_n1`: the whole scaling point in one command")
    ap.add_argument("--cold-probe", choices=("reserve", "noreserve"), default=None, help="internal: the fresh-process leg of the `cold` block (see cold_probe)")
    ap.add_argument("--cold-warm-ms", type=float, default=100.0, help="warm_ms handed to isac_ctx_reserve by the cold probe")
    ap.add_argument("--no-cold", action="store_true", help="skip the `cold` block (two fresh processes after the timed region)")
    args = ap.parse_args()
    if args.cold_probe:
        cold_probe(args, importlib.import_module(PKG))
        return
    # (round 1 selected the one-CU Jacobi eigensolver for pipelined runs; with this round's kernels the library default -- the tridiagonal
    # pipeline above 16 antennas -- is as fast or faster pipelined and 0.7 ms shorter in the drain tail: no override any more)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn_under_torchrun(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(args.gpus, 1) and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; the launcher's world size is used", file=sys.stderr)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        n_dev = torch.cuda.device_count()
        backend = os.environ.get("ISAC_DIST_BACKEND", "nccl")          # "gloo": test hook (several ranks on one GPU)
        if backend == "nccl":
            # one rank per GPU, never two ranks on one device: a launcher that starts more local ranks than there are GPUs is an error here,
            # not something to paper over by wrapping device indices (the scaling point would be measured on shared devices)
            local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
            if n_dev < local_world or local_rank >= n_dev:
                sys.exit(f"bench.py: rank {rank}: {local_world} local rank(s) but {n_dev} visible GPU(s) (LOCAL_RANK={local_rank}); "
                         "the RCCL run needs one GPU per rank (ISAC_DIST_BACKEND=gloo is the several-ranks-per-GPU test hook)")
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % max(n_dev, 1)
            dist.init_process_group(backend)
    pkg = importlib.import_module(PKG)
    if args.workload == "config5":
        if "--steps" not in sys.argv:
            args.steps = 3
        if "--warmup" not in sys.argv:
            args.warmup = 1
        run_config5(args, pkg, rank, world, local_rank, dist, torch)
        if dist is not None:
            dist.destroy_process_group()
        return
    lazy_native = args.fuse and args.noise_domain == "spectral" and 48 < args.ants <= 64 and args.targets <= 2
    args.lazy = args.echo == "lazy" or (args.echo == "auto" and lazy_native)
    pool = SlotPool(pkg, local_rank, args.inflight, ordered=args.schedule == "ordered")          # the GPU's execution slots, shared by all its cells
    if pool.ordered:
        args.pace_ms = 0.0                                   # submission order IS the device order: nothing to stagger
    pool.pace_s = 1e-3 * max(args.pace_ms, 0.0)
    d = importlib.import_module(PKG + "._dist")
    my_cells = d.shard_cells(args.cells, rank, world) if args.cells > 0 else [rank * args.cells_per_gpu + c for c in range(args.cells_per_gpu)]
    n_total_cells = args.cells if args.cells > 0 else args.cells_per_gpu * world
    n_buf = -(-args.inflight // max(len(my_cells), 1))        # CPIs of one cell that can be in flight at once
    cells = [Cell(pkg, local_rank, cid, args.ants, args.slots, args.targets, fuse=args.fuse, pool=pool, n_buf=n_buf, noise_domain=args.noise_domain, lazy=args.lazy)
             for cid in my_cells]

    def barrier():
        pool.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # Device priming (untimed, before the W warm-up steps): a GPU that has been idle during the host-side set-up runs its first
    # tens of milliseconds below its sustained clocks -- 20 timed steps behind 5 warm-up steps measured 35 % under the
    # steady-state rate, behind 100 warm-up steps they measure the steady state (profiles/r02_warmup_sensitivity.txt).  The
    # timed region below is still exactly K steps of the full hot path between two barriers.
    # (the host interpreter's collector is switched off from here to the end of the timed region: a pause inside 16 ms of timed steps would be the
    #  measurement, and a collection BETWEEN warm-up and the timed steps idles the GPU long enough for its clocks to drop: 18.2 k instead of 19.5-20.3 k)
    if os.environ.get("ISAC_BENCH_GC", "off") == "off":
        gc.collect()
        gc.disable()
    prime_steps, t_prime = 0, time.perf_counter()
    stamps, unpaced_ms = [], None                            # auto pacing: un-paced period over the last CPIs before 60 % of the priming phase
    n_win = max(3 * args.inflight, 16)                       # (the first tens of ms run at low clocks and hold the first-call allocations)
    while 1e3 * (time.perf_counter() - t_prime) < args.prime_ms:
        for cell in cells:
            pool.submit(cell)
        prime_steps += 1
        if args.pace_ms < 0 and unpaced_ms is None and args.inflight > 1:
            now = time.perf_counter()
            stamps.append((now, pool.k))
            if 1e3 * (now - t_prime) >= 0.6 * args.prime_ms:
                est = []                                     # the fastest of the last few windows of n_win CPIs (a hiccup must not become the pace)
                for j in range(len(stamps) - 1, max(len(stamps) - 1 - 4 * n_win, 0), -max(n_win // 2, 1)):
                    old = [st for st in stamps[:j] if stamps[j][1] - st[1] >= n_win]
                    if old:
                        est.append(1e3 * (stamps[j][0] - old[-1][0]) / (stamps[j][1] - old[-1][1]))
                if est:
                    unpaced_ms = min(est)
                    pool.pace_s = 0.93e-3 * unpaced_ms       # the rest of the priming phase, the warm-up and the timed steps run paced
    pool.drain()
    prime_ms = 1e3 * (time.perf_counter() - t_prime)
    for _ in range(args.warmup):
        for cell in cells:
            pool.submit(cell)
    pool.drain()
    # optional in-run N = 1 leg (multi-GPU only): rank 0 runs `steps` CPIs of ITS per-GPU workload with every other GPU idle
    n1_in_run = None
    if world > 1 and args.n1_leg:
        barrier()
        if rank == 0:
            t1 = time.perf_counter()
            for _ in range(args.steps):
                for cell in cells:
                    pool.submit(cell)
            pool.drain(); pool.sync()
            n1_in_run = args.steps * len(cells) * args.slots / (time.perf_counter() - t1)
        barrier()
    sink = [] if (rank == 0 and args.fuse) else None
    if sink is not None:
        for c_ in pool.ctxs:
            c_.check(c_.lib.isac_profile_enable(c_.handle, 1))
        for cell in cells:
            cell.profile_sink = sink
    barrier()
    pool.timeline = []
    pool.next_t, pool.spin_run = 0.0, 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for cell in cells:
            pool.submit(cell)
    pool.drain()
    pool.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    # one line per rank on stderr: which device it ran on, how many cells it held, its own timed region (the first multi-GPU run must be diagnosable)
    os.write(2, (f"bench.py: rank {rank}/{world}: device {local_rank} ({torch.cuda.get_device_name(local_rank) if torch.cuda.is_available() else 'no GPU'}), "
                 f"{len(cells)} cell(s) {my_cells}, timed region {1e3 * dt:.3f} ms for {args.steps} step(s), backend {dist.get_backend() if dist is not None else 'none'}\n").encode())   # (one write: lines of different ranks do not interleave)
    tl, pool.timeline = np.array(pool.timeline).reshape(-1, 2), None
    for cell in cells:
        cell.profile_sink = None
    dom_ms_timed = float(np.mean(sink)) if sink else None                       # fused kernel, launches of the timed region (other CPIs co-running)
    dom_ms_iso = cells[0].time_dominant_kernel_isolated() if (rank == 0 and args.fuse and not args.trace_only) else None
    cov_ms_iso = cells[0].time_covariance_kernel_isolated() if (rank == 0 and args.fuse and not args.trace_only) else None
    blocking_ms = None
    if rank == 0 and not args.trace_only:                    # latency of one blocking CPI (submit + collect, nothing else in flight)
        cells[0].step()
        tb = time.perf_counter()
        for _ in range(3):
            cells[0].step()
        blocking_ms = 1e3 * (time.perf_counter() - tb) / 3
    stages = stage_table(cells[0]) if (rank == 0 and not args.trace_only) else []
    # per-cell result record gather -- the only collective (KB-scale, RCCL over xGMI); max over ranks of the timed region
    recs = np.array([d.make_record(cid, cell.last, dt) for cid, cell in zip(my_cells, cells)]).reshape(-1, d.RECORD_LEN)
    on_gpu = dist is not None and dist.get_backend() == "nccl"
    allr = d.gather_records(recs, dist, torch.device("cuda", local_rank) if on_gpu else None)
    # per-rank timed-region durations (every record of a rank carries that rank's dt): the max defines `value`
    per_rank_ms = {}
    for r in allr:
        rk = (int(r[0]) % world) if args.cells > 0 else (int(r[0]) // max(args.cells_per_gpu, 1))
        per_rank_ms[rk] = round(1e3 * float(r[6]), 3)
    dt = float(np.nanmax(allr[:, 6])) if allr.size else dt
    n_cpi = args.steps * n_total_cells
    slots = n_cpi * args.slots
    if rank == 0:
        echo_b, rdm_b = cells[0].algorithmic_bytes()
        per_cpi_ms = 1e3 * dt / (args.steps * max(len(my_cells), 1))
        res = {
            "metric": "sensing slots/sec (CDL echo->2D-FFT->2D-CFAR)", "value": round(slots / dt, 2), "unit": "sensing slots/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True, "scaling": "strong" if args.cells > 0 else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "per_rank_ms": [per_rank_ms.get(r) for r in range(world)],
            "hw_queues": {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "hip_streams": 2 * args.inflight,
                          "note": "set by bench.py before the HIP runtime starts (the library loader does not touch the environment): one hardware queue per "
                                  "HIP stream, two streams per in-flight CPI; INTEGRATION.md section 4"},
            "host_timeline": None if not tl.size else {
                "collect_wait_ms": {"p50": round(1e3 * float(np.median(tl[:, 0])), 3), "p99": round(1e3 * float(np.percentile(tl[:, 0], 99)), 3), "max": round(1e3 * float(tl[:, 0].max()), 3)},
                "enqueue_ms": {"p50": round(1e3 * float(np.median(tl[:, 1])), 3), "p99": round(1e3 * float(np.percentile(tl[:, 1], 99)), 3), "max": round(1e3 * float(tl[:, 1].max()), 3)},
                "enqueue_over_1ms": int((tl[:, 1] > 1e-3).sum()), "enqueue_total_ms": round(1e3 * float(tl[:, 1].sum()), 2), "collect_total_ms": round(1e3 * float(tl[:, 0].sum()), 2),
                "note": "host wall time per submitted CPI in the timed region: waiting for + post-processing the oldest CPI of the slot (collect), then the launches of the new one (enqueue)"},
            "pacing": {"mode": "off" if pool.pace_s == 0.0 else ("auto" if args.pace_ms < 0 else "fixed"), "pace_ms": round(1e3 * pool.pace_s, 4),
                       "unpaced_ms_per_cpi_during_priming": None if unpaced_ms is None else round(unpaced_ms, 4),
                       "note": "host-side minimum spacing of consecutive CPI submissions (staggered arrivals; 0.93 x the un-paced period measured while priming)"},
            "pipeline": {"cpis_in_flight": args.inflight, "schedule": args.schedule, "blocking_cpi_ms": None if blocking_ms is None else round(blocking_ms, 3),
                         "note": "the timed region starts with an empty device and ends fully drained: its K steps include one pipeline fill and "
                                 "one drain (about one blocking CPI latency in total); steady-state rate = the same command with --steps 100"},
            "priming": {"untimed_steps_before_warmup": prime_steps, "ms": round(prime_ms, 1),
                        "why": "idle GPU ramps to sustained clocks; timed region = exactly `steps` CPIs between barriers"},
            "config": {"workload": f"{(str(args.cells) + ' cells round-robin over the ranks') if args.cells > 0 else (str(args.cells_per_gpu) + ' cell(s)/GPU')}, {args.ants}-antenna ULA echo -> 2D-FFT -> 2D-CFAR -> MUSIC, "
                                   f"100 MHz / 273 PRB, K=3276 L={14 * args.slots} T={cells[0].T} nIFFT=4096 nFFT=256, "
                                   f"{args.targets} target(s), Philox AWGN drawn on the {'demodulated grid (Philox4x32-10 + SINGLE-precision hardware Box-Muller: the field is defined to float32 accuracy, echo_dev.hpp box_muller32_hw; all signal arithmetic fp64)' if args.noise_domain == 'spectral' else 'time samples (fp64 Box-Muller)'}, "
                                   f"{'fused synthesis + range kernel' if args.fuse else 'separate echo / range kernels'}, "
                                   + ("echo grid LAZY (kept inside the context as a descriptor: never written to HBM, the covariance kernel re-forms its operands from the same D, a, seed -- identical CFAR lists and estimates, isac.h 'LAZY echo grid'), " if args.lazy else "echo grid materialised (written by monoStaticSensing, read back by the covariance), ")
                                   + f"{args.inflight} CPIs in flight",
                       "parallelism": f"cells sharded over {world} GPU(s)"},
            "roofline": roofline_entry(cells[0], args, dom_ms_timed, dom_ms_iso, len(sink or []), stages, echo_b + rdm_b, per_cpi_ms, cov_ms_iso),
        }
        res["cells"] = [d.record_json(r) for r in allr[:64]]                      # every cell's whole estResults, gathered from every rank
        if world > 1:
            n1 = n1_in_run if n1_in_run is not None else args.n1_value
            per_gpu = "the same per-GPU workload" if args.cells == 0 else "rank 0's share of the cells"
            res["scaling_point"] = {"n1_value": None if n1 is None else round(n1, 2), "n1_source": "in-run leg (rank 0 alone, other GPUs idle)" if n1_in_run is not None
                                    else ("--n1-value" if args.n1_value is not None else None),
                                    "efficiency_vs_n1": None if n1 is None else round(res["value"] / (world * n1), 4),
                                    "note": f"efficiency = value / (N x N=1 value of {per_gpu}); no multi-GPU scaling curve has been measured on hardware yet "
                                            "(DESIGN.md section 7) -- the driver computes its own from the per-N lines"}
        if world == 1 and not args.trace_only and not args.no_cold and args.workload == "config2":
            res["cold"] = cold_block(args)
        if not args.no_cpu_baseline and world == 1 and not args.trace_only:
            res["cpu_baseline"] = cpu_baseline(cells[0])
            res["gpu_vs_cpu_baseline"] = round(res["value"] / res["cpu_baseline"]["value"], 1)
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
