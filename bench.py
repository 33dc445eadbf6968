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
"""
from __future__ import annotations

import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np

# 4 CPIs in flight x 2 HIP streams each: give every stream its own hardware queue (the ROCm default of 4 makes
# streams share queues, and a 1.4 ms single-CU eig kernel at the head of a shared queue stalls the wide kernels
# behind it).  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "5g_based_system_level_integrated_sensing_and_communication_simulator_amd"

RANGE_KERNEL_HBM_BYTES_A64 = int((2 * 735413 + 84227) * 1024)   # 1.592e9 B vs 1.503e9 B algorithmic: no wasted re-reads
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured-achievable)
FP64_MFMA_PEAK_TFLOPS = 78.6     # MI355X fp64 matrix (= vector) dense peak


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
    CPIs are in flight on the GPU however many cells there are (more than 4 contexts = 8 streams oversubscribe the
    hardware queues and lose 10-25 %).  Results are collected in submission order."""

    def __init__(self, pkg, device, n):
        self.ctxs = [pkg.Context(device) for _ in range(max(1, n))]
        self.owner = [None] * len(self.ctxs)
        self.k = 0

    def collect(self, slot):
        cell, self.owner[slot] = self.owner[slot], None
        return cell.finish(self.ctxs[slot]) if cell is not None else None

    def submit(self, cell):
        slot = self.k % len(self.ctxs)
        self.collect(slot)
        cell.enqueue(self.ctxs[slot])
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

    def __init__(self, pkg, device, cell_id, n_ants, n_slots, n_targets, inflight=1, fuse=False, pool=None, n_buf=None):
        L = pkg._lib
        self.fuse = fuse
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
        ctx.sync()

    def finish(self, ctx):
        """Collect the CPI this cell enqueued on `ctx`."""
        est = None
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
                                                  seed=self.seed + self.n_sub, nfft=4096, out=self.echo[b], ctx=c,
                                                  fuse_fft2d=(self.rp, self.cfar, tx_grid) if self.fuse else None)
        self.pkg.sensing.estimation.fft2D_submit(self.rp, self.cfar, echo, tx_grid, ctx=c)
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

    def time_range_kernel(self, reps=10):
        """Average duration of the dominant HBM-bound kernel (range stage of fft2D) measured with HIP events on
        the stream it is launched on; algorithmic bytes per launch = rxGrid + txGrid = 2 K L A 16 B."""
        from importlib import import_module
        m = import_module(self.pkg.__name__ + ".sensing.estimation.fft2D")
        mm = import_module(self.pkg.__name__ + ".sensing._marshal")
        c = self.ctx
        r0, r1, c0, c1 = m._cut_rectangle(self.cfar.CUTIdx)
        det = self.cfar.cfarDetector2D
        cf = self.L.CfarConfig(det.ProbabilityFalseAlarm, (C.c_int32 * 2)(*det.GuardBandSize), (C.c_int32 * 2)(*det.TrainingBandSize), r0, r1, c0, c1)
        ep = mm.est_block(self.rp)
        def launch():
            c.check(c.lib.isac_fft2d_range_stage_dev(c.handle, C.byref(ep), C.byref(cf), C.c_void_p(self.echo[0].ptr), C.c_void_p(self.tx_grid.ptr),
                                                     self.K, self.Lsym, self.A))
        launch(); c.sync()
        c.timer_start()
        for _ in range(reps):
            launch()
        return c.timer_stop_ms() / reps


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

    def mono(noise):
        kw = dict(seed=cell.seed if noise else None, nfft=4096, out=cell.echo[0], ctx=c)
        return lambda: cell.pkg.sensing.monoStaticSensing(cell.tx_wave, (K, L, A), cell.carrier, cell.rp, cell.los, **kw)

    ra = c.empty((A, A))
    cov = lambda: c.check(c.lib.isac_covariance_dev(c.handle, C.c_void_p(cell.echo[0].ptr), C.c_int64(K * L), C.c_int32(A), C.c_void_p(ra.ptr)))
    out = []
    echo_b = T * A * b + K * L * A * b
    for name, fn, note in (("monoStaticSensing = beamsum + coef + demod, Philox AWGN", mono(True),
                            "fp64-VALU-bound: Philox4x32-10 + Box-Muller for 4096*L*A complex samples, then the OFDM FFT"),
                           ("monoStaticSensing, noise off", mono(False), "HBM: read txWaveform once, write echoGrid once")):
        ms = timed(fn)
        out.append({"stage": name, "ms": round(ms, 4), "bound": "hbm", "algorithmic_bytes": echo_b,
                    "achieved_GBps": round(echo_b / 1e9 / (ms / 1e3), 1), "frac": round(echo_b / 1e9 / (ms / 1e3) / HBM_PEAK_GBS, 4), "note": note})
    ms = timed(cov)
    fl = 8.0 * A * A * K * L
    out.append({"stage": "covariance Ra = X X^H / N (fp64 MFMA, Hermitian half issued)", "ms": round(ms, 4), "bound": "mfma",
                "algorithmic_flops": fl, "achieved_TFLOPs": round(fl / 1e12 / (ms / 1e3), 2), "peak_TFLOPs": FP64_MFMA_PEAK_TFLOPS,
                "frac": round(fl / 1e12 / (ms / 1e3) / FP64_MFMA_PEAK_TFLOPS, 4),
                "note": "nominal 8 A^2 K L flop; the kernel issues the upper-triangular 10 of 16 tiles"})
    return out


def cpu_baseline(n_ants, budget_s=25.0):
    """The NumPy/SciPy oracle ("port": the MATLAB reference cannot run here; FFTs = multi-threaded pocketfft, the RDM in
    its shift-free form) timed on the host cores on a
    bounded sample of the same workload: the full 273-PRB / 224-symbol CPI with a reduced antenna count,
    scaled linearly to `n_ants` (every stage of the chain is linear in the antenna count except the
    A x A covariance/eig, which is negligible on the CPU at these sizes)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    from conftest import make_scene
    a_s = 16
    sc = make_scene(n_ants=a_s, n_slots=16, nrb=273, targets=((100.0, 20.0, 1.5),), velocity=(7.0,), seed=3)
    cf = O.cfar2d_config(sc.rp)
    t0 = time.perf_counter()
    reps = 0
    while True:
        echo = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, sc.los, sc.noise)
        try:
            O.fft2d(sc.rp, cf, echo, sc.tx_grid, rdm_fn=O.rdm_explicit)   # same bits as the literal form, no shift copies
        except ValueError:
            pass
        reps += 1
        if time.perf_counter() - t0 > budget_s or reps >= 8:
            break
    dt = (time.perf_counter() - t0) / reps
    cpi_s = dt * (n_ants / a_s)
    return {"value": round(16.0 / cpi_s, 3), "unit": "sensing slots/sec", "cores": os.cpu_count(), "kind": "port",
            "sample": f"NumPy/SciPy oracle, full 273-PRB x 224-symbol CPI at {a_s} antennas x{reps} reps, scaled x{n_ants // a_s} "
                      f"to {n_ants} antennas (noise pre-drawn, not timed); scipy.fft workers = all cores"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--ants", type=int, default=64)
    ap.add_argument("--slots", type=int, default=16)
    ap.add_argument("--targets", type=int, default=1)
    ap.add_argument("--cells-per-gpu", type=int, default=1)
    ap.add_argument("--inflight", type=int, default=4, help="CPIs in flight per cell (contexts)")
    ap.add_argument("--fuse", action="store_true", help="fuse the fft2D range stage into monoStaticSensing (measured slower: off by default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    # eigensolver trade-off (music.hip isac_eigh_dev): with several CPIs in flight the one-workgroup Jacobi solver (1.4 ms at
    # A = 64, hidden behind the other CPIs, one CU) gives a 3-5 % higher rate than the latency-optimised tridiagonal pipeline
    # (0.9 ms, up to five CUs); a blocking caller (--inflight 1, the reference's call order) gets the pipeline
    if args.inflight > 1:
        os.environ.setdefault("ISAC_EIG_JACOBI_MAX", "64")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        n_dev = torch.cuda.device_count()
        backend = os.environ.get("ISAC_DIST_BACKEND", "nccl")          # "gloo": test hook (several ranks on one GPU)
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % max(n_dev, 1)
            dist.init_process_group(backend)
    pkg = importlib.import_module(PKG)
    pool = SlotPool(pkg, local_rank, args.inflight)          # the GPU's execution slots, shared by all its cells
    n_buf = -(-args.inflight // args.cells_per_gpu)           # CPIs of one cell that can be in flight at once
    cells = [Cell(pkg, local_rank, rank * args.cells_per_gpu + c, args.ants, args.slots, args.targets, fuse=args.fuse, pool=pool, n_buf=n_buf)
             for c in range(args.cells_per_gpu)]

    def barrier():
        pool.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        for cell in cells:
            pool.submit(cell)
    pool.drain()
    barrier()
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
    range_ms = cells[0].time_range_kernel() if rank == 0 else 0.0
    stages = stage_table(cells[0]) if rank == 0 else []
    # per-cell result record gather -- the only collective (KB-scale, RCCL over xGMI); max over ranks of the timed region
    d = importlib.import_module(PKG + "._dist")
    recs = np.array([d.make_record(rank * args.cells_per_gpu + i, cell.last, dt) for i, cell in enumerate(cells)])
    on_gpu = dist is not None and dist.get_backend() == "nccl"
    allr = d.gather_records(recs, dist, torch.device("cuda", local_rank) if on_gpu else None)
    dt = float(np.nanmax(allr[:, 6])) if allr.size else dt
    n_cpi = args.steps * args.cells_per_gpu * world
    slots = n_cpi * args.slots
    if rank == 0:
        echo_b, rdm_b = cells[0].algorithmic_bytes()
        per_cpi_ms = 1e3 * dt / (args.steps * args.cells_per_gpu)
        res = {
            "metric": "sensing slots/sec (CDL echo->2D-FFT->2D-CFAR)", "value": round(slots / dt, 2), "unit": "sensing slots/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.cells_per_gpu} cell(s)/GPU, {args.ants}-antenna ULA echo -> 2D-FFT -> 2D-CFAR -> MUSIC, "
                                   f"100 MHz / 273 PRB, K=3276 L={14 * args.slots} T={cells[0].T} nIFFT=4096 nFFT=256, "
                                   f"{args.targets} target(s), Philox AWGN, {args.inflight} CPIs in flight",
                       "parallelism": f"cells sharded over {world} GPU(s)"},
            "roofline": {"bound": "hbm", "kernel": "range_kernel<Fft4096> (fft2D range stage: rx.*conj(tx), Kaiser, 4096-pt IFFT)",
                         "achieved": round(rdm_b / 1e9 / (range_ms / 1e3), 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(rdm_b / 1e9 / (range_ms / 1e3) / HBM_PEAK_GBS, 4),
                         # HBM bytes per launch from rocprofv3 PMC passes (profiles/r01_pmc_*.csv): 2 x FETCH_SIZE (gfx950 counts
                         # wide coalesced reads at half, MI355X_MICROARCH.md) + WRITE_SIZE, KB -> B; measured at the A=64 shape only
                         "traffic": RANGE_KERNEL_HBM_BYTES_A64 if (args.ants == 64 and args.slots == 16) else None,
                         "traffic_source": "profiles/r01_pmc_fetch_size.csv + r01_pmc_write_size.csv (separate --pmc passes)",
                         "avg_launch_ms": round(range_ms, 4), "algorithmic_bytes_per_launch": rdm_b,
                         "other_stages": stages,
                         "whole_cpi": {"algorithmic_bytes": echo_b + rdm_b, "ms": round(per_cpi_ms, 4),
                                       "achieved_GBps": round((echo_b + rdm_b) / 1e9 / (per_cpi_ms / 1e3), 1),
                                       "frac": round((echo_b + rdm_b) / 1e9 / (per_cpi_ms / 1e3) / HBM_PEAK_GBS, 4)}},
        }
        res["cells"] = [{"cell": int(r[0]), "nRng": None if np.isnan(r[1]) else int(r[1]), "rngEst0": None if np.isnan(r[2]) else round(float(r[2]), 6),
                         "velEst0": None if np.isnan(r[3]) else round(float(r[3]), 6), "aziEst0": None if np.isnan(r[4]) else float(r[4])}
                        for r in allr[:8]]
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(args.ants)
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
