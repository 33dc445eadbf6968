"""CDL apply at config 5's OWN shape against the oracle (VERDICT r5 'what's missing' #3, 'weak' #2): 122.88 MHz sampling, T = 61 440 + MaxChannelDelay = 61 909 samples,
64 -> 2 (downlink, cdl.m:57-64, stepped at uePhy.m:729-731) and 2 -> 64 (uplink, cdl.m:78-85, stepped at gNBPhy.m:833-864), CDL-A and CDL-D, >= 20 (UE, slot) jobs per
library call, downlink jobs of a slot on ONE shared waveform, jobs that cross a path-gain refresh -- i.e. the launches `bench.py --workload config5` times
(cdl_fused_kernel<NCT,NSLOT> and cdl_fused_ul_kernel<NCT> in their persistent multi-job form), every output element <= 1e-10 of the TR 38.901 restatement
(oracle/cdl.py).  The second test drives bench.py's own CommCell for one frame and re-computes one downlink job, one uplink job and the precoded PDSCH input with the oracle."""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import load_pkg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RTOL = 1e-10
FS, SLOT_T = 122.88e6, 61440
UE, GNB64 = (1, 1, 2, 1, 1), (4, 8, 2, 1, 1)


@pytest.fixture(scope="module")
def pkg():
    return load_pkg()


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(np.asarray(b)).max())


@pytest.mark.parametrize("profile,uplink,n_ue", [("CDL-A", False, 10), ("CDL-A", True, 10), ("CDL-D", False, 4), ("CDL-D", True, 4)])
def test_cdl_batch_at_config5_shape_matches_oracle(pkg, profile, uplink, n_ue):
    """One isac_cdl_apply_batch_dev call of n_ue UEs x 2 consecutive slots (20 jobs for CDL-A, 8 for CDL-D) at T = 61 909: every channel appears twice (its time advances from
    job to job); half of the UEs share the reference's seed 73 (one device evaluation of their path gains, cdl.m:57-64), the others have their own; UE 1 starts half a
    waveform in front of a path-gain refresh (its first job holds two gain blocks, the second starts right behind the refresh), UE 2 crosses one in its second slot;
    downlink: the UEs of a slot share ONE waveform (uePhy.m:729-731 inside the per-UE loop), uplink: one waveform per UE (gNBPhy.m:833-864)."""
    import oracle.cdl as OC
    CM = pkg.communication.channelModels
    ctx = pkg.default_context()
    tx, rx = (UE, GNB64) if uplink else (GNB64, UE)
    nt, nr = int(np.prod(tx)), int(np.prod(rx))
    seeds = [73 if u % 2 == 0 else 70 + u for u in range(n_ue)]
    chans = [CM.CDLChannel(profile, 300e-9, 3.5e9, tx, rx, FS, Seed=sd) for sd in seeds]
    T = SLOT_T + max(ch.info().MaxChannelDelay for ch in chans)
    if profile == "CDL-D":
        assert T == 61909                                                  # the waveform length of bench.py's config-5 frame (CDL-D has the longest delay)
    T = 61909
    refresh = 1.0 / (2 * 64 * 5.0)                                         # SampleDensity 64, MaximumDopplerShift 5 Hz: a new gain block every 1.5625 ms
    t_start = [0.0] * n_ue
    t_start[1] = refresh - (T // 2) / FS                                   # refresh in the middle of the first slot
    t_start[2] = 3 * refresh - (T + 20000) / FS                            # refresh inside the second slot
    t_start[3] = 0.25
    for ch, t in zip(chans, t_start):
        ch.time = t
    rng = np.random.default_rng(1000 * nt + nr + n_ue)
    n_wave = n_ue if uplink else 2
    xs = [np.asfortranarray(rng.standard_normal((T, nt)) + 1j * rng.standard_normal((T, nt))) for _ in range(n_wave)]
    d_xs = [ctx.to_device(x) for x in xs]
    jobs = [(u, s) for s in range(2) for u in range(n_ue)]                 # slot-major, as CommCell.enqueue_frame lists them
    which = [(u if uplink else s) for u, s in jobs]
    outs = CM.applyCDLBatch([chans[u] for u, _ in jobs], [d_xs[w] for w in which], ctx=ctx)
    assert len(outs) == 2 * n_ue >= 8
    n_blocks = 0
    for (u, s), w, o in zip(jobs, which, outs):
        cfg = OC.cdl_config(profile, 3.5e9, tx, rx, FS, seed=seeds[u])
        t0 = t_start[u] + s * T / FS
        blk, _ = OC.snapshot_index(cfg, t0 + np.array([0, T - 1]) / FS)
        n_blocks += int(blk[1] - blk[0]) + 1
        want = OC.apply_cdl(cfg, xs[w], t0)
        got = o.numpy()
        assert got.shape == (T, nr)
        assert rel(got, want) < RTOL, (profile, uplink, u, s)
    assert n_blocks >= 2 * n_ue + 2                                        # at least two jobs really held two gain blocks
    for ch, t in zip(chans, t_start):
        assert ch.time == pytest.approx(t + 2 * T / FS)


@pytest.mark.parametrize("profile,tx,fs,t_len,n_ue", [("CDL-A", (1, 4, 2, 1, 1), 122.88e6, 9000, 3), ("CDL-D", (1, 8, 2, 1, 1), 30.72e6, 20011, 2), ("CDL-A", (2, 8, 2, 1, 1), 61.44e6, 12345, 9),
                                                       ("CDL-D", (4, 8, 2, 1, 1), 122.88e6, 8192, 1)])
def test_cdl_overlap_save_shapes_match_oracle(pkg, profile, tx, fs, t_len, n_ue):
    """The frequency-domain downlink apply (cdl_os.hip: 4096-point overlap-save, the waveform's transforms shared by its UEs) at 8 / 16 / 32 / 64 transmit elements, ragged
    lengths (the last window mostly zeros), nine UEs on one waveform (two mix chunks), every UE next to a path-gain refresh (two gain blocks: two (job, block) pairs whose
    windows overlap at the boundary) -- every output sample <= 1e-10 of the oracle."""
    import oracle.cdl as OC
    CM = pkg.communication.channelModels
    ctx = pkg.default_context()
    nt = int(np.prod(tx))
    refresh = 1.0 / 640
    rng = np.random.default_rng(nt + t_len)
    x = np.asfortranarray(rng.standard_normal((t_len, nt)) + 1j * rng.standard_normal((t_len, nt)))
    d_x = ctx.to_device(x)
    chans, t_start = [], []
    for u in range(n_ue):
        ch = CM.CDLChannel(profile, 300e-9, 3.5e9, tx, UE, fs, Seed=73 + (u % 3))
        t0 = (2 + u) * refresh - (1 + (u * t_len) // max(n_ue, 1) % t_len) / fs if u % 2 == 0 else 0.01 * u
        ch.time = t0
        chans.append(ch); t_start.append(t0)
    outs = CM.applyCDLBatch(chans, [d_x] * n_ue, ctx=ctx)
    for u, (ch, o) in enumerate(zip(chans, outs)):
        cfg = OC.cdl_config(profile, 3.5e9, tx, UE, fs, seed=73 + (u % 3))
        want = OC.apply_cdl(cfg, x, t_start[u])
        assert rel(o.numpy(), want) < RTOL, (profile, tx, u)


@pytest.mark.parametrize("profile,rx,fs,t_len,n_ue,tx", [("CDL-A", (1, 4, 2, 1, 1), 122.88e6, 9000, 3, UE), ("CDL-D", (1, 8, 2, 1, 1), 30.72e6, 20011, 2, UE), ("CDL-A", (2, 8, 2, 1, 1), 61.44e6, 12345, 5, UE),
                                                          ("CDL-D", (4, 8, 2, 1, 1), 122.88e6, 8192, 1, UE), ("CDL-A", (1, 3, 1, 1, 1), 122.88e6, 7777, 2, (1, 1, 1, 1, 1))])
def test_cdl_overlap_save_uplink_shapes_match_oracle(pkg, profile, rx, fs, t_len, n_ue, tx):
    """The frequency-domain UPLINK apply (cdl_os_ul_kernel: one workgroup per (job, gain block, receive element), the combined impulse response transformed once, Y(f) formed
    in the inverse transform's registers) at 8 / 16 / 32 / 64 receive elements and a single-antenna UE into three, ragged lengths, every UE with its own waveform, UEs next to
    a path-gain refresh (two gain blocks whose windows overlap at the boundary) -- every output sample <= 1e-10 of the oracle."""
    import oracle.cdl as OC
    CM = pkg.communication.channelModels
    ctx = pkg.default_context()
    nt, nr = int(np.prod(tx)), int(np.prod(rx))
    refresh = 1.0 / 640
    rng = np.random.default_rng(nr + t_len)
    xs = [np.asfortranarray(rng.standard_normal((t_len, nt)) + 1j * rng.standard_normal((t_len, nt))) for _ in range(n_ue)]
    d_xs = [ctx.to_device(x) for x in xs]
    chans, t_start = [], []
    for u in range(n_ue):
        ch = CM.CDLChannel(profile, 300e-9, 3.5e9, tx, rx, fs, Seed=73 + (u % 3))
        t0 = (2 + u) * refresh - (1 + (u * t_len) // max(n_ue, 1) % t_len) / fs if u % 2 == 0 else 0.01 * u
        ch.time = t0
        chans.append(ch); t_start.append(t0)
    outs = CM.applyCDLBatch(chans, d_xs, ctx=ctx)
    for u, (ch, o) in enumerate(zip(chans, outs)):
        cfg = OC.cdl_config(profile, 3.5e9, tx, rx, fs, seed=73 + (u % 3))
        want = OC.apply_cdl(cfg, xs[u], t_start[u])
        assert o.numpy().shape == (t_len, nr)
        assert rel(o.numpy(), want) < RTOL, (profile, rx, u)


def test_shared_forward_spectra_between_delay_profile_groups(pkg):
    """ISAC_OPT_CDL_SHARE_SPECTRA: the CDL-D UEs and the CDL-A UEs of a cell receive the same two slot waveforms; their two batched calls on one context share the forward
    transforms (common window step, Mpad = 512) -- every output <= 1e-10 of the oracle and <= 1e-12 of the un-shared calls; a call on OTHER waveforms in between transforms anew,
    and so does the same pointer list after the option is switched off."""
    import oracle.cdl as OC
    CM = pkg.communication.channelModels
    T = 61909
    rng = np.random.default_rng(99)
    xs = [np.asfortranarray(rng.standard_normal((T, 64)) + 1j * rng.standard_normal((T, 64))) for _ in range(3)]
    outs = {}
    for share in (True, False):
        ctx = pkg.Context(0)
        ctx.set_cdl_share_spectra(share)
        d_xs = [ctx.to_device(x) for x in xs]
        res = []
        for profile, n_ue in (("CDL-D", 2), ("CDL-A", 3)):
            chans = [CM.CDLChannel(profile, 300e-9, 3.5e9, GNB64, UE, FS, Seed=73 + u) for u in range(n_ue)]
            for ch in chans:
                ch.time = 0.004
            o = CM.applyCDLBatch([chans[u] for s_ in range(2) for u in range(n_ue)], [d_xs[s_] for s_ in range(2) for u in range(n_ue)], ctx=ctx)
            res.append([a.numpy() for a in o])
        # other waveforms: (x2, x0) -- a different pointer list, transformed anew; then the first list again
        ch = CM.CDLChannel("CDL-A", 300e-9, 3.5e9, GNB64, UE, FS, Seed=80)
        ch.time = 0.004
        o2 = CM.applyCDLBatch([ch, ch], [d_xs[2], d_xs[0]], ctx=ctx)
        res.append([a.numpy() for a in o2])
        outs[share] = res
    for a_l, b_l in zip(outs[True], outs[False]):
        for a, b in zip(a_l, b_l):
            assert rel(a, b) < 1e-12
    for gi, (profile, n_ue) in enumerate((("CDL-D", 2), ("CDL-A", 3))):
        for j in range(2 * n_ue):
            s_, u = divmod(j, n_ue)
            cfg = OC.cdl_config(profile, 3.5e9, GNB64, UE, FS, seed=73 + u)
            want = OC.apply_cdl(cfg, xs[s_], 0.004 + s_ * T / FS)
            assert rel(outs[True][gi][j], want) < RTOL, (profile, s_, u)
    cfg = OC.cdl_config("CDL-A", 3.5e9, GNB64, UE, FS, seed=80)
    assert rel(outs[True][2][0], OC.apply_cdl(cfg, xs[2], 0.004)) < RTOL and rel(outs[True][2][1], OC.apply_cdl(cfg, xs[0], 0.004 + T / FS)) < RTOL


def test_bench_config5_frame_against_oracle(pkg):
    """bench.py's CommCell (the object `--workload config5` times) stepped for one frame: (a) the precoded PDSCH input of a downlink slot = oracle prgPrecode of the
    same layers and precoders + the oracle's CP-OFDM modulator; (b) one downlink job of the frame's last call (UE, slot 15) and (c) one uplink job (UE, third 'U' slot:
    cdl_fused_ul_kernel, 2 -> 64, T = 61 909) <= 1e-10 of oracle/cdl.py at the channel times the frame loop gave them; an NLoS (CDL-A) and a LoS (CDL-D) UE each."""
    sys.path.insert(0, ROOT)
    import bench
    import oracle as O
    import oracle.cdl as OC
    import oracle.precode as OPR
    ctx_a, ctx_b, ctx_csi = pkg.Context(0), pkg.Context(0), pkg.Context(0)
    cell_id, n_ants, n_ues = 4, 64, 10
    cc = bench.CommCell(pkg, [ctx_a, ctx_b], ctx_csi, cell_id, n_ants, n_ues)
    assert cc.T == 61909 and len(cc.groups) == 2 and cc.WITH_UL
    K = 3276
    # ---- (a) precoded input of slot 9
    s_chk = 9
    layers = ctx_a.empty((K, 14, cc.LAYERS))
    ctx_a.check(ctx_a.lib.isac_synth_qpsk_grid_dev(ctx_a.handle, C.c_void_p(layers.ptr), K, 14, cc.LAYERS, C.c_uint64(cc.layer_seeds[s_chk]), 0))
    lay = layers.numpy()
    F = cc.precoders[s_chk]
    lin = np.arange(K * 14 * cc.LAYERS, dtype=np.int64).reshape(K * 14, cc.LAYERS, order="F") + 1       # every RE of every layer, 1-based linear indices
    sym = lay.reshape(K * 14, cc.LAYERS, order="F")
    antsym, antind = OPR.prg_precode((K, 14), 0, sym, lin, F)
    grid = np.zeros(K * 14 * n_ants, dtype=np.complex128)
    grid[antind.reshape(-1) - 1] = antsym.reshape(-1)
    want_wave = np.zeros((cc.T, n_ants), dtype=np.complex128)
    want_wave[:SLOT_T] = O.ofdm_modulate(grid.reshape((K, 14, n_ants), order="F"), 4096, 30)
    got_wave = cc.waves[s_chk].numpy()
    assert rel(got_wave, want_wave) < RTOL
    # ---- one frame as the bench issues it
    cc.enqueue_frame()
    for c_ in (ctx_a, ctx_b):
        c_.sync()
    nt_shape = (n_ants // 16, 8, 2, 1, 1)
    for gi, g in enumerate(cc.groups):
        profile = cc.chans[g[0]].DelayProfile
        i = len(g) // 2
        u = g[i]
        # (b) downlink: the last call of the frame covered slots 8..15; its outputs are rx[(s - 8) * len(g) + i]
        s_dl = cc.DL_SLOTS - 1
        s0 = (s_dl // cc.SLOTS_PER_CALL) * cc.SLOTS_PER_CALL
        got = cc.rx[gi][(s_dl - s0) * len(g) + i].numpy()
        cfg = OC.cdl_config(profile, 3.5e9, nt_shape, UE, FS, seed=73)
        want = OC.apply_cdl(cfg, cc.waves[s_dl].numpy(), s_dl * cc.T / FS)
        assert got.shape == want.shape == (cc.T, 2) and rel(got, want) < RTOL, (profile, "DL")
        # (c) uplink: third 'U' slot of the frame, this UE's own packet
        j = 2
        got = cc.ul_rx[gi][j * len(g) + i].numpy()
        cfg = OC.cdl_config(profile, 3.5e9, UE, nt_shape, FS, seed=73)
        want = OC.apply_cdl(cfg, cc.ul_waves[u].numpy(), j * cc.T / FS)
        assert got.shape == want.shape == (cc.T, n_ants) and rel(got, want) < RTOL, (profile, "UL")
    for c_ in (ctx_a, ctx_b, ctx_csi):
        c_.close()
