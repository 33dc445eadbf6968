"""The MEX gateway itself, executed: mex/isac_mex.cpp is linked against an in-process implementation of the mx / mex API subset it uses
(tests/mex_runtime/) and libisac_hip.so, and tests/mex_host.cpp calls mexFunction() with the argument lists MATLAB would pass -- the
radarParams struct, carrierInfo, cfar2D's struct with its phased.CFARDetector2D value object, char vectors, interleaved-complex arrays
and uint64 device handles.  Checked against the committed golden fixture (echo grid <= 1e-10, estimates exact), the oracle (DoA
entries, basicRadarChannel, the stand-alone CFAR detector) and the error identifiers cellSimulation.m:196-202 relies on."""
from __future__ import annotations

import os
import struct
import subprocess

import numpy as np
import pytest

import oracle as O
from conftest import ROOT, make_scene

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "tests", "_build", "mex_host")


def _exe():
    if not os.path.exists(EXE):
        import __graft_entry__ as g
        g.build_mex_host()
    return EXE


def _est(buf, off):
    (n_echo,) = struct.unpack_from("<Q", buf, off); off += 8
    echo = np.frombuffer(buf, dtype=np.complex128, count=n_echo, offset=off); off += 16 * n_echo
    n = struct.unpack_from("<3i", buf, off); off += 12
    out = []
    for k in n:
        out.append(np.frombuffer(buf, dtype=np.float64, count=k, offset=off)); off += 8 * k
    return echo, out, off


def _vec(buf, off):
    (n,) = struct.unpack_from("<i", buf, off); off += 4
    return np.frombuffer(buf, dtype=np.float64, count=n, offset=off), off + 8 * n


def test_mex_gateway_runs_the_reference_call_sequences(tmp_path):
    g = np.load(os.path.join(ROOT, "tests", "golden", "chain_small.npz"))
    sc = make_scene(n_ants=int(g["n_ants"]), n_slots=int(g["n_slots"]), nrb=int(g["nrb"]), targets=tuple(map(tuple, g["targets"])),
                    velocity=tuple(g["velocity"]), num_slots_param=int(g["num_slots_param"]), seed=int(g["seed"]))
    cf = O.cfar2d_config(sc.rp)
    hdr = struct.pack("<18i q 8d", sc.K, sc.L, sc.A, int(sc.rp.nTargets), sc.wave.Nfft, 30, int(sc.rp.nIFFT), int(sc.rp.nFFT), 2, 2, 1, 1,
                      int(cf.CUTIdx[0].min()), int(cf.CUTIdx[0].max()), int(cf.CUTIdx[1].min()), int(cf.CUTIdx[1].max()), 1, 0, sc.T,
                      float(sc.rp.fc), float(sc.rp.fs), float(sc.rp.N0), float(sc.rp.rRes), float(sc.rp.vRes), float(cf.Pfa),
                      float(sc.rp.azimuthScanScale), float(sc.rp.azimuthScanGranularity))
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(hdr)
        for a in (np.asarray(sc.rp.range, np.float64), np.asarray(sc.rp.velocity, np.float64), np.asarray(sc.rp.largeScaleFading, np.float64)):
            f.write(np.ascontiguousarray(a).tobytes())
        f.write(np.asfortranarray(sc.rp.RxSteeringVec.astype(np.complex128)).tobytes(order="F"))
        f.write(np.ones(int(sc.rp.nTargets), np.uint8).tobytes())
        for a in (sc.tx_wave, sc.noise, sc.tx_grid):
            f.write(np.asfortranarray(a).tobytes(order="F"))
        f.write(struct.pack("<5d", float(sc.rp.Tsri), *np.asfortranarray(sc.rp.cfarEstZone, dtype=np.float64).ravel(order="F")))
    r = subprocess.run([_exe(), "chain", str(fin), str(fout)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.split()[-4:] == ["isac:NO_LOS", "isac:INVALID_ARG", "isac:INVALID_ARG", "isac:UNSUPPORTED"]
    buf = open(fout, "rb").read()
    off = 0
    echo1 = None
    for path in ("MATLAB arrays", "device handles", "fused entry + cached fft2D", "fused entry with a LAZY echo grid + materializeEcho"):
        echo, (rng, vel, azi), off = _est(buf, off)
        echo = echo.reshape((sc.K, sc.L, sc.A), order="F")
        echo1 = echo if echo1 is None else echo1
        sub = echo[::5, ::3, :]
        assert np.abs(sub - g["echo_grid_sub"]).max() <= 1e-10 * np.abs(g["echo_grid_sub"]).max(), path
        assert np.array_equal(rng, g["rngEst"]) and np.array_equal(vel, g["velEst"]) and np.array_equal(azi, g["aziEst"]), path
    # DoA entries on Ra = X X' / N of the echo grid
    gm = echo1.reshape(-1, sc.A, order="F")
    ra = gm.conj().T @ gm / gm.shape[0]
    ra = (ra + ra.conj().T) / 2
    nd = int(g["rngEst"].size)
    (l_music,) = struct.unpack_from("<i", buf, off); off += 4
    azi_music, off = _vec(buf, off)
    azi_bf, off = _vec(buf, off)
    azi_mvdr, off = _vec(buf, off)
    want_l, want_azi, _ = O.music_doa(nd, sc.rp, ra)
    assert l_music == want_l and np.array_equal(azi_music, want_azi) and np.array_equal(azi_music, g["aziEst"])
    assert np.array_equal(azi_bf, O.digital_bf(nd, sc.rp, ra)[0]) and np.array_equal(azi_mvdr, O.mvdr_bf(nd, sc.rp, ra)[0])
    # basicRadarChannel
    (n_w,) = struct.unpack_from("<Q", buf, off); off += 8
    rxw = np.frombuffer(buf, dtype=np.complex128, count=n_w, offset=off).reshape((sc.T, sc.A), order="F"); off += 16 * n_w
    want_w = O.basic_radar_channel(sc.tx_wave, sc.rp, sc.los, sc.noise)
    assert np.abs(rxw - want_w).max() <= 1e-10 * np.abs(want_w).max()
    # the stand-alone detector
    det, off = _vec(buf, off)
    det = det.reshape((2, -1), order="F").astype(np.int64)
    rows, cols = np.meshgrid(np.arange(10, 21), np.arange(5, 13), indexing="ij")
    cut = np.vstack([rows.ravel(order="F"), cols.ravel(order="F")])
    want_det = O.ca_cfar2d(np.abs(echo1[:, :, 0]) ** 2, cut, 0.3)
    assert np.array_equal(det, want_det) and det.shape[1] > 0
    # music2D
    _, (rng2, vel2, azi2), off = _est(buf, off)
    want2 = O.music2d(sc.rp, 30, echo1, sc.tx_grid)
    assert np.array_equal(rng2, want2.rngEst) and np.array_equal(vel2, want2.velEst) and np.array_equal(azi2, want2.aziEst)
    assert off == len(buf)


def _same(a, b):
    a, b = np.asarray(a, dtype=np.float64).reshape(-1), np.asarray(b, dtype=np.float64).reshape(-1)
    return a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])


def test_mex_gateway_communication_and_topology_entries(tmp_path):
    """applyCDL (toolbox path gains + path filters in, delay filtering and antenna contraction on the device), precodedSINR, csiReport,
    senTxAppend on allocDevice handles, checkLoS -- each through mexFunction(), against plain NumPy / the oracle."""
    from types import SimpleNamespace
    from scipy.signal import lfilter
    import oracle.cqi as OQ
    import oracle.pmi as OP
    from oracle import los as OL
    from conftest import load_pkg
    from test_gpu_csi import channel
    from test_gpu_los import _random_city
    from test_gpu_sentx import qpsk
    pkg = load_pkg()
    rng = np.random.default_rng(11)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    cn = lambda *s: rng.standard_normal(s) + 1j * rng.standard_normal(s)
    with open(fin, "wb") as f:
        # ---- applyCDL: UL-like shape (2 transmit, 8 receive antennas), 3 gain snapshots, 5 paths, 16-tap path filters
        T, Nt, Nr, Np, Ncs, Nh, fs = 3000, 2, 8, 5, 3, 16, 61.44e6
        x, pg = cn(T, Nt), cn(Ncs, Np, Nt, Nr) / np.sqrt(Np)
        st = 1e-3 + np.arange(Ncs) * (1000 / fs)                                   # snapshot b starts 1000 samples after snapshot b-1
        pf = rng.standard_normal((Nh, Np)) / 4
        f.write(struct.pack("<7i d", T, Nt, Nr, Np, Ncs, Nh, 1, fs))
        for a in (x, pg, st, pf):
            f.write(np.asfortranarray(a).tobytes(order="F"))
        # ---- precodedSINR
        h1, w1, sigma = cn(2, 4), cn(4, 2) / 2, 0.3
        f.write(struct.pack("<3i d", 2, 4, 2, sigma))
        f.write(np.asfortranarray(h1).tobytes(order="F")); f.write(np.asfortranarray(w1).tobytes(order="F"))
        # ---- csiReport: 52 PRB, 4 ports, rank 2, subband PMI and CQI
        nrb, ports, layers, sbsize, nvar = 52, 4, 2, 8, 0.02
        rep = SimpleNamespace(NSizeBWP=nrb, NStartBWP=0, PanelDimensions=(2, 1), CodebookMode=1, PMIMode="Subband", CQIMode="Subband", SubbandSize=sbsize)
        hc = channel(np.random.default_rng(3), nrb, 2, ports) * 3.0
        k = np.concatenate([[12 * r + 1, 12 * r + 2] for r in range(nrb)])
        l = np.ones_like(k)
        hre = np.asfortranarray(hc[k - 1, l - 1, :, :])
        f.write(struct.pack("<10i d 2d", k.size, 2, ports, layers, nrb, 0, sbsize, 1, 1, 1, nvar, 2.0, 1.0))
        f.write(k.astype(np.float64).tobytes()); f.write(l.astype(np.float64).tobytes()); f.write(hre.tobytes(order="F"))
        tab = np.asarray(OQ.DOWNLINK_SINR90PC, dtype=np.float64)
        f.write(struct.pack("<i", tab.size)); f.write(tab.tobytes())
        # ---- srsReportBatch: 52 PRB, 8 receive elements, 2 SRS ports, comb 2 on RBs 4..47 (leading / trailing bands take the mean PMI), three UEs
        import oracle.srs as OS
        nrb_s, band_s, R_s, P_s, nu_s = 52, 4, 8, 2, 3
        mask = np.zeros(12 * nrb_s, dtype=bool); mask[12 * 4:12 * 48:2] = True
        k_s = np.flatnonzero(mask)
        rs = np.random.default_rng(17)
        hs_full, nv_s = [], [0.02, 0.2, 1.0]
        for u in range(nu_s):
            g_ = (rs.standard_normal((4, R_s, P_s)) + 1j * rs.standard_normal((4, R_s, P_s))) / np.sqrt(8)
            ph = np.exp(-2j * np.pi * np.outer(np.arange(12 * nrb_s), rs.uniform(0, 60, 4)) / 4096)
            hs_full.append(np.einsum("kt,trp->krp", ph, g_) * [0.1, 0.5, 2.0][u] * mask[:, None, None])
        f.write(struct.pack("<7i", k_s.size, R_s, P_s, nu_s, nrb_s, band_s, 1))
        f.write((k_s + 1).astype(np.float64).tobytes()); f.write(np.asarray(nv_s, dtype=np.float64).tobytes())
        f.write(np.asfortranarray(np.stack([h[k_s] for h in hs_full], axis=3)).tobytes(order="F"))
        tab_ul = np.asarray(OQ.UPLINK_SINR90PC, dtype=np.float64)
        f.write(struct.pack("<i", tab_ul.size)); f.write(tab_ul.tobytes())
        # ---- senTx accumulation: 24 PRB, 3 antennas, DDDSU, windowing 18
        nrb2, A2, tdd, win, nfft2 = 24, 3, "DDDSU", 18, 512
        slots = [s for s in range(5) if tdd[s % 5] != "U"]
        ref = O.sentx.SenTx(nfft2, 30, tdd, 46.0, windowing=win)
        amp = O.sentx.signal_amp(46.0, nfft2, 12 * nrb2, A2)
        t_slot = O.ofdm_modulate(np.zeros((12 * nrb2, 14, 1), dtype=complex), nfft2, 30).shape[0]
        f.write(struct.pack("<6i d", nrb2, A2, len(slots), 30, win, t_slot, amp))
        for s in slots:
            g = qpsk((12 * nrb2, 14, A2), 200 + s)
            ref.append(g, s)
            f.write(struct.pack("<2i", s, 1 if tdd[s % 5] == "D" else 0)); f.write(g.tobytes(order="F"))
        # ---- senTx accumulation at 60 kHz: 24 PRB, 2 antennas, slots 0..5 of DDDSU -- slots 0 / 2 / 4 carry the long cyclic prefix,
        # the accumulators are over-sized by two slots and trimmed (SenTx.m: running sample count + 'trim')
        nrb3, A3, win3, nfft3, scs3 = 24, 2, 9, 512, 60
        slots3 = [s for s in range(6) if tdd[s % 5] != "U"]
        ref3 = O.sentx.SenTx(nfft3, scs3, tdd, 46.0, windowing=win3)
        amp3 = O.sentx.signal_amp(46.0, nfft3, 12 * nrb3, A3)
        t_cap3 = max(O.ofdm_modulate(np.zeros((12 * nrb3, 14, 1), dtype=complex), nfft3, scs3, 0, 14 * q).shape[0] for q in range(4))
        f.write(struct.pack("<6i d", nrb3, A3, len(slots3), scs3, win3, t_cap3, amp3))
        for s in slots3:
            g = qpsk((12 * nrb3, 14, A3), 300 + s)
            ref3.append(g, s)
            f.write(struct.pack("<2i", s, 1 if tdd[s % 5] == "D" else 0)); f.write(g.tobytes(order="F"))
        lens3 = {O.ofdm_modulate(np.zeros((12 * nrb3, 14, 1), dtype=complex), nfft3, scs3, 0, 14 * q).shape[0] for q in range(4)}
        assert len(lens3) == 2                                   # the slots of a 60 kHz subframe really differ in length
        # ---- checkLoS
        plans, heights = _random_city(np.random.default_rng(4), 20)
        B = pkg.networkTopology.blockages
        walls = [w for fp, hgt in zip(plans, heights) for w in B.building(fp, hgt, 3.0).wallList]
        from importlib import import_module
        pack = import_module(pkg.__name__ + ".networkTopology.blockages.wallBlockage").pack_walls
        co, off, no, nd = pack(walls)
        n_links = 500
        ue = np.asfortranarray(np.stack([rng.uniform(-400, 400, n_links), rng.uniform(-400, 400, n_links), np.full(n_links, 1.5)]))
        ant = np.asfortranarray(np.repeat(np.array([[0.0], [0.0], [30.0]]), n_links, axis=1))
        f.write(struct.pack("<3i", co.shape[1], off.size - 1, n_links))
        for a in (co, off, no, nd, ue, ant):
            f.write(np.asfortranarray(a).tobytes(order="F"))
        # ---- prgPrecode: 24 PRB, 3 layers onto 16 antennas, 5 PRGs, carrier starting at CRB 3, a PDSCH-like RE set
        import oracle.precode as OPR
        Kp, Lp, nup, Pp, nprg, nstart = 288, 7, 3, 16, 5, 3
        kk = np.arange(Kp)
        re_p = np.concatenate([(kk if l_ % 3 else kk[kk % 2 == 0]) + Kp * l_ for l_ in range(1, Lp)])
        pind = (re_p[:, None] + Kp * Lp * np.arange(nup)[None, :] + 1).astype(np.float64)
        psym = cn(re_p.size, nup)
        Fp = cn(nup, Pp, nprg) / 4
        f.write(struct.pack("<7i", Kp, Lp, nup, Pp, nprg, nstart, re_p.size))
        f.write(np.asfortranarray(pind).tobytes(order="F")); f.write(np.asfortranarray(psym).tobytes(order="F")); f.write(np.asfortranarray(Fp).tobytes(order="F"))
    r = subprocess.run([_exe(), "comm", str(fin), str(fout)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    buf = open(fout, "rb").read()
    off_b = 0
    # applyCDL
    (n,) = struct.unpack_from("<Q", buf, off_b); off_b += 8
    y = np.frombuffer(buf, dtype=np.complex128, count=n, offset=off_b).reshape((T, Nr), order="F"); off_b += 16 * n
    start = np.rint((st - st[0]) * fs).astype(int)
    blk = np.searchsorted(start, np.arange(T), side="right") - 1
    want = np.zeros((T, Nr), dtype=complex)
    for b in range(Ncs):
        acc = sum(lfilter(pf[:, p], [1.0], x @ pg[b, p], axis=0) for p in range(Np))
        want[blk == b] = acc[blk == b]
    want /= np.sqrt(Nr)
    assert np.abs(y - want).max() <= 1e-10 * np.abs(want).max()
    # precodedSINR
    (sinr,) = struct.unpack_from("<d", buf, off_b); off_b += 8
    den = (sigma ** 2 * np.eye(2)) @ np.linalg.inv((w1.conj().T @ h1.conj().T) @ h1 @ w1 + sigma ** 2 * np.eye(2))
    assert abs(sinr - np.real(np.sum(1.0 / np.diag(den) - 1.0))) <= 1e-10 * abs(sinr)
    # csiReport
    want_cqi, want_pmi, want_ci, _ = OP.cqi_select(rep, layers, hc, k, l, nvar, OQ.DOWNLINK_SINR90PC)
    cqi, off_b = _vec(buf, off_b); i1, off_b = _vec(buf, off_b); i2, off_b = _vec(buf, off_b)
    sbc, off_b = _vec(buf, off_b); sps, off_b = _vec(buf, off_b)
    assert _same(cqi, want_cqi) and _same(i1, want_pmi.i1) and _same(i2, want_pmi.i2) and _same(sbc, want_ci.SubbandCQI)
    b_ = np.asarray(want_ci.SINRPerSubbandPerCW, dtype=np.float64).reshape(-1)
    assert np.array_equal(np.isnan(sps), np.isnan(b_)) and np.abs(sps[~np.isnan(sps)] - b_[~np.isnan(b_)]).max() <= 1e-10 * np.nanmax(np.abs(b_))
    # csiReportBatch's sixth output: riSelect's totalSINR of the rank for the three noise variances
    ri_tot, off_b = _vec(buf, off_b)
    for u, nv_ in enumerate((nvar, 4.0 * nvar, 0.25 * nvar)):
        pmi_u, info_u = OP.dl_pmi_select(rep, layers, hc, k, l, nv_)
        i11, i12, i13 = (int(v) - 1 for v in pmi_u.i1)
        sub_ = np.stack([info_u.SINRPerSubband[s_, :, int(pmi_u.i2[s_]) - 1, i11, i12, i13] * layers for s_ in range(pmi_u.i2.size)])
        lay = np.nanmean(sub_, axis=0)
        assert abs(ri_tot[u] - np.sum(lay[lay >= 1])) <= 1e-10 * max(1.0, abs(ri_tot[u]))
    # srsReportBatch
    g_pmi, off_b = _vec(buf, off_b); g_sel, off_b = _vec(buf, off_b); g_cqi, off_b = _vec(buf, off_b)
    n_sb = -(-nrb_s // band_s)
    for u in range(nu_s):
        w_pmi, w_sel, w_cqi = OS.srs_report(1, hs_full[u][:, None, :, :], nv_s[u], band_s, nrb_s, OQ.UPLINK_SINR90PC)
        assert np.array_equal(g_pmi[u * n_sb:(u + 1) * n_sb], w_pmi) and np.array_equal(g_cqi[u * nrb_s:(u + 1) * nrb_s], w_cqi)
        assert np.abs(g_sel[u * n_sb:(u + 1) * n_sb] - w_sel).max() <= 1e-10 * np.abs(w_sel).max()
    # senTx accumulators
    (n,) = struct.unpack_from("<Q", buf, off_b); off_b += 8
    grid = np.frombuffer(buf, dtype=np.complex128, count=n, offset=off_b).reshape(ref.grid.shape, order="F"); off_b += 16 * n
    (n,) = struct.unpack_from("<Q", buf, off_b); off_b += 8
    wave = np.frombuffer(buf, dtype=np.complex128, count=n, offset=off_b).reshape(ref.wave.shape, order="F"); off_b += 16 * n
    assert np.array_equal(grid, ref.grid) and np.abs(wave - ref.wave).max() <= 1e-10 * np.abs(ref.wave).max()
    # senTx accumulators at 60 kHz: exactly the reference's cat() results (no gaps between slots of different length, no zero tail)
    (n,) = struct.unpack_from("<Q", buf, off_b); off_b += 8
    assert n == ref3.grid.size
    grid3 = np.frombuffer(buf, dtype=np.complex128, count=n, offset=off_b).reshape(ref3.grid.shape, order="F"); off_b += 16 * n
    (n,) = struct.unpack_from("<Q", buf, off_b); off_b += 8
    assert n == ref3.wave.size
    wave3 = np.frombuffer(buf, dtype=np.complex128, count=n, offset=off_b).reshape(ref3.wave.shape, order="F"); off_b += 16 * n
    assert np.array_equal(grid3, ref3.grid) and np.abs(wave3 - ref3.wave).max() <= 1e-10 * np.abs(ref3.wave).max()
    # checkLoS
    los = np.frombuffer(buf, dtype=np.uint8, count=n_links, offset=off_b).astype(bool); off_b += n_links
    want_los = OL.check_los(list(zip(plans, heights)), ue.T, ant.T)
    assert np.array_equal(los, want_los) and 0 < los.sum() < n_links
    # prgPrecode: [antsym, antind] against the loop-for-loop restatement of prgPrecode.m
    (n,) = struct.unpack_from("<Q", buf, off_b); off_b += 8
    antsym = np.frombuffer(buf, dtype=np.complex128, count=n, offset=off_b).reshape((re_p.size, Pp), order="F"); off_b += 16 * n
    antind = np.frombuffer(buf, dtype=np.float64, count=re_p.size * Pp, offset=off_b).reshape((re_p.size, Pp), order="F"); off_b += 8 * re_p.size * Pp
    want_sym, want_ind = OPR.prg_precode((Kp, Lp), nstart, psym, pind.astype(np.int64), Fp)
    assert np.array_equal(antind, want_ind) and np.abs(antsym - want_sym).max() <= 1e-12 * np.abs(want_sym).max()
    assert off_b == len(buf)
