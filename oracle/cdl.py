"""Oracle restatement of the CDL MIMO channel apply the reference performs with the 5G Toolbox object
``nrCDLChannel`` (configured in +parameters/+channelModels/+communication/cdl.m:57-64,78-85; profile chosen by
+communication/+channelModels/updateCDLModels.m:9-14; stepped at +communication/+phyLayer/uePhy.m:729-731 and
gNBPhy.m:838-840).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED, doubly so: ``nrCDLChannel`` is proprietary, its source is not under /root/reference, and its random
initial phases / ray couplings come from MATLAB's mt19937ar stream (Seed 73).  What is restated here is the PUBLISHED
algorithm -- 3GPP TR 38.901 v16 section 7.7.1 (CDL tables 7.7.1-1 'CDL-A', 7.7.1-4 'CDL-D', ray offsets table 7.5-3,
element pattern table 7.3-1, polarisation model-2 of 7.3.2, channel coefficients eq. 7.5-22/-28/-29) with the
toolbox's documented defaults (MaximumDopplerShift 5 Hz, UTDirectionOfTravel [0;90], SampleDensity 64 with
sample-and-hold path gains, NormalizePathGains, NormalizeChannelOutputs, Tx element '38.901' slanted +-45 deg,
Rx element isotropic [0 90], 0.5 lambda spacing).  Random draws come from the Philox generator of oracle/philox.py, so
parity with MATLAB can only be statistical; parity of the HIP path with THIS restatement is exact to rounding.
Oracle-defined choices (unobservable toolbox internals): 16-tap Hann-windowed-sinc fractional-delay channel filter with
7 samples of filter delay; element index = m + M (n + N (p + P (mg + Mg ng))).
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np

from .philox import philox4x32_10

LIGHTSPEED = 299792458.0

# TR 38.901 Table 7.5-3: ray offset angles within a cluster (normalised to 1 deg rms)
RAY_OFFSETS = np.array([0.0447, -0.0447, 0.1413, -0.1413, 0.2492, -0.2492, 0.3715, -0.3715, 0.5129, -0.5129,
                        0.6797, -0.6797, 0.8844, -0.8844, 1.1481, -1.1481, 1.5195, -1.5195, 2.1551, -2.1551])

# columns: normalised delay, power dB, AOD, AOA, ZOD, ZOA  (deg)
CDL_A = np.array([
    [0.0000, -13.4, -178.1, 51.3, 50.2, 125.4], [0.3819, 0.0, -4.2, -152.7, 93.2, 91.3], [0.4025, -2.2, -4.2, -152.7, 93.2, 91.3],
    [0.5868, -4.0, -4.2, -152.7, 93.2, 91.3], [0.4610, -6.0, 90.2, 76.6, 122.0, 94.0], [0.5375, -8.2, 90.2, 76.6, 122.0, 94.0],
    [0.6708, -9.9, 90.2, 76.6, 122.0, 94.0], [0.5750, -10.5, 121.5, -1.8, 150.2, 47.1], [0.7618, -7.5, -81.7, -41.9, 55.2, 56.0],
    [1.5375, -15.9, 158.4, 94.2, 26.4, 30.1], [1.8978, -6.6, -83.0, 51.9, 126.4, 58.8], [2.2242, -16.7, 134.8, -115.9, 171.6, 26.0],
    [2.1718, -12.4, -153.0, 26.6, 151.4, 49.2], [2.4942, -15.2, -172.0, 76.6, 157.2, 143.1], [2.5119, -10.8, -129.9, -7.0, 47.2, 117.4],
    [3.0582, -11.3, -136.0, -23.0, 40.4, 122.7], [4.0810, -12.7, 165.4, -47.2, 43.3, 123.2], [4.4579, -16.2, 148.4, 110.4, 161.8, 32.6],
    [4.5695, -18.3, 132.7, 144.5, 10.8, 27.2], [4.7966, -18.9, -118.6, 155.3, 16.7, 15.2], [5.0066, -16.6, -154.1, 102.0, 171.7, 146.0],
    [5.3043, -19.9, 126.5, -151.8, 22.7, 150.7], [9.6586, -29.7, -56.2, 55.2, 144.9, 156.1]])
CDL_A_SPREADS = dict(cASD=5.0, cASA=11.0, cZSD=3.0, cZSA=3.0, XPR=10.0)

# first row = specular LOS path; second row = its Rayleigh part (same delay/angles)
CDL_D = np.array([
    [0.0000, -0.2, 0.0, -180.0, 98.5, 81.5], [0.0000, -13.5, 0.0, -180.0, 98.5, 81.5],
    [0.0350, -18.8, 89.2, 89.2, 85.5, 86.9], [0.6120, -21.0, 89.2, 89.2, 85.5, 86.9], [1.3630, -22.8, 89.2, 89.2, 85.5, 86.9],
    [1.4050, -17.9, 13.0, 163.0, 97.5, 79.4], [1.8040, -20.1, 13.0, 163.0, 97.5, 79.4], [2.5960, -21.9, 13.0, 163.0, 97.5, 79.4],
    [1.7750, -22.9, 34.6, -137.0, 98.5, 78.2], [4.0420, -27.8, -64.5, 74.5, 88.4, 73.6], [7.9370, -23.6, -32.9, 127.7, 91.3, 78.3],
    [9.4240, -24.8, 52.6, -119.6, 103.8, 87.0], [9.7080, -30.0, -132.1, -9.1, 80.3, 70.6], [12.5250, -27.7, 77.2, -83.8, 86.5, 72.9]])
CDL_D_SPREADS = dict(cASD=5.0, cASA=8.0, cZSD=3.0, cZSA=3.0, XPR=11.0)

PROFILES = {"CDL-A": (CDL_A, CDL_A_SPREADS, False), "CDL-D": (CDL_D, CDL_D_SPREADS, True)}
FILTER_TAPS = 16
FILTER_DELAY = 7


def cdl_config(delay_profile, carrier_frequency, tx_size, rx_size, sample_rate, delay_spread=300e-9,
               max_doppler=5.0, seed=73, sample_density=64):
    """The properties cdl.m:57-64 sets plus the toolbox defaults the reference leaves untouched."""
    return SimpleNamespace(DelayProfile=delay_profile, DelaySpread=delay_spread, CarrierFrequency=float(carrier_frequency),
                           TxSize=tuple(int(v) for v in tx_size), RxSize=tuple(int(v) for v in rx_size),
                           SampleRate=float(sample_rate), MaximumDopplerShift=float(max_doppler), Seed=int(seed),
                           SampleDensity=int(sample_density), UTDirectionOfTravel=(0.0, 90.0),
                           TxPolAngles=(45.0, -45.0), RxPolAngles=(0.0, 90.0), TxElement="38.901", RxElement="isotropic",
                           NormalizePathGains=True, NormalizeChannelOutputs=True)


def _uniform(seed, stream, n):
    """n uniforms in [0,1) from Philox counter (i, 0, stream, 0xCD1), key = seed."""
    i = np.arange(n, dtype=np.uint64)
    x0, x1, _, _ = philox4x32_10((i & np.uint64(0xFFFFFFFF)).astype(np.uint32), np.uint32(0), np.uint32(stream), np.uint32(0xCD1),
                                 np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF))
    w = x0.astype(np.uint64) | (x1.astype(np.uint64) << np.uint64(32))
    return (w >> np.uint64(11)).astype(np.float64) * 2.0 ** -53


def element_positions(size):
    """[n_elem x 3] positions in wavelengths (array in the y-z plane, 0.5 lambda spacing) + polarisation index."""
    m_, n_, p_, mg_, ng_ = size
    pos, pol = [], []
    for ng in range(ng_):
        for mg in range(mg_):
            for p in range(p_):
                for n in range(n_):
                    for m in range(m_):
                        pos.append([0.0, 0.5 * n + ng * 0.5 * n_, 0.5 * m + mg * 0.5 * m_])
                        pol.append(p)
    return np.array(pos), np.array(pol)


def field_pattern(theta_deg, phi_deg, element, slant_deg):
    """(F_theta, F_phi) of one element, polarisation model-2 (TR 38.901 7.3.2, eq. 7.3-4/5)."""
    if element == "38.901":
        phi = (np.asarray(phi_deg) + 180.0) % 360.0 - 180.0
        a_v = -np.minimum(12.0 * ((np.asarray(theta_deg) - 90.0) / 65.0) ** 2, 30.0)
        a_h = -np.minimum(12.0 * (phi / 65.0) ** 2, 30.0)
        a_db = -np.minimum(-(a_v + a_h), 30.0) + 8.0
        amp = 10.0 ** (a_db / 20.0)
    else:
        amp = np.ones_like(np.asarray(theta_deg, dtype=np.float64))
    z = math.radians(slant_deg)
    return amp * math.cos(z), amp * math.sin(z)


def _unit(theta_deg, phi_deg):
    t, p = np.radians(theta_deg), np.radians(phi_deg)
    return np.stack([np.sin(t) * np.cos(p), np.sin(t) * np.sin(p), np.cos(t)], axis=-1)


def draw_rays(cfg):
    """Deterministic (seeded) ray angles, couplings and initial phases for every cluster."""
    tab, spr, has_los = PROFILES[cfg.DelayProfile]
    n_cl = tab.shape[0] - (1 if has_los else 0)            # NLOS clusters (the LOS row is separate)
    nl = tab[1:] if has_los else tab
    m = RAY_OFFSETS.size
    u = _uniform(cfg.Seed, 1, n_cl * m * 4).reshape(n_cl, m, 4)
    phases = (2.0 * u - 1.0) * np.pi                        # Phi^{tt,tp,pt,pp} ~ U(-pi, pi)
    perm_u = _uniform(cfg.Seed, 2, n_cl * m * 3).reshape(n_cl, 3, m)
    perms = np.argsort(perm_u, axis=2, kind="stable")       # random couplings AOA / ZOD / ZOA vs AOD (step 8)
    aod = nl[:, 2:3] + spr["cASD"] * RAY_OFFSETS[None, :]
    aoa = nl[:, 3:4] + spr["cASA"] * RAY_OFFSETS[perms[:, 0, :]]
    zod = nl[:, 4:5] + spr["cZSD"] * RAY_OFFSETS[perms[:, 1, :]]
    zoa = nl[:, 5:6] + spr["cZSA"] * RAY_OFFSETS[perms[:, 2, :]]
    p_lin = 10.0 ** (tab[:, 1] / 10.0)
    if cfg.NormalizePathGains:
        p_lin = p_lin / p_lin.sum()
    return SimpleNamespace(aod=aod, aoa=aoa, zod=zod, zoa=zoa, phases=phases, power=p_lin, has_los=has_los, table=tab,
                           kappa=10.0 ** (spr["XPR"] / 10.0))


def path_delays(cfg):
    tab, _, has_los = PROFILES[cfg.DelayProfile]
    d = tab[:, 0] * cfg.DelaySpread
    return d[1:] if has_los else d                          # the LOS row shares the first path's delay


def channel_info(cfg):
    d = path_delays(cfg)
    return SimpleNamespace(PathDelays=d, ChannelFilterDelay=FILTER_DELAY,
                           MaxChannelDelay=int(math.ceil(np.max(d * cfg.SampleRate))) + FILTER_DELAY)


def path_gains(cfg, t_snap):
    """H[n, s, u] at snapshot time t_snap [s]  (TR 38.901 eq. 7.5-22, LOS 7.5-29 folded into path 0 for CDL-D)."""
    rays = draw_rays(cfg)
    lam_fc = 1.0                                            # positions are already in wavelengths
    txp, txpol = element_positions(cfg.TxSize)
    rxp, rxpol = element_positions(cfg.RxSize)
    vhat = _unit(cfg.UTDirectionOfTravel[1], cfg.UTDirectionOfTravel[0])
    n_paths = rays.aod.shape[0]
    h = np.zeros((n_paths, txp.shape[0], rxp.shape[0]), dtype=np.complex128)
    sk = math.sqrt(1.0 / rays.kappa)
    p_nl = rays.power[1:] if rays.has_los else rays.power
    for n in range(n_paths):
        for m in range(RAY_OFFSETS.size):
            r_tx = _unit(rays.zod[n, m], rays.aod[n, m])
            r_rx = _unit(rays.zoa[n, m], rays.aoa[n, m])
            ph = rays.phases[n, m]
            xp = np.array([[np.exp(1j * ph[0]), sk * np.exp(1j * ph[1])], [sk * np.exp(1j * ph[2]), np.exp(1j * ph[3])]])
            dop = np.exp(2j * np.pi * cfg.MaximumDopplerShift * float(r_rx @ vhat) * t_snap)
            a_tx = np.exp(2j * np.pi * (txp @ r_tx) / lam_fc)
            a_rx = np.exp(2j * np.pi * (rxp @ r_rx) / lam_fc)
            for s in range(txp.shape[0]):
                ft = np.array(field_pattern(rays.zod[n, m], rays.aod[n, m], cfg.TxElement, cfg.TxPolAngles[txpol[s]]))
                for u in range(rxp.shape[0]):
                    fr = np.array(field_pattern(rays.zoa[n, m], rays.aoa[n, m], cfg.RxElement, cfg.RxPolAngles[rxpol[u]]))
                    h[n, s, u] += (fr @ xp @ ft) * a_rx[u] * a_tx[s] * dop
        h[n] *= math.sqrt(p_nl[n] / RAY_OFFSETS.size)
    if rays.has_los:
        row = rays.table[0]
        r_tx, r_rx = _unit(row[4], row[2]), _unit(row[5], row[3])
        dop = np.exp(2j * np.pi * cfg.MaximumDopplerShift * float(r_rx @ vhat) * t_snap)
        a_tx = np.exp(2j * np.pi * (txp @ r_tx))
        a_rx = np.exp(2j * np.pi * (rxp @ r_rx))
        for s in range(txp.shape[0]):
            ft = np.array(field_pattern(row[4], row[2], cfg.TxElement, cfg.TxPolAngles[txpol[s]]))
            for u in range(rxp.shape[0]):
                fr = np.array(field_pattern(row[5], row[3], cfg.RxElement, cfg.RxPolAngles[rxpol[u]]))
                h[0, s, u] += math.sqrt(rays.power[0]) * (fr[0] * ft[0] - fr[1] * ft[1]) * a_rx[u] * a_tx[s] * dop
    return h


def filter_taps(cfg):
    """[n_paths x FILTER_TAPS] fractional-delay taps and the integer sample shift of each path."""
    d = path_delays(cfg) * cfg.SampleRate
    shift = np.floor(d).astype(np.int64)
    frac = d - shift
    k = np.arange(FILTER_TAPS, dtype=np.float64)
    x = k[None, :] - FILTER_DELAY - frac[:, None]
    g = np.sinc(x) * np.where(np.abs(x) < FILTER_TAPS / 2, 0.5 + 0.5 * np.cos(np.pi * x / (FILTER_TAPS / 2)), 0.0)
    return g, shift


def snapshot_index(cfg, t):
    """Sample-and-hold: path gains are refreshed every 1/(2 SampleDensity fD) seconds."""
    rate = 2.0 * cfg.SampleDensity * cfg.MaximumDopplerShift
    return np.floor(np.asarray(t) * rate + 1e-9).astype(np.int64), rate


def apply_cdl(cfg, waveform, t0=0.0, order=None):
    """y = channel(waveform):  waveform [T x Nt] -> [T x Nr]; t0 = channel time of the first sample [s].
    y[t,u] = (1/sqrt(Nr)) sum_n sum_k g_n[k] sum_s H_n^{(b(t))}[s,u] x_s[t - shift_n - k].
    ``order``: the two sums over k and s commute -- "contract_first" forms z = x H_n [T x Nr] and filters it (the literal reading of the line above),
    "filter_first" filters x [T x Nt] and contracts the result; the same numbers to rounding (tests/test_cdl_cpu.py), 30x cheaper for the uplink's
    2 -> 64 shape at config 5's 61 909 samples.  Default: whichever touches fewer elements."""
    x = np.asarray(waveform, dtype=np.complex128)
    t_len, nt = x.shape
    g, shift = filter_taps(cfg)
    tt = t0 + np.arange(t_len) / cfg.SampleRate
    blk, rate = snapshot_index(cfg, tt)
    nr = int(np.prod(cfg.RxSize))
    if order is None:
        order = "filter_first" if nr > nt else "contract_first"
    y = np.zeros((t_len, nr), dtype=np.complex128)
    xf = None
    if order == "filter_first":                              # the delayed / filtered transmit signals do not depend on the gain block
        xf = np.zeros((g.shape[0], t_len, nt), dtype=np.complex128)
        for n in range(g.shape[0]):
            for k in range(FILTER_TAPS):
                lag = int(shift[n]) + k
                if lag < t_len:
                    xf[n, lag:] += g[n, k] * x[: t_len - lag]
    for b in np.unique(blk):
        h = path_gains(cfg, b / rate)                        # [n, s, u]
        sel = blk == b
        for n in range(h.shape[0]):
            if xf is not None:
                y[sel] += xf[n][sel] @ h[n]
                continue
            z = x @ h[n]                                     # [T x Nr] contraction over transmit antennas
            acc = np.zeros((t_len, nr), dtype=np.complex128)
            for k in range(FILTER_TAPS):
                lag = int(shift[n]) + k
                if lag < t_len:
                    acc[lag:] += g[n, k] * z[: t_len - lag]
            y[sel] += acc[sel]
    if cfg.NormalizeChannelOutputs:
        y = y / math.sqrt(nr)
    return y


def freq_response(cfg, t, k_sub, n_sc, scs_hz, ports):
    """Perfect channel estimate at the 1-based subcarriers k_sub: Hf[i, u, p] = sum_n h[n, p, u](t_blk) exp(-2 pi j f_i tau_n), f_i = (k_i - 1 - n_sc / 2) scs, with the
    sample-and-hold path gains of the gain block that holds t (TR 38.901 7.5-22 evaluated in the frequency domain; what nrPerfectChannelEstimate returns for a
    channel whose filters are ideal delays).  [n_re x Nr x ports]."""
    blk, rate = snapshot_index(cfg, np.array([t]))
    h = path_gains(cfg, float(blk[0]) / rate)[:, :ports, :]                # [n, p, u]
    tau = path_delays(cfg)
    f = ((np.asarray(k_sub, dtype=np.float64) - 1.0) - n_sc / 2.0) * float(scs_hz)
    e = np.exp(-2j * np.pi * f[:, None] * tau[None, :])                    # [i, n]
    return np.einsum("kn,npu->kup", e, h)
