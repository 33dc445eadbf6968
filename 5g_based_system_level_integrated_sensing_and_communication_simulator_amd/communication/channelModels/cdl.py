"""CDL MIMO channel apply -- the seam the reference fills with the toolbox object ``nrCDLChannel``
(+parameters/+channelModels/+communication/cdl.m:57-64,78-85; stepped at uePhy.m:729-731, gNBPhy.m:838-840).

Host side (this file): TR 38.901 7.7.1 parameter preparation -- CDL-A / CDL-D cluster tables, 20-ray clusters,
38.901 element pattern, polarisation model-2, seeded ray phases/couplings, sample-and-hold path gains, fractional-delay
filter taps.  Scalar/array prep of a few 10^4 values per call, like ``sensing.radarParams``.
Device side (csrc/cdl.hip): the antenna contraction as one complex GEMM on fp64 MFMA + the per-path delay FIR.

The toolbox source is not in the reference and its random stream is MATLAB's; this follows the published 3GPP model
with the toolbox's documented defaults, so agreement with MATLAB is statistical only."""
from __future__ import annotations

import ctypes as C
import hashlib
import math
from types import SimpleNamespace

import numpy as np

from ... import _lib as L
from ..._philox import uniform

RAY_OFFSETS = np.array([0.0447, -0.0447, 0.1413, -0.1413, 0.2492, -0.2492, 0.3715, -0.3715, 0.5129, -0.5129,
                        0.6797, -0.6797, 0.8844, -0.8844, 1.1481, -1.1481, 1.5195, -1.5195, 2.1551, -2.1551])   # TR 38.901 Table 7.5-3
# TR 38.901 Table 7.7.1-1 (CDL-A) and 7.7.1-4 (CDL-D): normalised delay, power [dB], AOD, AOA, ZOD, ZOA [deg]
_TABLES = {
    "CDL-A": (np.array([
        [0.0000, -13.4, -178.1, 51.3, 50.2, 125.4], [0.3819, 0.0, -4.2, -152.7, 93.2, 91.3], [0.4025, -2.2, -4.2, -152.7, 93.2, 91.3],
        [0.5868, -4.0, -4.2, -152.7, 93.2, 91.3], [0.4610, -6.0, 90.2, 76.6, 122.0, 94.0], [0.5375, -8.2, 90.2, 76.6, 122.0, 94.0],
        [0.6708, -9.9, 90.2, 76.6, 122.0, 94.0], [0.5750, -10.5, 121.5, -1.8, 150.2, 47.1], [0.7618, -7.5, -81.7, -41.9, 55.2, 56.0],
        [1.5375, -15.9, 158.4, 94.2, 26.4, 30.1], [1.8978, -6.6, -83.0, 51.9, 126.4, 58.8], [2.2242, -16.7, 134.8, -115.9, 171.6, 26.0],
        [2.1718, -12.4, -153.0, 26.6, 151.4, 49.2], [2.4942, -15.2, -172.0, 76.6, 157.2, 143.1], [2.5119, -10.8, -129.9, -7.0, 47.2, 117.4],
        [3.0582, -11.3, -136.0, -23.0, 40.4, 122.7], [4.0810, -12.7, 165.4, -47.2, 43.3, 123.2], [4.4579, -16.2, 148.4, 110.4, 161.8, 32.6],
        [4.5695, -18.3, 132.7, 144.5, 10.8, 27.2], [4.7966, -18.9, -118.6, 155.3, 16.7, 15.2], [5.0066, -16.6, -154.1, 102.0, 171.7, 146.0],
        [5.3043, -19.9, 126.5, -151.8, 22.7, 150.7], [9.6586, -29.7, -56.2, 55.2, 144.9, 156.1]]),
        dict(cASD=5.0, cASA=11.0, cZSD=3.0, cZSA=3.0, XPR=10.0), False),
    "CDL-D": (np.array([
        [0.0000, -0.2, 0.0, -180.0, 98.5, 81.5], [0.0000, -13.5, 0.0, -180.0, 98.5, 81.5],
        [0.0350, -18.8, 89.2, 89.2, 85.5, 86.9], [0.6120, -21.0, 89.2, 89.2, 85.5, 86.9], [1.3630, -22.8, 89.2, 89.2, 85.5, 86.9],
        [1.4050, -17.9, 13.0, 163.0, 97.5, 79.4], [1.8040, -20.1, 13.0, 163.0, 97.5, 79.4], [2.5960, -21.9, 13.0, 163.0, 97.5, 79.4],
        [1.7750, -22.9, 34.6, -137.0, 98.5, 78.2], [4.0420, -27.8, -64.5, 74.5, 88.4, 73.6], [7.9370, -23.6, -32.9, 127.7, 91.3, 78.3],
        [9.4240, -24.8, 52.6, -119.6, 103.8, 87.0], [9.7080, -30.0, -132.1, -9.1, 80.3, 70.6], [12.5250, -27.7, 77.2, -83.8, 86.5, 72.9]]),
        dict(cASD=5.0, cASA=8.0, cZSD=3.0, cZSA=3.0, XPR=11.0), True),
}
FILTER_TAPS, FILTER_DELAY = 16, 7
_TAPS_CACHE: dict = {}
_STATIC_CACHE: dict = {}                      # time-independent channel quantities shared by equally configured channels


def _unit(theta_deg, phi_deg):
    t, p = np.radians(theta_deg), np.radians(phi_deg)
    return np.stack([np.sin(t) * np.cos(p), np.sin(t) * np.sin(p), np.cos(t)], axis=-1)


def _positions(size):
    m_, n_, p_, mg_, ng_ = (int(v) for v in size)
    idx = np.indices((ng_, mg_, p_, n_, m_)).reshape(5, -1)           # m fastest
    ng, mg, p, n, m = idx
    pos = np.stack([np.zeros(m.size), 0.5 * n + ng * 0.5 * n_, 0.5 * m + mg * 0.5 * m_], axis=1)
    return pos, p


def _pattern(theta, phi, element):
    if element == "38.901":                                            # TR 38.901 Table 7.3-1
        ph = (phi + 180.0) % 360.0 - 180.0
        a_v = -np.minimum(12.0 * ((theta - 90.0) / 65.0) ** 2, 30.0)
        a_h = -np.minimum(12.0 * (ph / 65.0) ** 2, 30.0)
        return 10.0 ** ((-np.minimum(-(a_v + a_h), 30.0) + 8.0) / 20.0)
    return np.ones_like(theta)


class CDLChannel:
    """The subset of ``nrCDLChannel`` the reference configures (cdl.m:57-64) with the toolbox defaults it leaves
    alone.  Stateful like the System object: channel time advances by the waveform duration on every call."""

    def __init__(self, DelayProfile="CDL-D", DelaySpread=300e-9, CarrierFrequency=3.5e9, TransmitAntennaArraySize=(1, 8, 2, 1, 1),
                 ReceiveAntennaArraySize=(1, 1, 2, 1, 1), SampleRate=122.88e6, MaximumDopplerShift=5.0, Seed=73, SampleDensity=64):
        if DelayProfile not in _TABLES:
            raise ValueError("DelayProfile must be 'CDL-A' or 'CDL-D' (updateCDLModels.m:9-14)")
        self.DelayProfile, self.DelaySpread, self.CarrierFrequency = DelayProfile, float(DelaySpread), float(CarrierFrequency)
        self.TransmitAntennaArraySize = tuple(int(v) for v in TransmitAntennaArraySize)
        self.ReceiveAntennaArraySize = tuple(int(v) for v in ReceiveAntennaArraySize)
        self.SampleRate, self.MaximumDopplerShift = float(SampleRate), float(MaximumDopplerShift)
        self.Seed, self.SampleDensity = int(Seed), int(SampleDensity)
        self.UTDirectionOfTravel = (0.0, 90.0)
        self.TxPolAngles, self.RxPolAngles = (45.0, -45.0), (0.0, 90.0)
        self.TxElement, self.RxElement = "38.901", "isotropic"
        self.NormalizePathGains = self.NormalizeChannelOutputs = True
        self.time = 0.0                                                 # InitialTime
        self.n_tx, self.n_rx = int(np.prod(self.TransmitAntennaArraySize)), int(np.prod(self.ReceiveAntennaArraySize))
        self._rays = None
        self._st = None

    # ---- info(channel): uePhy.m:288-289
    def info(self):
        d = self.path_delays()
        return SimpleNamespace(PathDelays=d, ChannelFilterDelay=FILTER_DELAY,
                               MaxChannelDelay=int(math.ceil(np.max(d * self.SampleRate))) + FILTER_DELAY)

    def path_delays(self):
        tab, _, los = _TABLES[self.DelayProfile]
        d = tab[:, 0] * self.DelaySpread
        return d[1:] if los else d

    def _draw(self):
        if self._rays is not None:
            return self._rays
        tab, spr, los = _TABLES[self.DelayProfile]
        nl = tab[1:] if los else tab
        n_cl, m = nl.shape[0], RAY_OFFSETS.size
        phases = (2.0 * uniform(self.Seed, 1, n_cl * m * 4).reshape(n_cl, m, 4) - 1.0) * np.pi
        perms = np.argsort(uniform(self.Seed, 2, n_cl * m * 3).reshape(n_cl, 3, m), axis=2, kind="stable")
        p_lin = 10.0 ** (tab[:, 1] / 10.0)
        if self.NormalizePathGains:
            p_lin = p_lin / p_lin.sum()
        self._rays = SimpleNamespace(
            aod=nl[:, 2:3] + spr["cASD"] * RAY_OFFSETS[None, :], aoa=nl[:, 3:4] + spr["cASA"] * RAY_OFFSETS[perms[:, 0, :]],
            zod=nl[:, 4:5] + spr["cZSD"] * RAY_OFFSETS[perms[:, 1, :]], zoa=nl[:, 5:6] + spr["cZSA"] * RAY_OFFSETS[perms[:, 2, :]],
            phases=phases, power=p_lin, los=los, table=tab, kappa=10.0 ** (spr["XPR"] / 10.0))
        return self._rays

    def _static(self):
        """Everything of eq. 7.5-22 / 7.5-29 that does not depend on time: per-ray antenna responses x polarisation
        coupling x array phases (`base` [n, m, s, u]), the per-ray Doppler rates, and the LOS term.  Computed once per
        channel object; a snapshot is then one weighted sum over the rays."""
        if getattr(self, "_st", None) is not None:
            return self._st
        key = (self.DelayProfile, self.DelaySpread, self.CarrierFrequency, self.TransmitAntennaArraySize, self.ReceiveAntennaArraySize,
               self.MaximumDopplerShift, self.Seed, self.UTDirectionOfTravel, self.TxPolAngles, self.RxPolAngles, self.TxElement,
               self.RxElement, self.NormalizePathGains)
        if key in _STATIC_CACHE:             # the reference gives every UE the same seed (cdl.m:57-64): identical ray draws
            self._st = _STATIC_CACHE[key]
            return self._st
        r = self._draw()
        txp, txpol = _positions(self.TransmitAntennaArraySize)
        rxp, rxpol = _positions(self.ReceiveAntennaArraySize)
        vhat = _unit(self.UTDirectionOfTravel[1], self.UTDirectionOfTravel[0])
        r_tx, r_rx = _unit(r.zod, r.aod), _unit(r.zoa, r.aoa)            # [n, m, 3]
        amp_t, amp_r = _pattern(r.zod, r.aod, self.TxElement), _pattern(r.zoa, r.aoa, self.RxElement)
        zt, zr = np.radians(np.asarray(self.TxPolAngles)[txpol]), np.radians(np.asarray(self.RxPolAngles)[rxpol])
        ft = np.stack([amp_t[..., None] * np.cos(zt), amp_t[..., None] * np.sin(zt)], axis=-1)     # [n, m, s, 2]
        fr = np.stack([amp_r[..., None] * np.cos(zr), amp_r[..., None] * np.sin(zr)], axis=-1)     # [n, m, u, 2]
        sk = math.sqrt(1.0 / r.kappa)
        e = np.exp(1j * r.phases)
        xp = np.stack([np.stack([e[..., 0], sk * e[..., 1]], -1), np.stack([sk * e[..., 2], e[..., 3]], -1)], -2)   # [n, m, 2, 2]
        a_tx = np.exp(2j * np.pi * np.einsum("sd,nmd->nms", txp, r_tx))
        a_rx = np.exp(2j * np.pi * np.einsum("ud,nmd->nmu", rxp, r_rx))
        core = np.einsum("nmui,nmij,nmsj->nmsu", fr, xp, ft)
        p_nl = r.power[1:] if r.los else r.power
        base = np.einsum("nmsu,nms,nmu->nmsu", core, a_tx, a_rx) * np.sqrt(p_nl / RAY_OFFSETS.size)[:, None, None, None]
        st = SimpleNamespace(base=np.ascontiguousarray(base), rate=2.0 * np.pi * self.MaximumDopplerShift * (r_rx @ vhat), los=None)
        if r.los:
            row = r.table[0]
            rt, rr = _unit(row[4], row[2]), _unit(row[5], row[3])
            at_, ar_ = _pattern(np.array(row[4]), np.array(row[2]), self.TxElement), _pattern(np.array(row[5]), np.array(row[3]), self.RxElement)
            ft0 = np.stack([at_ * np.cos(zt), at_ * np.sin(zt)], -1)    # [s, 2]
            fr0 = np.stack([ar_ * np.cos(zr), ar_ * np.sin(zr)], -1)    # [u, 2]
            los_core = fr0[None, :, 0] * ft0[:, None, 0] - fr0[None, :, 1] * ft0[:, None, 1]          # [s, u]
            st.los = math.sqrt(r.power[0]) * los_core * np.exp(2j * np.pi * (txp @ rt))[:, None] * np.exp(2j * np.pi * (rxp @ rr))[None, :]
            st.los_rate = 2.0 * np.pi * self.MaximumDopplerShift * float(rr @ vhat)
        self._st = _STATIC_CACHE[key] = st
        return st

    def path_gains(self, t_snap: float) -> np.ndarray:
        """H[n, s, u] at channel time t_snap (TR 38.901 eq. 7.5-22; LOS term 7.5-29 folded into path 0)."""
        st = self._static()
        dop = np.exp(1j * st.rate * t_snap)                               # [n, m]
        h = np.einsum("nmsu,nm->nsu", st.base, dop)
        if st.los is not None:
            h[0] += st.los * np.exp(1j * st.los_rate * t_snap)
        return h

    def filter_taps(self):
        key = (self.DelayProfile, self.DelaySpread, self.SampleRate)
        if key in _TAPS_CACHE:
            return _TAPS_CACHE[key]
        d = self.path_delays() * self.SampleRate
        shift = np.floor(d).astype(np.int32)
        x = np.arange(FILTER_TAPS, dtype=np.float64)[None, :] - FILTER_DELAY - (d - shift)[:, None]
        g = np.sinc(x) * np.where(np.abs(x) < FILTER_TAPS / 2, 0.5 + 0.5 * np.cos(np.pi * x / (FILTER_TAPS / 2)), 0.0)
        g = np.ascontiguousarray(g)
        g.setflags(write=False); shift.setflags(write=False)              # shared between channels: read-only
        _TAPS_CACHE[key] = (g, shift)
        return _TAPS_CACHE[key]

    def __call__(self, waveform, *, ctx=None):
        return applyCDL(self, waveform, ctx=ctx)

    # ---- device-resident form (round 4): the time-independent per-ray terms live on the device, the sample-and-hold path gains of any number
    # of gain blocks are formed there (isac_cdl_path_gains_dev), and applyCDLBatch applies many (channel, waveform) pairs in one call
    def _device_static(self, ctx):
        """(d_base [n][m][s][u], d_rate [n][m], d_los [s][u] | None, los_rate) on `ctx`'s device; cached per context and channel configuration."""
        st = self._static()
        # the device copies live in the CONTEXT's own registry (ctx.device_cache, emptied by Context.close()): a cache on the channel side whose values are
        # DeviceArrays would keep the context alive through d.ctx and never release anything
        key = ("cdl_static", id(st))
        ent = ctx.device_cache.get(key)
        if ent is None or ent[0] is not st:
            d_base = ctx.to_device(np.ascontiguousarray(st.base).reshape(-1))
            d_rate = ctx.to_device(np.ascontiguousarray(st.rate, dtype=np.float64).reshape(-1))
            d_los = ctx.to_device(np.ascontiguousarray(st.los).reshape(-1)) if st.los is not None else None
            ent = ctx.device_cache[key] = (st, (d_base, d_rate, d_los, float(getattr(st, "los_rate", 0.0))))
        return ent[1]

    def block_plan(self, T: int):
        """Gain blocks that the next T samples touch: (snapshot times, first output sample of each block) -- plain Python scalars (this runs once
        per (UE, slot) job in front of every batched apply)."""
        rate = 2.0 * self.SampleDensity * self.MaximumDopplerShift
        if not rate > 0.0:                                                # static channel (MaximumDopplerShift = 0, valid for nrCDLChannel): one block at the current time
            return [self.time], [0]
        b0 = math.floor(self.time * rate + 1e-9)
        b1 = math.floor((self.time + (T - 1) / self.SampleRate) * rate + 1e-9)
        # first output sample of each block: smallest t with floor((time + t/fs) rate + 1e-9) >= b
        starts = [0] + [max(0, math.ceil(((b - 1e-9) / rate - self.time) * self.SampleRate - 1e-6)) for b in range(b0 + 1, b1 + 1)]
        return [b / rate for b in range(b0, b1 + 1)], starts

    def path_gains_device(self, t_snaps, ctx, out=None):
        """H [b][n][s][u] (u fastest) on the device for the channel times `t_snaps` -- the device evaluation of path_gains()."""
        d_base, d_rate, d_los, los_rate = self._device_static(ctx)
        st = self._static()
        n_paths, n_rays, nt, nr = st.base.shape
        t = np.ascontiguousarray(np.asarray(t_snaps, dtype=np.float64).reshape(-1))
        d_h = out if out is not None else ctx.empty((t.size * n_paths * nt * nr,))
        ctx.check(ctx.lib.isac_cdl_path_gains_dev(ctx.handle, C.c_void_p(d_base.ptr), C.c_void_p(d_rate.ptr), C.c_int32(n_paths), C.c_int32(n_rays), C.c_int32(nt),
                                                  C.c_int32(nr), C.c_void_p(d_los.ptr if d_los is not None else 0), C.c_double(los_rate),
                                                  t.ctypes.data_as(C.c_void_p), C.c_int32(t.size), C.c_void_p(d_h.ptr)))
        return d_h


    def _fr_tables(self, ctx, k_sub, n_sc, scs_hz):
        """(d_tau [n_paths], d_freq [n_re], n_re) on `ctx`'s device for the 1-based subcarriers k_sub; cached per context and subcarrier set."""
        st = self._static()
        key = ("cdl_fr", id(st), hashlib.sha1(np.ascontiguousarray(k_sub, dtype=np.int64).tobytes()).digest(), int(n_sc), float(scs_hz))   # keyed on the CONTENT of the subcarrier set
        cache = ctx.device_cache                                           # (per context, released by Context.close())
        if key not in cache or cache[key][0] is not st:
            f = ((np.asarray(k_sub, dtype=np.float64) - 1.0) - n_sc / 2.0) * float(scs_hz)
            cache[key] = (st, ctx.to_device(np.ascontiguousarray(self.path_delays(), dtype=np.float64)), ctx.to_device(np.ascontiguousarray(f)), f.size)
        return cache[key][1:]

    def snap_time(self, t=None):
        """The sample-and-hold gain-block time that holds channel time t (default: now) -- what the apply uses for a sample at t."""
        if t is None:
            t = self.time
        rate = 2.0 * self.SampleDensity * self.MaximumDopplerShift
        return float(t) if not rate > 0.0 else math.floor(float(t) * rate + 1e-9) / rate

    def freq_response_device(self, k_sub, n_sc, scs_hz, ports, ctx, *, t=None, out=None, gains=None):
        """Perfect channel estimate Hf [n_re x Nr x ports] ON THE DEVICE at the 1-based subcarriers `k_sub` of an n_sc-subcarrier grid, for the channel time
        `t` (default: the current one, snapped to its sample-and-hold gain block as the apply does): path gains of that block from isac_cdl_path_gains_dev,
        then Hf[i, u, p] = sum_n h[n, p, u] exp(-2 pi j f_i tau_n) (isac_cdl_freq_response_dev) -- nothing is evaluated on the host.  What the CSI report of a
        CSI-RS occasion works on (uePhy.m:901-908); the estimator itself is out of scope."""
        st = self._static()
        n_paths, _, nt, nr = st.base.shape
        t_snap = self.snap_time(t)                                       # an explicit t is snapped to its gain block too (as csiEstimateBatch and the apply do)
        d_h = self.path_gains_device([t_snap], ctx, out=gains)
        d_tau, d_f, n_re = self._fr_tables(ctx, k_sub, n_sc, scs_hz)
        if out is None:
            out = ctx.empty((n_re, nr, int(ports)))
        ctx.check(ctx.lib.isac_cdl_freq_response_dev(ctx.handle, C.c_void_p(d_h.ptr), C.c_int32(n_paths), C.c_int32(nt), C.c_int32(nr), C.c_int32(int(ports)),
                                                     C.c_void_p(d_tau.ptr), C.c_void_p(d_f.ptr), C.c_int64(n_re), C.c_void_p(out.ptr)))
        return out


def csiEstimateBatch(channels, k_sub, n_sc, scs_hz, ports, *, ctx, times=None, outs=None):
    """The perfect CSI-RS channel estimates Hf_j [n_re x Nr x ports] of MANY UEs of one delay profile at their own channel times in ONE library call
    (isac_cdl_csi_estimate_batch_dev): per UE the sample-and-hold path gains of `times[j]` (default: its current channel time) and their response at the
    1-based subcarriers k_sub, formed on the device without the gains leaving the CU.  A cell's CSI-RS occasion (uePhy.m:901-908 works on these estimates)
    is one call per delay profile.  Returns the list of DeviceArrays (`outs` if given)."""
    channels = list(channels)
    if not channels:
        return []
    ch0 = channels[0]
    st0 = ch0._static()
    n_paths, n_rays, nt, nr = st0.base.shape
    tau0 = ch0.path_delays()
    for ch in channels[1:]:
        if ch._static().base.shape != st0.base.shape or not np.array_equal(ch.path_delays(), tau0):
            raise ValueError("csiEstimateBatch: all channels must share the delay profile, delay spread and antenna counts")
    d_tau, d_f, n_re = ch0._fr_tables(ctx, k_sub, n_sc, scs_hz)
    n = len(channels)
    if outs is None:
        outs = [ctx.empty((n_re, nr, int(ports))) for _ in range(n)]
    statics = [ch._device_static(ctx) for ch in channels]
    vp = C.c_void_p * n
    base = vp(*[st[0].ptr for st in statics])
    rate = vp(*[st[1].ptr for st in statics])
    los = vp(*[(st[2].ptr if st[2] is not None else None) for st in statics])
    los_rate = (C.c_double * n)(*[st[3] for st in statics])
    t = (C.c_double * n)(*[ch.snap_time(None if times is None else times[j]) for j, ch in enumerate(channels)])
    hf = vp(*[o.ptr for o in outs])
    ctx.check(ctx.lib.isac_cdl_csi_estimate_batch_dev(ctx.handle, C.c_int32(n), base, rate, los, los_rate, t, C.c_int32(n_paths), C.c_int32(n_rays), C.c_int32(nt), C.c_int32(nr),
                                                      C.c_int32(int(ports)), C.c_void_p(d_tau.ptr), C.c_void_p(d_f.ptr), C.c_int64(n_re), hf))
    return outs


def applyCDLBatch(channels, waveforms, *, ctx=None, outs=None, gains=None):
    """rxWaveform_i = channel_i(waveform_i) for many (UE, slot) pairs in ONE library call (isac_cdl_apply_batch_dev): uePhy.m:729-731 inside the
    per-UE loop of a cell, or the slots of a frame.  All channels must share antenna counts, sample rate, delay profile length and filter taps
    (the reference builds every UE's channel from the same cdl.m:57-64 template); `waveforms` are DeviceArrays [T x Nt] -- the same array may appear
    several times (the UEs of one cell receive one downlink waveform), and so may a channel (consecutive slots of one UE: its time advances from
    job to job in list order).  Path gains are formed on the device.  Advances every channel's time by T samples per appearance.
    `outs` / `gains`: caller-owned output arrays and a path-gain scratch DeviceArray (>= total gain blocks x n_paths x Nt x Nr elements) that a
    frame loop reuses from slot to slot -- without them every call allocates and (when the previous outputs are dropped) frees, and a free
    synchronises the stream.  Returns the list of output DeviceArrays [T x Nr]."""
    channels, waveforms = list(channels), list(waveforms)
    if not channels or len(channels) != len(waveforms):
        raise ValueError("applyCDLBatch: one waveform per channel")
    ctx = ctx or waveforms[0].ctx
    c0 = channels[0]
    T, nt = waveforms[0].shape
    nr = c0.n_rx
    g, shift = c0.filter_taps()
    n_paths = g.shape[0]
    checked = set()                                                       # (a channel that appears many times -- the slots of a frame -- is validated once)
    for ch, w in zip(channels, waveforms):
        if not isinstance(w, L.DeviceArray) or tuple(w.shape) != (T, nt):
            raise ValueError("applyCDLBatch: waveforms must be DeviceArrays of one shape [T x Nt]")
        if id(ch) in checked:
            continue
        checked.add(id(ch))
        if ch.n_tx != nt or ch.n_rx != nr:
            raise ValueError("applyCDLBatch: channels must share the antenna counts")
        g2, s2 = ch.filter_taps()
        if g2 is not g and (g2.shape != g.shape or not np.array_equal(s2, shift) or not np.array_equal(g2, g)):
            raise ValueError("applyCDLBatch: channels must share path delays / filter taps (one delay profile and delay spread per batch)")
    scale = 1.0 / math.sqrt(nr) if c0.NormalizeChannelOutputs else 1.0
    jobs = (L.CdlJob * len(channels))()
    keep = []
    outs = list(outs) if outs is not None else [ctx.empty((T, nr)) for _ in channels]
    per_block = n_paths * nt * nr
    # (channel time advances job by job: a channel that appears several times in the list -- consecutive slots of one UE in one call -- sees
    # consecutive stretches of channel time, exactly as consecutive single calls would give it)
    plans = []
    for ch in channels:
        plans.append(ch.block_plan(T))
        ch.time += T / ch.SampleRate
    # path gains: ONE device evaluation per distinct channel configuration (the reference gives every UE of a delay profile the same seed,
    # cdl.m:57-64: their channels differ in channel time only) over the concatenated snapshot times of its jobs
    groups: dict = {}
    for i, ch in enumerate(channels):
        groups.setdefault(id(ch._static()), []).append(i)
    n_blk_total = sum(len(p[0]) for p in plans)
    if gains is not None and math.prod(gains.shape) < n_blk_total * per_block:
        raise ValueError("applyCDLBatch: `gains` scratch too small for this batch's gain blocks")
    d_h_all = gains if gains is not None else ctx.empty((n_blk_total * per_block,))
    off = 0
    for idx in groups.values():
        t_cat = np.array([t for i in idx for t in plans[i][0]], dtype=np.float64)
        d_grp = L.DeviceArray(ctx, d_h_all.ptr + 16 * off * per_block, (t_cat.size * per_block,), np.complex128, owner=False)   # view into d_h_all
        channels[idx[0]].path_gains_device(t_cat, ctx, out=d_grp)
        for i in idx:
            st = np.array(plans[i][1], dtype=np.int64)
            keep.append(st)
            jobs[i].d_x, jobs[i].d_y, jobs[i].d_H = waveforms[i].ptr, outs[i].ptr, d_h_all.ptr + 16 * off * per_block
            jobs[i].block_start = st.ctypes.data_as(C.c_void_p).value
            jobs[i].n_blocks = st.size
            off += st.size
    ctx.check(ctx.lib.isac_cdl_apply_batch_dev(ctx.handle, jobs, C.c_int32(len(channels)), C.c_int64(T), C.c_int32(nt), C.c_int32(nr), C.c_int32(n_paths),
                                               g.ctypes.data_as(C.c_void_p), C.c_int32(FILTER_TAPS), shift.ctypes.data_as(C.c_void_p), C.c_double(scale)))
    if gains is None:
        for o in outs:
            o._cdl_keep = d_h_all                   # the gains stay alive until the outputs are dropped (the launches are asynchronous)
    return outs


def applyCDL(channel: CDLChannel, waveform, *, ctx=None):
    """rxWaveform = channel(waveform)   (uePhy.m:731 / gNBPhy.m:840).  waveform [T x Nt] -> [T x Nr]; numpy in ->
    numpy out, DeviceArray in -> DeviceArray out.  Advances the channel time by T / SampleRate."""
    dev = isinstance(waveform, L.DeviceArray)
    ctx = ctx or (waveform.ctx if dev else L.default_context())
    T, nt = (waveform.shape if dev else np.shape(waveform))
    if nt != int(np.prod(channel.TransmitAntennaArraySize)):
        raise ValueError("waveform columns differ from the transmit array size")
    nr = int(np.prod(channel.ReceiveAntennaArraySize))
    t_snap, st_list = channel.block_plan(T)
    starts = np.array(st_list, dtype=np.int64)
    h = np.ascontiguousarray(np.stack([channel.path_gains(t) for t in t_snap]))               # [b, n, s, u]
    g, shift = channel.filter_taps()
    scale = 1.0 / math.sqrt(nr) if channel.NormalizeChannelOutputs else 1.0
    d_x = waveform if dev else ctx.to_device(L.as_c128_f(waveform))
    d_y = ctx.empty((T, nr))
    ctx.check(ctx.lib.isac_cdl_apply_dev(ctx.handle, C.c_void_p(d_x.ptr), C.c_int64(T), C.c_int32(nt), C.c_int32(nr), C.c_int32(h.shape[1]),
                                         h.ctypes.data_as(C.c_void_p), C.c_int32(len(t_snap)), starts.ctypes.data_as(C.c_void_p),
                                         g.ctypes.data_as(C.c_void_p), C.c_int32(FILTER_TAPS), shift.ctypes.data_as(C.c_void_p),
                                         C.c_double(scale), C.c_void_p(d_y.ptr)))
    channel.time += T / channel.SampleRate
    return d_y if dev else d_y.numpy()
