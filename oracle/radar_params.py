"""Oracle restatement of ``sensing.radarParams`` (+sensing/radarParams.m:1-146).

Host-side scalar preparation: link budget, resolutions, steering vectors,
scan configuration and CFAR zone.  TEST INFRASTRUCTURE ONLY (see package doc).
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np

from .matlab_compat import LIGHTSPEED, BOLTZMANN, db2pow, pow2db, sind, cosd


def nr_ofdm_info(nrb: int, scs_khz: float):
    """Subset of 5G Toolbox ``nrOFDMInfo(NRB, SCS)`` (gNBPhy.m:772).

    Nfft: smallest power of two with >= 85 % occupancy rule satisfied
    (NRB*12/Nfft <= 0.85), minimum 128; SampleRate = Nfft*SCS.
    """
    k = 12 * nrb
    nfft = max(128, 2 ** math.ceil(math.log2(k / 0.85)))
    return SimpleNamespace(Nfft=nfft, SampleRate=float(nfft * scs_khz * 1e3),
                           SymbolsPerSlot=14, SlotsPerSubframe=int(scs_khz // 15))


def default_cell_params(n_ants=16, target_pos=((100.0, 0.0, 1.5),), velocity=(0.0,),
                        rcs=None, num_frames=1):
    """The flat per-cell struct of ``assignCellSimulationParameters.m:27-101`` restricted
    to the fields ``radarParams`` reads, filled with the defaults of
    ``scenarios/openStreetMapCity.m:53-77`` and ``+parameters/+baseStation``.
    """
    target_pos = np.atleast_2d(np.asarray(target_pos, dtype=np.float64))
    q = target_pos.shape[0]
    p = SimpleNamespace()
    p.numTargets = q
    p.targetPosition = target_pos                        # [Q x 3]
    p.gNBPosition = np.array([0.0, 0.0, 30.0])           # openStreetMapCity.m:56
    p.tddPattern = "DDDSU"                               # :65
    p.numDLSlots = 3
    p.numSlots = 20 * num_frames                         # gNBParameters numSlotsFrame @30 kHz
    p.gNBTxAnts = int(n_ants)
    p.dlCarrierFreq = 3.5e9                              # :60
    p.gNBNoiseFigure = 6.0                               # gNBParameters.m:35
    p.gNBTemperature = 290.0                             # gNBParameters.m:40
    p.gNBTxPower = 46.0                                  # :73
    p.gNBRxGain = 25.5                                   # :74
    p.rcs = np.ones(q) if rcs is None else np.asarray(rcs, dtype=np.float64)
    p.velocity = np.asarray(velocity, dtype=np.float64).reshape(q)
    p.gNBSenAntenna = SimpleNamespace(kind="ula", numElements=int(n_ants), d=0.5,
                                      nV=int(n_ants) // 2, p=2)
    p.Pfa = 1e-9                                         # radar.m:15
    p.detectionArea = np.array([[50.0, 500.0], [-50.0, 50.0]])   # radar.m:10
    return p


def radar_params(cell, carrier_info, wave_info):
    """radarParams.m:1-146.  ``carrier_info``: NRBsDL, SubcarrierSpacing [kHz];
    ``wave_info``: SampleRate, Nfft, SymbolsPerSlot."""
    rp = SimpleNamespace()
    q = int(cell.numTargets)
    # :12-14  relative coordinates -> cart2sph -> degrees
    coords = np.asarray(cell.targetPosition, dtype=np.float64).T - np.asarray(cell.gNBPosition)[:, None]
    x, y, z = coords
    azi_rad = np.arctan2(y, x)
    ele_rad = np.arctan2(z, np.hypot(x, y))
    rng = np.sqrt(x * x + y * y + z * z)
    azi = np.rad2deg(azi_rad)
    ele = np.rad2deg(ele_rad)

    # :18-24
    dl_ratio = cell.numDLSlots / len(cell.tddPattern)
    n_dl_slots = dl_ratio * cell.numSlots
    n_sc = carrier_info.NRBsDL * 12
    n_sym = n_dl_slots * wave_info.SymbolsPerSlot
    uf = 1
    ut = 1
    n_tx = int(cell.gNBTxAnts)

    # :27-35
    c = LIGHTSPEED
    fc = float(cell.dlCarrierFreq)
    scs = carrier_info.SubcarrierSpacing * 1e3
    lam = c / fc
    fs = float(wave_info.SampleRate)
    ts = 1.0 / fs
    t_ofdm = 1.0 / scs
    t_cp = ts * math.ceil(n_sc / 8)
    t_sri = t_ofdm + t_cp

    # :38-43
    nf = float(db2pow(cell.gNBNoiseFigure))
    teq = cell.gNBTemperature + 290.0 * (nf - 1.0)
    n0 = fs * BOLTZMANN * teq
    pt = float(db2pow(cell.gNBTxPower - 30.0)) * math.sqrt(wave_info.Nfft ** 2 / (carrier_info.NRBsDL * 12 * n_tx))
    ar = float(db2pow(cell.gNBRxGain))
    at = ar

    # :46-51
    rcs = np.asarray(cell.rcs, dtype=np.float64).reshape(q)
    r = rng.reshape(q)
    v = np.asarray(cell.velocity, dtype=np.float64).reshape(q)
    pr = pt * at * ar * (lam ** 2 * rcs) / ((4.0 * np.pi) ** 3 * r ** 4)
    snr = pr / n0
    snr_db = pow2db(snr)

    # :54-65
    rp.fc, rp.fs, rp.Tsri, rp.N0 = fc, fs, t_sri, n0
    rp.nTxAnts, rp.nTargets = n_tx, q
    rp.range, rp.velocity = r, v
    rp.largeScaleFading = np.sqrt(pr / pt)
    rp.snrdB = snr_db
    rp.txPower = cell.gNBTxPower
    rp.Pfa = cell.Pfa

    # :69-78
    n_ifft = 2 ** math.ceil(math.log2(n_sc / uf))
    rp.nIFFT = n_ifft
    rp.rRes = c / (2 * (scs * uf) * n_ifft)
    rp.rMax = c / (2 * (scs * uf))
    n_fft = 2 ** math.ceil(math.log2(n_sym / ut))
    rp.nFFT = n_fft
    rp.vRes = lam / (2 * (t_sri * ut) * n_fft)
    rp.vMax = lam / (2 * (t_sri * ut))

    # :81-118 steering vectors (note: spacing in wavelengths divided by lambda in metres, :95,109)
    arr = cell.gNBSenAntenna
    if getattr(arr, "kind", "ula") == "upa":
        ant_x = np.arange(arr.nV) * arr.dV                 # [1 x nX]
        ant_y = (np.arange(arr.nH) * arr.dH)[:, None]      # [nY x 1]
        cols = []
        for t in range(q):
            m = np.exp(2j * np.pi * sind(ele[t]) * (ant_x * cosd(azi[t]) + ant_y * sind(azi[t])) / lam)
            cols.append(m.reshape(n_tx, order="F"))        # reshape(.., nRxAnts, 1) is column-major
        sv = np.stack(cols, axis=1)
    else:
        ant = np.arange(n_tx) * arr.d
        sv = np.stack([np.exp(2j * np.pi * ant * sind(azi[t]) / lam) for t in range(q)], axis=1)
    rp.antennaType = arr
    rp.azimuthScanScale = 360
    rp.elevationScanScale = 180
    rp.azimuthScanGranularity = 1
    rp.elevationScanGranularity = 1
    rp.RxSteeringVec = sv                                   # [nAnts x Q]

    rp.cfarEstZone = np.asarray(cell.detectionArea, dtype=np.float64)   # :129

    # :132-144 ground truth sorted by SNR descending (stable)
    order = np.argsort(-snr_db, kind="stable")
    rp.targetRealPos = [dict(ID=i + 1, Range=r[j], Velocity=v[j], Elevation=ele[j],
                             Azimuth=azi[j], snrdB=snr_db[j]) for i, j in enumerate(order)]
    # extras the rest of the oracle wants (not reference fields)
    rp.azimuth_deg, rp.elevation_deg = azi, ele
    return rp
