"""sensing.radarParams (+sensing/radarParams.m:1-146): host-side link budget, 2D-FFT resolutions,
steering vectors, scan configuration and CFAR zone.  Scalar work that stays on the host and feeds
the kernels."""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np

LIGHTSPEED = 299792458.0        # physconst('Lightspeed')
BOLTZMANN = 1.380649e-23        # physconst('Boltzmann')


def _db2pow(x):
    return 10.0 ** (float(x) / 10.0)


def sind(x):
    """Degree sine with exact quadrant values and bit-exact mirror symmetry sind(180-p) == sind(p)."""
    x = np.fmod(np.asarray(x, dtype=np.float64), 360.0)
    x = np.where(x > 180.0, x - 360.0, x)
    x = np.where(x < -180.0, x + 360.0, x)
    x = np.where(x > 90.0, 180.0 - x, x)
    x = np.where(x < -90.0, -180.0 - x, x)
    ax = np.abs(x)
    k = math.pi / 180.0
    return np.where(ax <= 45.0, np.sin(x * k), np.sign(x) * np.cos((90.0 - ax) * k))


def cosd(x):
    return sind(90.0 - np.fmod(np.abs(np.asarray(x, dtype=np.float64)), 360.0))


def radarParams(cellSimuParams, carrierInfo, waveInfo):
    """radarParams = sensing.radarParams(cellSimuParams, carrierInfo, waveInfo)  (radarParams.m:1)."""
    p = SimpleNamespace()
    c = cellSimuParams
    nTargets = int(c.numTargets)                                                    # :11
    coords = np.asarray(c.targetPosition, dtype=np.float64).T - np.asarray(c.gNBPosition, dtype=np.float64)[:, None]   # :12
    x, y, z = coords
    azi = np.rad2deg(np.arctan2(y, x))                                              # :13-14 cart2sph
    ele = np.rad2deg(np.arctan2(z, np.hypot(x, y)))
    rng = np.sqrt(x * x + y * y + z * z)

    dlRatio = c.numDLSlots / len(c.tddPattern)                                      # :18
    nDLSlots = dlRatio * c.numSlots                                                 # :19
    nSc = carrierInfo.NRBsDL * 12                                                   # :20
    nSym = nDLSlots * waveInfo.SymbolsPerSlot                                       # :21
    uf = ut = 1                                                                     # :22-23
    nTxAnts = int(c.gNBTxAnts)                                                      # :24

    fc = float(c.dlCarrierFreq)                                                     # :28
    scs = carrierInfo.SubcarrierSpacing * 1e3                                       # :29
    lam = LIGHTSPEED / fc                                                           # :30
    fs = float(waveInfo.SampleRate)                                                 # :31
    Ts = 1.0 / fs
    Tofdm = 1.0 / scs
    Tcp = Ts * math.ceil(nSc / 8)                                                   # :34
    Tsri = Tofdm + Tcp                                                              # :35

    NF = _db2pow(c.gNBNoiseFigure)                                                  # :38
    Teq = c.gNBTemperature + 290.0 * (NF - 1.0)                                     # :39
    N0 = fs * BOLTZMANN * Teq                                                       # :40
    Pt = _db2pow(c.gNBTxPower - 30.0) * math.sqrt(waveInfo.Nfft ** 2 / (carrierInfo.NRBsDL * 12 * nTxAnts))   # :41
    Ar = _db2pow(c.gNBRxGain)                                                       # :42
    At = Ar

    rcs = np.asarray(c.rcs, dtype=np.float64).reshape(nTargets)                     # :46
    r = rng.reshape(nTargets)
    v = np.asarray(c.velocity, dtype=np.float64).reshape(nTargets)
    Pr = Pt * At * Ar * (lam ** 2 * rcs) / ((4.0 * np.pi) ** 3 * r ** 4)            # :49
    snrdB = 10.0 * np.log10(Pr / N0)                                                # :50-51

    p.fc, p.fs, p.Tsri, p.N0 = fc, fs, Tsri, N0                                     # :54-57
    p.nTxAnts, p.nTargets = nTxAnts, nTargets
    p.range, p.velocity = r, v
    p.largeScaleFading = np.sqrt(Pr / Pt)                                           # :62
    p.snrdB, p.txPower, p.Pfa = snrdB, c.gNBTxPower, c.Pfa

    p.nIFFT = 2 ** math.ceil(math.log2(nSc / uf))                                   # :69
    p.rRes = LIGHTSPEED / (2 * (scs * uf) * p.nIFFT)                                # :71
    p.rMax = LIGHTSPEED / (2 * (scs * uf))
    p.nFFT = 2 ** math.ceil(math.log2(nSym / ut))                                   # :75
    p.vRes = lam / (2 * (Tsri * ut) * p.nFFT)                                       # :77
    p.vMax = lam / (2 * (Tsri * ut))

    arr = c.gNBSenAntenna                                                           # :81
    if getattr(arr, "kind", "ula") == "upa":                                        # :84-100
        antX = np.arange(arr.nV) * arr.dV
        antY = (np.arange(arr.nH) * arr.dH)[:, None]
        cols = [np.exp(2j * np.pi * sind(ele[t]) * (antX * cosd(azi[t]) + antY * sind(azi[t])) / lam).reshape(nTxAnts, order="F")
                for t in range(nTargets)]
    else:                                                                           # :102-114  (spacing in wavelengths / lambda in metres, as written)
        ant = np.arange(nTxAnts) * arr.d
        cols = [np.exp(2j * np.pi * ant * sind(azi[t]) / lam) for t in range(nTargets)]
    p.antennaType = arr                                                             # :120
    p.azimuthScanScale, p.elevationScanScale = 360, 180                             # :121-122
    p.azimuthScanGranularity = p.elevationScanGranularity = 1                       # :123-124
    p.RxSteeringVec = np.stack(cols, axis=1) if cols else np.zeros((nTxAnts, 0), complex)   # :125
    p.cfarEstZone = np.asarray(c.detectionArea, dtype=np.float64)                   # :129
    order = np.argsort(-snrdB, kind="stable")                                       # :133
    p.targetRealPos = [dict(ID=i + 1, Range=r[j], Velocity=v[j], Elevation=ele[j], Azimuth=azi[j], snrdB=snrdB[j])
                       for i, j in enumerate(order)]                                # :137-144
    return p
