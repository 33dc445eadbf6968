"""Oracle restatement of the SINR -> CQI arithmetic of config 5. TEST INFRASTRUCTURE ONLY.

* ``precoded_sinr``  -- +communication/+phyLayer/precodedSINR.m:11-17 (LMMSE SINR of a precoded unit-power signal)
* ``get_cqi``        -- +communication/+phyLayer/cqiSelect.m:697-722 (largest table entry <= measured SINR)
* tables             -- +communication/setupSINRtoCQIMappingTable.m:7-11
The exhaustive Type-I codebook search around them (dlPMISelect.m, 1 800 lines of MathWorks helper code) is out of
scope; the batched per-RE SINR evaluation + averaging + lookup is the data-parallel part.
"""
from __future__ import annotations

import numpy as np

DOWNLINK_SINR90PC = np.array([-3.46, 1.54, 6.54, 11.05, 13.54, 16.04, 17.54, 20.04, 22.04, 24.43, 26.93, 27.43, 29.43, 32.43, 35.43])
UPLINK_SINR90PC = np.array([-5.46, -0.46, 4.54, 9.05, 11.54, 14.04, 15.54, 18.04, 20.04, 22.43, 24.93, 25.43, 27.43, 30.43, 33.43])


def precoded_sinr(h, sigma, w):
    """precodedSINR.m:11-17:  noise = sigma^2 I;  den = noise / (W' H' H W + noise);  sinr = real(sum(1./diag(den) - 1))."""
    h = np.asarray(h, dtype=np.complex128)
    w = np.asarray(w, dtype=np.complex128)
    n_l = w.shape[1]
    noise = sigma ** 2 * np.eye(n_l)
    g = h @ w
    den = noise @ np.linalg.inv(g.conj().T @ g + noise)      # A/B == A*inv(B)
    return float(np.real(np.sum(1.0 / np.diag(den) - 1.0)))


def precoded_sinr_batch(h, sigma, w):
    """h [nRE x Nr x P] -> sinr [nRE]."""
    return np.array([precoded_sinr(h[i], sigma, w) for i in range(h.shape[0])])


def get_cqi(linear_sinr, sinr_table):
    """cqiSelect.m:697-722."""
    if np.all(np.isnan(linear_sinr)):
        return float("nan")
    with np.errstate(divide="ignore", invalid="ignore"):
        s_db = 10.0 * np.log10(linear_sinr)
    idx = np.flatnonzero(np.asarray(sinr_table) <= s_db)
    return 0 if idx.size == 0 else int(idx[-1]) + 1
