"""TEST INFRASTRUCTURE (oracle): CPU restatement of communication.phyLayer.prgPrecode -- the product never imports this.

+communication/+phyLayer/prgPrecode.m:53-144 (getPRGSet :93-99, hPrecode :102-143), loop for loop: per PRG a port grid holding that PRG's symbols is
beamformed with F(:,:,prg) and added to the antenna grid; nrExtractResources then reads the antenna grid back at the RE positions of the port
indices.  1-based linear indices as MATLAB uses them.  Parity unpinned like the rest of the oracle (no MATLAB in the image)."""
from __future__ import annotations

import numpy as np


def get_prg_set(nrb: int, nstartgrid: int, nprg: int) -> np.ndarray:
    """prgPrecode.m:93-99: 1-based PRG number of every CRB of the carrier."""
    pd_bwp = -(-(nrb + nstartgrid) // nprg)
    prgset = np.tile(np.arange(1, nprg + 1), (pd_bwp, 1)).reshape(-1, order="F")      # repmat(1:NPRG, [Pd_BWP 1]) read column-major
    return prgset[nstartgrid + np.arange(nrb)]


def prg_precode(siz, nstartgrid, portsym, portind, F):
    """[antsym, antind] = prgPrecode(siz, nstartgrid, portsym, portind, F) (:53-90).  siz = (K, L[, P]); portsym / portind [nRE x nu] (1-based linear indices
    into the [K x L x nu] port grid); F [nu x P x NPRG].  Returns antsym [nRE x P], antind [nRE x P] (1-based, into [K x L x P])."""
    F = np.asarray(F, dtype=np.complex128)
    if F.ndim == 2:
        F = F[:, :, None]
    nu, P, nprg = F.shape
    K, L = int(siz[0]), int(siz[1])
    nrb = K // 12
    prgset = get_prg_set(nrb, int(nstartgrid), nprg)                                  # :63
    portind = np.asarray(portind, dtype=np.int64).reshape(-1, nu) if np.ndim(portind) > 1 else np.asarray(portind, dtype=np.int64).reshape(-1, 1)
    portsym = np.asarray(portsym, dtype=np.complex128).reshape(portind.shape)
    lin0 = portind - 1
    resubs = lin0 % K                                                                  # :71-72 (0-based RE subscript)
    prgsubs = prgset[resubs // 12]                                                     # :81-84
    antgrid = np.zeros((K * L, P), dtype=np.complex128)                                # :113
    for prg in range(1, nprg + 1):                                                     # :116-140
        this = prgsubs == prg
        if not this.any():
            continue
        portgrid = np.zeros((K * L * nu,), dtype=np.complex128)                       # :128 ([K L x nu] read column-major)
        portgrid[lin0[this]] = portsym[this]                                           # :131
        antgrid = antgrid + portgrid.reshape(nu, K * L).T @ F[:, :, prg - 1]           # :134-137
    re = lin0[:, 0] % (K * L)                                                          # nrExtractResources: the RE positions of the (first) port plane on every antenna plane
    antind = re[:, None] + (K * L) * np.arange(P)[None, :] + 1
    antsym = antgrid[re, :]
    return antsym, antind
