"""communication.phyLayer.prgPrecode (+communication/+phyLayer/prgPrecode.m:53-144; gNBPhy.m:822-827 calls it for PDSCH and its DM-RS)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ... import _lib as L


def prgPrecodeGrid(layers, F, nstartgrid=0, *, ctx=None, out=None):
    """Dense, device-resident form: ``layers`` [K x L x nu] (DeviceArray or array; zero where a layer carries nothing) -> antenna grid [K x L x P] =
    layers(k, l, :) * F(:, :, prg(k)) with the PRG of every RE as getPRGSet assigns it (prgPrecode.m:93-99).  F [nu x P x NPRG] (host).  The result is the
    txSlotGrid the OFDM modulator takes next (isac_prg_precode_dev)."""
    dev = isinstance(layers, L.DeviceArray)
    ctx = ctx or (layers.ctx if dev else L.default_context())
    F = np.asarray(F, dtype=np.complex128)
    if F.ndim == 2:
        F = F[:, :, None]
    nu, P, nprg = F.shape
    d_l = layers if dev else ctx.to_device(L.as_c128_f(np.asarray(layers).reshape(np.shape(layers)[0], np.shape(layers)[1], -1)))
    K, Ls = d_l.shape[0], d_l.shape[1]
    if (d_l.shape[2] if len(d_l.shape) > 2 else 1) != nu:
        raise ValueError("layers must be [K x L x nu] with nu = size(F, 1)")
    if out is None:
        out = ctx.empty((K, Ls, P))
    Ff = np.asfortranarray(F)
    ctx.check(ctx.lib.isac_prg_precode_dev(ctx.handle, C.c_void_p(d_l.ptr), C.c_int32(K), C.c_int32(Ls), C.c_int32(nu), Ff.ctypes.data_as(C.c_void_p), C.c_int32(P),
                                           C.c_int32(nprg), C.c_int32(int(nstartgrid)), C.c_void_p(out.ptr)))
    return out if dev else out.numpy()


def prgPrecode(siz, nstartgrid, portsym, portind, F, *, ctx=None):
    """[antsym, antind] = prgPrecode(siz, nstartgrid, portsym, portind, F): the reference's index-list signature (1-based linear indices into the
    [K x L x nu] port grid).  The symbols are scattered into the dense layer grid on the host, precoded on the device, and read back at the RE positions
    of the port indices on every antenna plane (nrExtractResources, prgPrecode.m:141)."""
    F = np.asarray(F, dtype=np.complex128)
    if F.ndim == 2:
        F = F[:, :, None]
    nu, P, _ = F.shape
    K, Ls = int(siz[0]), int(siz[1])
    ind = np.asarray(portind, dtype=np.int64)
    ind = ind.reshape(-1, nu) if ind.ndim > 1 else ind.reshape(-1, 1)
    sym = np.asarray(portsym, dtype=np.complex128).reshape(ind.shape)
    layers = np.zeros(K * Ls * nu, dtype=np.complex128)
    layers[ind.reshape(-1) - 1] = sym.reshape(-1)
    grid = prgPrecodeGrid(layers.reshape((K, Ls, nu), order="F"), F, nstartgrid, ctx=ctx)
    re = (ind[:, 0] - 1) % (K * Ls)
    antind = re[:, None] + (K * Ls) * np.arange(P)[None, :] + 1
    return np.asarray(grid).reshape(K * Ls, P, order="F")[re, :], antind
