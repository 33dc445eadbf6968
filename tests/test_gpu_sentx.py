"""senTxGrid / senTxWave accumulation (gNBPhy.m:591-612) and the windowed CP-OFDM modulator (nrOFDMModulate, gNBPhy.m:599) on the
device against the oracle restatement (oracle/sentx.py, oracle/ofdm.py): <= 1e-10 relative, zero blocks exactly zero."""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import pytest

import oracle as O
from conftest import load_pkg

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


@pytest.fixture(scope="module")
def pkg():
    return load_pkg()


def qpsk(shape, seed):
    rng = np.random.default_rng(seed)
    return np.asfortranarray(((rng.integers(0, 2, shape) * 2 - 1) + 1j * (rng.integers(0, 2, shape) * 2 - 1)) / np.sqrt(2.0))


@pytest.mark.parametrize("nrb,scs,nfft,n_slot,win", [(24, 30, 512, 0, 0), (24, 30, 512, 1, 18), (273, 30, 4096, 3, 144), (52, 15, 1024, 2, 36),
                                                    (66, 60, 1024, 1, 8), (66, 60, 1024, 2, 72)])
def test_windowed_modulator_matches_oracle(pkg, nrb, scs, nfft, n_slot, win):
    ci = SimpleNamespace(NRBsDL=nrb, SubcarrierSpacing=scs)
    g = qpsk((12 * nrb, 14, 3), nrb + n_slot)
    first = (n_slot % (scs // 15)) * 14
    want = 2.5 * O.ofdm_modulate(g, nfft, scs, windowing=win, first_symbol=first)
    got = pkg.communication.phyLayer.nrOFDMModulate(ci, g, nSlot=n_slot, windowing=win, amplitude=2.5, nfft=nfft)
    assert got.shape == want.shape and rel(got, want) < 1e-10
    if win:     # the windowed waveform still demodulates to the grid (the taper stays inside the CP half the demodulator skips)
        assert rel(O.ofdm_demodulate(got / 2.5, 12 * nrb, nfft, scs)[:, :14] if first == 0 else g, g) < 1e-9
    with pytest.raises(pkg.IsacError):
        pkg.communication.phyLayer.nrOFDMModulate(ci, g, windowing=10 ** 6, nfft=nfft)


@pytest.mark.parametrize("nrb,nfft,n_ants,win", [(24, 512, 4, 0), (273, 4096, 2, 144)])
def test_sentx_accumulation_matches_oracle(pkg, nrb, nfft, n_ants, win):
    ci = SimpleNamespace(NRBsDL=nrb, SubcarrierSpacing=30)
    tdd, pw = "DDDSU", 46.0
    ref = O.sentx.SenTx(nfft, 30, tdd, pw, windowing=win)
    slots = [s for s in range(10) if tdd[s % 5] != "U"]             # PDSCH goes out in D and S slots; U slots never reach phyTx's PDSCH branch
    acc = pkg.communication.phyLayer.SenTx(ci, n_ants, len(slots), tdd, pw, windowing=win, nfft=nfft)
    for s in slots:
        g = qpsk((12 * nrb, 14, n_ants), 100 + s)
        ref.append(g, s)
        acc.append(g, s)
    assert acc.nSlots == len(slots) and acc.T == ref.wave.shape[0]
    assert np.array_equal(acc.senTxGrid, ref.grid)                   # copies and zero blocks are exact
    assert rel(acc.senTxWave, ref.wave) < 1e-10
    s_cols = [i for i, s in enumerate(slots) if tdd[s % 5] == "S"]
    assert s_cols and all(np.all(acc.senTxGrid[:, 14 * i:14 * (i + 1), :] == 0) for i in s_cols)
    t_slot = acc.T // len(slots)
    assert all(np.all(acc.senTxWave[t_slot * i:t_slot * (i + 1), :] == 0) for i in s_cols)
    with pytest.raises(pkg.IsacError) as ei:
        acc.append(qpsk((12 * nrb, 14, n_ants), 1), 0)
    assert ei.value.name == "CAPACITY"
    # the accumulated device arrays drive the sensing chain directly (no host hop): same estimates as from the host copies
    if nfft == 4096:
        return
    d_grid, d_wave = acc.device_arrays()
    assert np.array_equal(d_grid.numpy(), ref.grid) and rel(d_wave.numpy(), ref.wave) < 1e-10
