"""Oracle restatement of the mono-static sensing transmit accumulation inside ``gNBPhy.phyTx``
(+communication/+phyLayer/gNBPhy.m:591-612): for every slot that carries PDSCH the slot grid is OFDM-modulated
(nrOFDMModulate, :599), scaled by signalAmp (:592,:602) and -- in a 'D' slot of the TDD pattern
(+communication/determineSlotType.m:5) -- appended to senTxGrid / senTxWave; any other slot type appends zeros of the same
size (:609-612).  Slots without PDSCH append nothing.  Note the asymmetry the sensing path relies on: senTxWave is the
SCALED waveform, senTxGrid the UNSCALED grid (:607-608).  TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import numpy as np

from .matlab_compat import db2mag
from .ofdm import ofdm_modulate


def determine_slot_type(tdd_pattern: str, slot_idx: int) -> str:
    """determineSlotType.m:5."""
    return tdd_pattern[slot_idx % len(tdd_pattern)]


def signal_amp(tx_power_dbm: float, nfft: int, n_sc: int, n_tx_ants: int) -> float:
    """gNBPhy.m:592."""
    return float(db2mag(tx_power_dbm - 30.0)) * np.sqrt(nfft ** 2 / (n_sc * n_tx_ants))


class SenTx:
    """senTxGrid [K x 14 n x A] / senTxWave [T x A] as gNBPhy accumulates them."""

    def __init__(self, nfft: int, scs_khz: float, tdd_pattern: str, tx_power_dbm: float, windowing: int = 0):
        self.nfft, self.scs, self.tdd, self.pw, self.win = nfft, scs_khz, tdd_pattern, tx_power_dbm, windowing
        self.grid = None
        self.wave = None

    def append(self, tx_grid_slot: np.ndarray, curr_slot: int):
        """One phyTx call with a non-empty PDSCHPDU (gNBPhy.m:595-612)."""
        k, l, a = tx_grid_slot.shape
        slots_per_subframe = int(round(self.scs / 15.0))
        first_symbol = (curr_slot % slots_per_subframe) * l                               # carrier.NSlot = CurrSlot (:579)
        wave = ofdm_modulate(tx_grid_slot, self.nfft, self.scs, self.win, first_symbol)   # :599
        wave = signal_amp(self.pw, self.nfft, k, a) * wave                                # :602
        if determine_slot_type(self.tdd, curr_slot) == "D":                               # :605-608
            g, w = tx_grid_slot, wave
        else:                                                                             # :609-612
            g, w = np.zeros_like(tx_grid_slot), np.zeros_like(wave)
        self.grid = g.copy() if self.grid is None else np.concatenate([self.grid, g], axis=1)
        self.wave = w.copy() if self.wave is None else np.concatenate([self.wave, w], axis=0)
