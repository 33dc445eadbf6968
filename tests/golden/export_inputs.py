"""Writes tests/golden/inputs_<name>.mat -- the seeded scene of tests/conftest.make_scene in a form MATLAB reads -- for
tests/golden/make_golden.m, which runs the UNMODIFIED reference on it and writes ref_<name>.mat (see README.md here).

    python tests/golden/export_inputs.py [chain_small] [config1_a16] ...

The AWGN is NOT exported: the reference draws it itself (basicRadarChannel.m:68); make_golden.m records the draws."""
from __future__ import annotations

import os
import sys

import numpy as np
from scipy import io as sio

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from conftest import make_scene  # noqa: E402
from make_golden import FULL  # noqa: E402

SCENES = {"chain_small": dict(n_ants=8, n_slots=4, nrb=24, targets=((150.0, 40.0, 1.5), (-90.0, 70.0, 5.0)), velocity=(0.0, 6.0),
                              num_slots_param=6, seed=21), **FULL}


def export(name):
    kw = dict(SCENES[name]); kw["with_noise"] = False
    sc = make_scene(**kw)
    c = sc.cell
    ant = c.gNBSenAntenna
    sio.savemat(os.path.join(HERE, f"inputs_{name}.mat"), dict(
        tx_grid=sc.tx_grid, tx_wave=sc.tx_wave, signalAmp=sc.amp, noiseSeed=kw["seed"], los=sc.los.astype(np.float64),
        numTargets=c.numTargets, targetPosition=np.atleast_2d(c.targetPosition), gNBPosition=np.asarray(c.gNBPosition, dtype=np.float64),
        tddPattern=np.array(list(c.tddPattern), dtype=object), numDLSlots=c.numDLSlots, numSlots=c.numSlots, gNBTxAnts=c.gNBTxAnts,
        dlCarrierFreq=c.dlCarrierFreq, gNBNoiseFigure=c.gNBNoiseFigure, gNBTemperature=c.gNBTemperature, gNBTxPower=c.gNBTxPower,
        gNBRxGain=c.gNBRxGain, rcs=np.asarray(c.rcs, dtype=np.float64), velocity=np.asarray(c.velocity, dtype=np.float64), Pfa=c.Pfa,
        detectionArea=np.asarray(c.detectionArea, dtype=np.float64), antenna_nV=ant.nV, antenna_p=ant.p, antenna_d=ant.d,
        NRBsDL=sc.carrier.NRBsDL, SubcarrierSpacing=sc.carrier.SubcarrierSpacing), do_compression=True)
    print(f"inputs_{name}.mat: txGrid {sc.tx_grid.shape} txWaveform {sc.tx_wave.shape}")


if __name__ == "__main__":
    for n in (sys.argv[1:] or ["chain_small"]):
        export(n)
