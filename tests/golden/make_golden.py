"""Generates tests/golden/*.npz from the CPU ORACLE (the MATLAB reference cannot run here and
ships no vectors of its own -- see oracle/__init__.py, "parity unpinned").  Inputs are
re-derived from the seed by tests/conftest.make_scene; outputs are the oracle's.

    python tests/golden/make_golden.py
"""
from __future__ import annotations

import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle as O  # noqa: E402
from conftest import make_scene  # noqa: E402


def chain_small():
    kw = dict(n_ants=8, n_slots=4, nrb=24, targets=((150.0, 40.0, 1.5), (-90.0, 70.0, 5.0)), velocity=(0.0, 6.0),
              num_slots_param=6, seed=21)
    sc = make_scene(**kw)
    echo = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, sc.los, sc.noise, nfft=sc.wave.Nfft)
    est, dbg = O.fft2d(sc.rp, O.cfar2d_config(sc.rp), echo, sc.tx_grid, return_debug=True)
    np.savez_compressed(os.path.join(HERE, "chain_small.npz"), n_ants=kw["n_ants"], n_slots=kw["n_slots"], nrb=kw["nrb"],
                        targets=np.array(kw["targets"]), velocity=np.array(kw["velocity"]), num_slots_param=kw["num_slots_param"],
                        seed=kw["seed"], tx_grid_sha256=hashlib.sha256(np.ascontiguousarray(sc.tx_grid).tobytes()).hexdigest(),
                        echo_grid_sub=echo[::5, ::3, :], rngEst=est.rngEst, velEst=est.velEst,
                        aziEst=est.aziEst, det_idx=np.concatenate(dbg.detections, axis=1), Ra=dbg.Ra)
    print("chain_small:", est)


if __name__ == "__main__":
    chain_small()
