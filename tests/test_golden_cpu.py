"""The committed golden vectors are what the oracle produces today (regression pin for the oracle
itself; the GPU parity test consumes the same file)."""
import hashlib
import os

import numpy as np

import oracle as O
from conftest import make_scene


def test_oracle_reproduces_golden_chain():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "chain_small.npz"))
    sc = make_scene(n_ants=int(g["n_ants"]), n_slots=int(g["n_slots"]), nrb=int(g["nrb"]), targets=tuple(map(tuple, g["targets"])),
                    velocity=tuple(g["velocity"]), num_slots_param=int(g["num_slots_param"]), seed=int(g["seed"]))
    assert hashlib.sha256(np.ascontiguousarray(sc.tx_grid).tobytes()).hexdigest() == str(g["tx_grid_sha256"])
    echo = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, sc.los, sc.noise, nfft=sc.wave.Nfft)
    assert np.abs(echo[::5, ::3, :] - g["echo_grid_sub"]).max() <= 1e-12 * np.abs(g["echo_grid_sub"]).max()
    est, dbg = O.fft2d(sc.rp, O.cfar2d_config(sc.rp), echo, sc.tx_grid, return_debug=True)
    assert np.array_equal(est.rngEst, g["rngEst"]) and np.array_equal(est.velEst, g["velEst"])
    assert np.array_equal(est.aziEst, g["aziEst"]) and est.aziEst.size >= 1
    assert np.array_equal(np.concatenate(dbg.detections, axis=1), g["det_idx"])
