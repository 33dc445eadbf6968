"""The C ABI driven from plain C (tests/abi_host.c, linked against libisac_hip.so, no Python in the process): the exact call sequences
the MEX gateway issues -- host-pointer `monoStaticSensing` + `fft2D` as with MATLAB arrays, and the device-handle sequence
(`toDevice` -> `monoStaticSensing` handle -> `fft2D` on handles -> `gather`) -- on the scene of the committed golden fixture
tests/golden/chain_small.npz; results compared with the fixture (echo grid <= 1e-10, estimates exact) and the reference's
all-targets-blocked error convention (ISAC_ERR_NO_LOS).  A second run times the host-pointer CPI (PCIe inclusive: what the MATLAB
drop-in pays when it hands over MATLAB arrays) beside the device-resident CPI at the benchmark shape."""
from __future__ import annotations

import json
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle as O
from conftest import ROOT, make_scene

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "tests", "_build", "abi_host")


def _exe():
    if not os.path.exists(EXE):
        import __graft_entry__ as g
        g.build_abi_host()
    return EXE


def _read_result(buf, off):
    (n_echo,) = struct.unpack_from("<Q", buf, off); off += 8
    echo = np.frombuffer(buf, dtype=np.complex128, count=n_echo, offset=off); off += 16 * n_echo
    n = struct.unpack_from("<3i", buf, off); off += 12
    out = []
    for k in n:
        out.append(np.frombuffer(buf, dtype=np.float64, count=k, offset=off)); off += 8 * k
    return echo, out, off


def test_c_host_matches_golden_fixture(tmp_path):
    g = np.load(os.path.join(ROOT, "tests", "golden", "chain_small.npz"))
    sc = make_scene(n_ants=int(g["n_ants"]), n_slots=int(g["n_slots"]), nrb=int(g["nrb"]), targets=tuple(map(tuple, g["targets"])),
                    velocity=tuple(g["velocity"]), num_slots_param=int(g["num_slots_param"]), seed=int(g["seed"]))
    cf = O.cfar2d_config(sc.rp)
    # struct scene_hdr (tests/abi_host.c): 18 int32 (K L A Q nfft scs n_ifft n_fft guard[2] train[2] row0 row1 col0 col1 has_noise pad), int64 T, 8 doubles
    hdr = struct.pack("<18i q 8d", sc.K, sc.L, sc.A, int(sc.rp.nTargets), sc.wave.Nfft, 30, int(sc.rp.nIFFT), int(sc.rp.nFFT), 2, 2, 1, 1,
                      int(cf.CUTIdx[0].min()), int(cf.CUTIdx[0].max()), int(cf.CUTIdx[1].min()), int(cf.CUTIdx[1].max()), 1, 0, sc.T,
                      float(sc.rp.fc), float(sc.rp.fs), float(sc.rp.N0), float(sc.rp.rRes), float(sc.rp.vRes), float(cf.Pfa),
                      float(sc.rp.azimuthScanScale), float(sc.rp.azimuthScanGranularity))
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(hdr)
        for a in (np.asarray(sc.rp.range, np.float64), np.asarray(sc.rp.velocity, np.float64), np.asarray(sc.rp.largeScaleFading, np.float64)):
            f.write(np.ascontiguousarray(a).tobytes())
        f.write(np.asfortranarray(sc.rp.RxSteeringVec.astype(np.complex128)).tobytes(order="F"))
        f.write(np.ones(int(sc.rp.nTargets), np.uint8).tobytes())
        for a in (sc.tx_wave, sc.noise, sc.tx_grid):
            f.write(np.asfortranarray(a).tobytes(order="F"))
    r = subprocess.run([_exe(), "chain", str(fin), str(fout)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    buf = open(fout, "rb").read()
    off = 0
    for path in ("host pointers", "device handles"):
        echo, (rng, vel, azi), off = _read_result(buf, off)
        echo = echo.reshape((sc.K, sc.L, sc.A), order="F")
        sub = echo[::5, ::3, :]
        assert np.abs(sub - g["echo_grid_sub"]).max() <= 1e-10 * np.abs(g["echo_grid_sub"]).max(), path
        assert np.array_equal(rng, g["rngEst"]) and np.array_equal(vel, g["velEst"]) and np.array_equal(azi, g["aziEst"]), path
    (code,) = struct.unpack_from("<i", buf, off)
    assert code == 3                                                   # ISAC_ERR_NO_LOS: every target blocked (basicRadarChannel.m:59,64)


def test_c_host_pointer_path_timing_beside_device_resident():
    r = subprocess.run([_exe(), "time", "64", "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    print("abi_host timing:", json.dumps(res))
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "r02_abi_host_timing.json"), "w") as f:
            json.dump(res, f)
    assert res["device_resident_cpi_ms"] > 0 and res["device_resident_fused_cpi_ms"] > 0
    assert res["host_pointer_cpi_ms"] > 3 * res["device_resident_cpi_ms"]          # the PCIe hop dominates the host-pointer path
    assert 90.0 < res["rngEst0"] < 120.0


@pytest.mark.parametrize("ants,n_ctx", [(16, 8), (64, 8)])
def test_c_host_batched_submit(ants, n_ctx):
    """isac_sensing_submit_n / isac_sensing_collect_n from the plain-C host: n_ctx cells' (monoStaticSensing -> fft2D) pairs per call pair with lazy echo grids -- the same
    estimates as the single call pairs on the same inputs and seeds, and the rate a host without an interpreter in its loop reaches (A = 16 is the reference's default
    array, ula.m:45: per-cell Python calls were host-bound at 62 k slots/s in round 5)."""
    r = subprocess.run([_exe(), "batch", str(ants), str(n_ctx), "60"], capture_output=True, text=True, timeout=600, env=dict(os.environ, GPU_MAX_HW_QUEUES="16"))
    assert r.returncode == 0, r.stdout + r.stderr
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    print("abi_host batch:", json.dumps(res))
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, f"r06_abi_host_batch_a{ants}.json"), "w") as f:
            json.dump(res, f)
    assert res["batch_equals_single_calls"] is True
    assert res["batch_slots_per_s"] > res["blocking_single_slots_per_s"]
    assert 90.0 < res["rngEst0"] < 120.0


def test_submit_n_python_mirror_matches_single_calls():
    """sensing.submitN / SensingBatch.collect (the Python face of the same two entry points): four cells on four contexts, lazy echo grids and caller-owned ones, one cell with
    every target blocked (its status is NO_LOS, the others are unaffected) -- results equal the per-cell calls."""
    from conftest import load_pkg
    pkg = load_pkg()
    scs = [make_scene(n_ants=a, n_slots=2, nrb=273, targets=((100.0 + 10 * i, 20.0, 1.5),), velocity=(7.0,), seed=60 + i, zero_s_slots=False, with_noise=False)
           for i, a in enumerate((64, 64, 64, 64))]
    ctxs = [pkg.Context() for _ in scs]
    rps = [pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave) for sc in scs]
    cf = pkg.sensing.detection.cfar2D(rps[0])
    waves = [c.to_device(sc.tx_wave) for c, sc in zip(ctxs, scs)]
    grids = [c.to_device(sc.tx_grid) for c, sc in zip(ctxs, scs)]
    los = [np.ones(1, np.uint8), np.ones(1, np.uint8), np.zeros(1, np.uint8), np.ones(1, np.uint8)]
    echo = [None, ctxs[1].empty(scs[1].tx_grid.shape), None, None]
    batch = pkg.sensing.submitN(ctxs, waves, grids, scs[0].tx_grid.shape, scs[0].carrier, rps, los, cf, seeds=[11, 12, 13, 14], nfft=4096, echoGrids=echo)
    res = batch.collect()
    assert isinstance(res[2], pkg.IsacError) and res[2].name == "NO_LOS"
    for i in (0, 1, 3):
        lz = pkg.sensing.monoStaticSensing(waves[i], scs[i].tx_grid.shape, scs[i].carrier, rps[i], los[i], nfft=4096, fuse_fft2d=(rps[i], cf, grids[i]), ctx=ctxs[i],
                                           lazy=True, seed=11 + i, noise_domain="spectral")
        try:
            want = pkg.sensing.estimation.fft2D(rps[i], cf, lz, grids[i], reuse_range=True)
        except pkg.IsacError as e:
            assert isinstance(res[i], pkg.IsacError) and res[i].name == e.name
            continue
        assert np.array_equal(res[i].rngEst, want.rngEst) and np.array_equal(res[i].velEst, want.velEst) and np.array_equal(res[i].aziEst, want.aziEst)
    for c in ctxs:
        c.close()


def test_graft_entry_smoke_runs():
    """The driver's own entry point: __graft_entry__.smoke() end to end (it is not otherwise part of the test suite)."""
    import __graft_entry__ as g
    g.smoke()
