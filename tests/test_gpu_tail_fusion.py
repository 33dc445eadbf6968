"""fft2D.m:59-99 after the power window: cfar_panel_kernel (CA-CFAR on row panels) + cfar_merge_kernel (CUT-order merge, numDets)
against the memset + per-antenna CFAR + count kernels of the same library (ISAC_OPT_TAIL_FUSION = 0) bit for bit, and against the oracle:
zone shapes that give one panel, many panels, a ragged last panel, wide / narrow Doppler zones; thousands of detections per antenna
(the merge); repeated and re-shaped calls on one context (nothing carries over between calls: every workgroup writes only its own slots)."""
from __future__ import annotations

import numpy as np
import pytest

import oracle as O
from conftest import load_pkg, make_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return load_pkg()


def _both(pkg, sc, pfa=None, ctxs=None):
    if pfa is not None:
        sc.rp.Pfa = pfa
    rx = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, sc.los, sc.noise, nfft=sc.wave.Nfft)
    ocf = O.cfar2d_config(sc.rp)
    try:
        want, dbg = O.fft2d(sc.rp, ocf, rx, sc.tx_grid, return_debug=True)
    except ValueError:
        want, dbg = None, None
    out = []
    for k, fused in enumerate((True, False)):
        c = ctxs[k] if ctxs else pkg.Context()
        c.set_tail_fusion(fused)
        rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
        if pfa is not None:
            rp.Pfa = pfa
        cf = pkg.sensing.detection.cfar2D(rp)
        d_rx, d_tx = c.to_device(rx), c.to_device(sc.tx_grid)
        try:
            out.append(pkg.sensing.estimation.fft2D(rp, cf, d_rx, d_tx, return_debug=True))
        except pkg.IsacError as e:
            assert e.name == "NO_DETECTION" and want is None
            out.append(None)
    if want is None:
        assert out == [None, None]
        return None
    (e1, d1), (e0, d0) = out
    assert np.array_equal(d1.power_window, d0.power_window)                       # same arithmetic, bit for bit
    for a in range(sc.A):
        assert np.array_equal(d1.detections[a], d0.detections[a]) and np.array_equal(d1.det_pow[a], d0.det_pow[a]), f"antenna {a}"
        assert np.array_equal(d1.detections[a], dbg.detections[a]), f"antenna {a} vs oracle"
    for e in (e1, e0):
        assert np.array_equal(e.rngEst, want.rngEst) and np.array_equal(e.velEst, want.velEst) and np.array_equal(e.aziEst, want.aziEst)
    return e1, d1


@pytest.mark.parametrize("area", [((50.0, 500.0), (-50.0, 50.0)),          # default zone
                                  ((60.0, 90.0), (-50.0, 50.0)),           # a few CUT rows: one (ragged) panel
                                  ((50.0, 101.0), (-20.0, 20.0)),          # exactly fills / just overflows one 42-row panel
                                  ((50.0, 2000.0), (-50.0, 50.0)),         # many panels
                                  ((100.0, 400.0), (-200.0, 200.0)),       # wide Doppler zone (more window columns)
                                  ((100.0, 400.0), (-5.0, 5.0))])          # three CUT columns
@pytest.mark.parametrize("n_ants", [1, 5])
def test_fused_tail_equals_separate_kernels(pkg, area, n_ants):
    sc = make_scene(n_ants=n_ants, n_slots=4, nrb=273, targets=((100.0, 20.0, 1.5), (250.0, -120.0, 1.5)), velocity=(7.0, -4.0), seed=31,
                    detection_area=area)
    assert sc.rp.nFFT == 256
    _both(pkg, sc)


def test_fused_tail_small_numerology(pkg):
    """24 PRB: nIFFT = 512 (the range rows come from the Stockham range kernel), nFFT = 256."""
    sc = make_scene(n_ants=3, n_slots=6, nrb=24, targets=((150.0, 40.0, 1.5), (-90.0, 70.0, 5.0)), velocity=(0.0, 6.0), seed=21)
    assert sc.rp.nFFT == 256 and sc.rp.nIFFT == 512
    _both(pkg, sc)


def test_fused_tail_merge_with_thousands_of_detections(pkg):
    """Pfa = 0.5: about every other CUT detects (> 4096 per antenna, every (column, panel) segment non-empty): the CUT-order merge."""
    sc = make_scene(n_ants=3, n_slots=4, nrb=273, targets=((100.0, 20.0, 1.5),), velocity=(7.0,), seed=21)
    e, d = _both(pkg, sc, pfa=0.5)
    assert min(x.shape[1] for x in d.detections) > 4096 and e.rngEst.size > 300
    for a in range(sc.A):                                                        # CUT order: column-major over the zone, rows fastest
        r, c = d.detections[a]
        key = c.astype(np.int64) * 100000 + r
        assert np.all(np.diff(key) > 0)


def test_fused_tail_repeated_and_reshaped_calls_on_one_context(pkg):
    """No state carries over between calls (lists, counts and row masks are rewritten in full): calls in a row, a zone change, an
    antenna-count change and a call with no detection at all in between must each give the right answer on the same context."""
    ctxs = [pkg.Context(), pkg.Context()]
    a = make_scene(n_ants=4, n_slots=4, nrb=273, targets=((100.0, 20.0, 1.5),), velocity=(7.0,), seed=41)
    b = make_scene(n_ants=2, n_slots=4, nrb=273, targets=((300.0, 50.0, 1.5),), velocity=(-3.0,), seed=42, detection_area=((200.0, 400.0), (-30.0, 30.0)))
    quiet = make_scene(n_ants=4, n_slots=4, nrb=273, targets=((100.0, 20.0, 1.5),), velocity=(7.0,), seed=43, detection_area=((700.0, 900.0), (-50.0, 50.0)))
    first = _both(pkg, a, ctxs=ctxs)
    for sc in (a, a, b, quiet, a, b, b, a):
        got = _both(pkg, sc, ctxs=ctxs)
        if sc is a:
            assert np.array_equal(got[0].rngEst, first[0].rngEst) and np.array_equal(got[1].power_window, first[1].power_window)
