"""CPU oracle for the ISAC sensing hot path  --  TEST INFRASTRUCTURE ONLY.

This package is a NumPy/SciPy fp64 *restatement* of the reference's MATLAB
sensing path (``/root/reference/+sensing/**``), written from the reference
sources and from the published behaviour of the MathWorks toolbox primitives it
calls.  Every function cites the reference file:line it follows.

PARITY UNPINNED: the reference is 100 % MATLAB and ships no tests, golden
vectors or fixtures; neither MATLAB nor Octave nor the proprietary toolboxes
(5G / Phased Array / Signal Processing) exist in the build container, so the
reference cannot be run or compiled here.  The oracle is therefore pinned only
by the analytic known-answer tests in ``tests/test_oracle_kat.py`` (closed-form
constants, range-bin/Doppler-bin placement, shift algebra, MUSIC on rank-1
covariances).  Toolbox semantics that could not be observed are listed in
DESIGN.md ("ambiguity list").

Who may import this package: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the *checker* or the timed CPU
baseline, never as part of the product path.  The product
(``5g_based_..._amd``) never imports it and fails loudly without its HIP
library.
"""
from .matlab_compat import (  # noqa: F401
    sind, cosd, kaiser, findpeaks, unique_stable, db2pow, db2mag, pow2db, mag2db,
    LIGHTSPEED, BOLTZMANN, EPS,
)
from .radar_params import radar_params, default_cell_params, nr_ofdm_info  # noqa: F401
from .ofdm import ofdm_modulate, ofdm_demodulate, cp_lengths, symbol_starts, raised_cosine_edge  # noqa: F401
from .radar_channel import basic_radar_channel, mono_static_sensing  # noqa: F401
from .cfar import cfar2d_config, ca_cfar2d, cfar_threshold_factor  # noqa: F401
from .fft2d import fft2d, rdm_literal, rdm_explicit, covariance  # noqa: F401
from .music import music_doa, determine_num_targets, music2d, digital_bf, mvdr_bf  # noqa: F401
from .philox import philox4x32_10, philox_normal_pairs, philox_spectral_noise  # noqa: F401
from . import cdl, cqi, sentx  # noqa: F401
