"""ctypes binding of libisac_hip.so (C ABI: include/isac.h).  Fails loudly when the
HIP library is missing -- there is deliberately no CPU fallback."""
from __future__ import annotations

import ctypes as C
import os
import sys
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libisac_hip.so")
_lib = None
_lock = threading.Lock()

ISAC_ABI_VERSION = 7          # include/isac.h ISAC_ABI_VERSION this binding was written against (checked at load)
ISAC_MAX_EST = 4096
NOISE_NONE, NOISE_INJECTED, NOISE_PHILOX, NOISE_PHILOX_SPECTRAL, NOISE_INJECTED_SPECTRAL = 0, 1, 2, 3, 4

STATUS_NAMES = {0: "OK", 1: "INVALID_ARG", 2: "HIP", 3: "NO_LOS", 4: "NO_DETECTION", 5: "CFAR_WINDOW",
                6: "CAPACITY", 7: "UNSUPPORTED", 8: "SHORT_WAVEFORM"}


class IsacError(RuntimeError):
    """Raised for every non-zero isac_status; ``.code`` / ``.name`` carry the status.

    The reference's caller wraps fft2D in try/catch and maps any error to
    ``senResults = NaN`` (cellSimulation.m:196-202); callers here keep that convention
    by catching IsacError."""

    def __init__(self, code: int, message: str):
        self.code = int(code)
        self.name = STATUS_NAMES.get(int(code), str(code))
        super().__init__(f"isac[{self.name}]: {message}")


class c64(C.Structure):
    _fields_ = [("re", C.c_double), ("im", C.c_double)]


class Carrier(C.Structure):
    _fields_ = [("n_sc", C.c_int32), ("nfft", C.c_int32), ("scs_khz", C.c_int32), ("reserved", C.c_int32)]


class RadarChannelParams(C.Structure):
    _fields_ = [("fc", C.c_double), ("fs", C.c_double), ("n0", C.c_double), ("n_ants", C.c_int32),
                ("n_targets", C.c_int32), ("range", C.POINTER(C.c_double)), ("velocity", C.POINTER(C.c_double)),
                ("large_scale_fading", C.POINTER(C.c_double)), ("rx_steering", C.c_void_p)]


class CfarConfig(C.Structure):
    _fields_ = [("pfa", C.c_double), ("guard", C.c_int32 * 2), ("train", C.c_int32 * 2),
                ("row0", C.c_int32), ("row1", C.c_int32), ("col0", C.c_int32), ("col1", C.c_int32)]


class EstParams(C.Structure):
    _fields_ = [("n_ifft", C.c_int32), ("n_fft", C.c_int32), ("r_res", C.c_double), ("v_res", C.c_double),
                ("array_is_upa", C.c_int32), ("n_ants_x", C.c_int32), ("n_ants_y", C.c_int32),
                ("azimuth_scan_scale", C.c_double), ("azimuth_scan_granularity", C.c_double),
                ("elevation_scan_scale", C.c_double), ("elevation_scan_granularity", C.c_double)]


class Music2dParams(C.Structure):
    _fields_ = [("fc", C.c_double), ("t_sri", C.c_double), ("scs_hz", C.c_double), ("r_max", C.c_double), ("v_max", C.c_double)]


ISAC_MAX_SUBBANDS = 70


class CsiReport(C.Structure):
    _fields_ = [("n_subbands_pmi", C.c_int32), ("n_subbands_cqi", C.c_int32), ("n_cqi", C.c_int32), ("reserved", C.c_int32),
                ("i1", C.c_double * 3), ("i2", C.c_double * ISAC_MAX_SUBBANDS), ("cqi", C.c_double * (ISAC_MAX_SUBBANDS + 1)),
                ("subband_cqi", C.c_double * (ISAC_MAX_SUBBANDS + 1)), ("sinr_per_subband_cw", C.c_double * (ISAC_MAX_SUBBANDS + 1)), ("ri_total_sinr", C.c_double)]


ISAC_MAX_RBS = 275


class SrsReport(C.Structure):
    _fields_ = [("n_subbands", C.c_int32), ("n_tpmi", C.c_int32), ("n_rb", C.c_int32), ("reserved", C.c_int32),
                ("pmi", C.c_double * (ISAC_MAX_SUBBANDS + 1)), ("sinr_subband_pmi", C.c_double * (ISAC_MAX_SUBBANDS + 1)), ("cqi_rb", C.c_double * ISAC_MAX_RBS)]


class CdlJob(C.Structure):
    _fields_ = [("d_x", C.c_void_p), ("d_y", C.c_void_p), ("d_H", C.c_void_p), ("block_start", C.c_void_p), ("n_blocks", C.c_int32), ("reserved", C.c_int32)]


class SensingJob(C.Structure):
    _fields_ = [("d_tx_wave", C.c_void_p), ("d_tx_grid", C.c_void_p), ("d_echo_grid", C.c_void_p), ("rp", C.c_void_p), ("los", C.c_void_p),
                ("d_noise_unit", C.c_void_p), ("seed", C.c_uint64), ("noise_mode", C.c_int32), ("reserved", C.c_int32)]


class EstResult(C.Structure):
    _fields_ = [("n_rng", C.c_int32), ("n_vel", C.c_int32), ("n_azi", C.c_int32), ("num_dets", C.c_int32),
                ("total_detections", C.c_int32), ("reserved", C.c_int32),
                ("rng_est", C.c_double * ISAC_MAX_EST), ("vel_est", C.c_double * ISAC_MAX_EST),
                ("azi_est", C.c_double * ISAC_MAX_EST), ("ele_est", C.c_double * ISAC_MAX_EST)]


# every symbol include/isac.h declares (tests check the library exports all of them)
EXPORTS = [
    "isac_abi_version", "isac_abi_sizeof", "isac_device_count", "isac_ctx_create", "isac_ctx_destroy", "isac_last_error",
    "isac_ctx_get_stream", "isac_sync", "isac_dev_alloc", "isac_dev_free", "isac_memcpy_h2d", "isac_memcpy_d2h", "isac_memcpy_d2d",
    "isac_memset_dev", "isac_timer_start", "isac_timer_stop_ms", "isac_profile_enable", "isac_profile_last_kernel_ms",
    "isac_basic_radar_channel_dev", "isac_basic_radar_channel", "isac_mono_static_sensing_dev",
    "isac_mono_static_sensing", "isac_mono_static_sensing_fused_dev", "isac_echo_grid_materialize_dev", "isac_ofdm_symbol_count", "isac_ofdm_demodulate_dev", "isac_ofdm_modulate_dev", "isac_ofdm_modulate_windowed_dev", "isac_sentx_append_dev",
    "isac_ofdm_waveform_length", "isac_cfar2d_ca", "isac_fft2d_dev", "isac_fft2d", "isac_fft2d_submit_dev", "isac_fft2d_submit_cached_dev", "isac_fft2d_collect", "isac_sensing_submit_n", "isac_sensing_collect_n", "isac_fft2d_range_stage_dev", "isac_fft2d_get_detections",
    "isac_fft2d_get_power_window", "isac_fft2d_get_covariance", "isac_fft2d_get_music_spectrum",
    "isac_rdm_plane_dev", "isac_covariance_dev", "isac_music_doa", "isac_ctx_set_option", "isac_ctx_share_streams", "isac_ctx_reserve", "isac_eigh_top", "isac_beamscan_doa", "isac_music2d_dev", "isac_eigh", "isac_cdl_apply_dev", "isac_cdl_apply_batch_dev", "isac_cdl_path_gains_dev", "isac_cdl_freq_response_dev", "isac_cdl_csi_estimate_batch_dev", "isac_prg_precode_dev", "isac_precoded_sinr_cqi_dev", "isac_type1sp_codebook", "isac_csi_report_dev", "isac_csi_report_batch_dev", "isac_pusch_codebook", "isac_srs_pmi_select_batch_dev", "isac_los_check_dev", "isac_winding_number_dev", "isac_synth_qpsk_grid_dev",
]


def library_path() -> str:
    return _LIB_PATH


def load():
    """Load libisac_hip.so once.  torch (if installed) is imported first so that both share one
    HIP runtime (same SONAME libamdhip64.so.7); loading order the other way round would map two."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the sensing hot path.")
        # (The loader does not touch the process environment.  Pipelined hosts want one hardware queue per HIP stream --
        # GPU_MAX_HW_QUEUES >= 2 x contexts in flight, set by the APPLICATION before the HIP runtime starts: INTEGRATION.md section 4,
        # bench.py and examples/ do so and report the effective value.)
        if "torch" not in sys.modules and os.environ.get("ISAC_NO_TORCH_PRELOAD") != "1":
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        lib = C.CDLL(_LIB_PATH, mode=C.RTLD_GLOBAL)
        lib.isac_last_error.restype = C.c_char_p
        for name in EXPORTS:
            fn = getattr(lib, name)  # AttributeError here = ABI drift between isac.h and the .so
            if name != "isac_last_error":
                fn.restype = C.c_int
        # the library writes whole structs into caller memory: version AND struct sizes must match this binding's mirrors
        if lib.isac_abi_version() != ISAC_ABI_VERSION:
            raise RuntimeError(f"{_LIB_PATH}: ABI version {lib.isac_abi_version()} but this binding was written for {ISAC_ABI_VERSION}; rebuild the library")
        for which, (name, cls) in enumerate((("isac_est_result", EstResult), ("isac_est_params", EstParams), ("isac_cfar_config", CfarConfig),
                                             ("isac_radar_channel_params", RadarChannelParams), ("isac_carrier", Carrier),
                                             ("isac_music2d_params", Music2dParams), ("isac_csi_report", CsiReport), ("isac_sensing_job", SensingJob), ("isac_srs_report", SrsReport))):
            if lib.isac_abi_sizeof(C.c_int32(which)) != C.sizeof(cls):
                raise RuntimeError(f"{_LIB_PATH}: sizeof({name}) = {lib.isac_abi_sizeof(C.c_int32(which))} in the library, {C.sizeof(cls)} in the binding")
        _lib = lib
        return lib


def _np_ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class DeviceArray:
    """Column-major array resident in HBM, owned by a Context (freed with it or via .free())."""

    def __init__(self, ctx: "Context", ptr: int, shape, dtype, owner: bool = True):
        self.ctx, self.ptr, self.shape, self.dtype, self._owner = ctx, int(ptr), tuple(int(s) for s in shape), np.dtype(dtype), owner

    @property
    def nbytes(self) -> int:
        return int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize

    def numpy(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype, order="F")
        self.ctx.check(self.ctx.lib.isac_memcpy_d2h(self.ctx.handle, _np_ptr(out), C.c_void_p(self.ptr), C.c_size_t(self.nbytes)))
        return out

    def free(self):
        if self._owner and self.ptr:
            self.ctx.lib.isac_dev_free(self.ctx.handle, C.c_void_p(self.ptr))
            self.ptr = 0

    def __del__(self):
        try:
            if self.ctx.handle:
                self.free()
        except Exception:
            pass


class Context:
    """One device + one HIP stream + scratch (isac_ctx).  Not thread-safe."""

    def __init__(self, device: int | None = None):
        self.lib = load()
        if device is None:
            device = int(os.environ.get("ISAC_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        n = C.c_int(0)
        st = self.lib.isac_device_count(C.byref(n))
        if st != 0 or n.value <= 0:
            raise IsacError(2, "no HIP device visible: the sensing hot path needs an MI355X (no CPU fallback)")
        h = C.c_void_p()
        st = self.lib.isac_ctx_create(C.c_int(device % n.value), C.byref(h))
        if st != 0:
            raise IsacError(st, f"isac_ctx_create(device={device}) failed")
        self.handle = h
        self.device = device % n.value
        self.device_cache = {}              # device-resident tables other modules keep per context (CDL per-ray terms, CSI frequency tables): dropped by close()

    def close(self):
        if getattr(self, "handle", None):
            for ent in list(self.device_cache.values()):                 # free the cached DeviceArrays while the context still exists
                for d in (ent if isinstance(ent, tuple) else (ent,)):
                    for dd in (d if isinstance(d, tuple) else (d,)):
                        if isinstance(dd, DeviceArray):
                            dd.free()
            self.device_cache.clear()
            self.lib.isac_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, status: int):
        if status != 0:
            raise IsacError(status, (self.lib.isac_last_error(self.handle) or b"").decode())

    def sync(self):
        self.check(self.lib.isac_sync(self.handle))

    def stream(self) -> int:
        s = C.c_void_p()
        self.check(self.lib.isac_ctx_get_stream(self.handle, C.byref(s)))
        return int(s.value or 0)

    def empty(self, shape, dtype=np.complex128) -> DeviceArray:
        dt = np.dtype(dtype)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
        p = C.c_void_p()
        self.check(self.lib.isac_dev_alloc(self.handle, C.c_size_t(nbytes), C.byref(p)))
        return DeviceArray(self, p.value, shape, dt)

    def to_device(self, a: np.ndarray) -> DeviceArray:
        a = np.asfortranarray(a)
        d = self.empty(a.shape, a.dtype)
        self.check(self.lib.isac_memcpy_h2d(self.handle, C.c_void_p(d.ptr), _np_ptr(a), C.c_size_t(a.nbytes)))
        return d

    def set_music_route(self, route: int):
        """0 = MUSIC through the signal-subspace eigensolver (default), 1 = always the full eigendecomposition (ISAC_OPT_MUSIC_ROUTE)."""
        self.check(self.lib.isac_ctx_set_option(self.handle, C.c_int32(0), C.c_int32(int(route))))

    def set_tail_fusion(self, on: bool):
        """True (default) = panel CFAR (antenna x 42-CUT-row workgroups) + per-antenna merge that also forms numDets, where the zone allows;
        False = memset of the row flags + one CFAR workgroup per antenna + a separate count kernel (ISAC_OPT_TAIL_FUSION, include/isac.h)."""
        self.check(self.lib.isac_ctx_set_option(self.handle, C.c_int32(1), C.c_int32(1 if on else 0)))

    def set_cdl_share_spectra(self, on: bool):
        """ISAC_OPT_CDL_SHARE_SPECTRA: consecutive overlap-save downlink batches on the SAME waveform arrays share their forward transforms (the caller promises the
        waveforms are not rewritten in between; see include/isac.h)."""
        self.check(self.lib.isac_ctx_set_option(self.handle, C.c_int32(3), C.c_int32(1 if on else 0)))

    def set_wide_order(self, on: bool):
        """ISAC_OPT_WIDE_ORDER: fft2D's covariance on the main stream, every narrow kernel on the second (see include/isac.h)."""
        self.check(self.lib.isac_ctx_set_option(self.handle, C.c_int32(2), C.c_int32(1 if on else 0)))

    def share_streams(self, owner: "Context | None"):
        """Enqueue on `owner`'s two streams from now on (None: back to this context's own): isac_ctx_share_streams."""
        self.check(self.lib.isac_ctx_share_streams(self.handle, owner.handle if owner is not None else None))
        self._stream_owner = owner          # keep the owner alive

    def eigh_top(self, h, n_top: int):
        """(w, U): all eigenvalues ascending + the eigenvectors of the n_top largest (descending order) -- isac_eigh_top."""
        h = as_c128_f(h)
        a = h.shape[0]
        w = np.zeros(a)
        u = np.zeros((a, max(int(n_top), 1)), dtype=np.complex128, order="F")
        self.check(self.lib.isac_eigh_top(self.handle, _np_ptr(h), C.c_int32(a), C.c_int32(int(n_top)), _np_ptr(w), _np_ptr(u)))
        return w, u[:, : int(n_top)]

    def timer_start(self):
        self.check(self.lib.isac_timer_start(self.handle))

    def timer_stop_ms(self) -> float:
        ms = C.c_double(0)
        self.check(self.lib.isac_timer_stop_ms(self.handle, C.byref(ms)))
        return ms.value


_default_ctx = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx


def as_f64(x) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1))


def as_c128_f(x) -> np.ndarray:
    return np.asfortranarray(np.asarray(x, dtype=np.complex128))
