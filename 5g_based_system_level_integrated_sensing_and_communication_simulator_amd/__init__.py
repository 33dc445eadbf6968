"""MI355X-native sensing hot path of the BUPT 5G ISAC system-level simulator.

Host-side mirror of the reference's MATLAB package API for the hot path only
(``+sensing`` and ``+communication/+channelModels``): same function names,
argument meaning and error behaviour, implemented as thin ctypes calls into
``libisac_hip.so`` (hand-written HIP for gfx950, C ABI in ``include/isac.h``).

There is no CPU fallback: importing the package works anywhere, but the first
call that needs the library raises if ``libisac_hip.so`` has not been built
(``__graft_entry__.build()``) or no MI355X is visible.

The package name starts with a digit, so import it by string::

    import importlib
    isac = importlib.import_module("5g_based_system_level_integrated_sensing_and_communication_simulator_amd")
    est = isac.sensing.estimation.fft2D(radarParams, cfarConfig, rxGrid, txGrid)
"""
from . import _lib  # noqa: F401
from ._lib import Context, DeviceArray, IsacError, default_context, library_path  # noqa: F401
from . import sensing, communication, networkTopology  # noqa: F401

__all__ = ["sensing", "communication", "networkTopology", "Context", "DeviceArray", "IsacError", "default_context", "library_path"]
