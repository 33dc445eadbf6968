"""Mirror of the reference's ``+sensing`` package (hot-path functions only)."""
from .radarParams import radarParams  # noqa: F401
from .monoStaticSensing import monoStaticSensing  # noqa: F401
from .reserve import reserve  # noqa: F401
from .submitN import submitN, SensingBatch  # noqa: F401
from . import channelModels, detection, estimation  # noqa: F401
