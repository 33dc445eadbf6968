from .music import music  # noqa: F401
from .beamforming import digitalBF, mvdrBF  # noqa: F401
