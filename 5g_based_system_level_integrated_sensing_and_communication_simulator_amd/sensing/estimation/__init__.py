from .fft2D import fft2D  # noqa: F401
from . import doaEstimation  # noqa: F401
