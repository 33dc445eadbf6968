from .fft2D import fft2D, fft2D_submit, fft2D_collect  # noqa: F401
from . import doaEstimation  # noqa: F401
from .music2D import music2D  # noqa: F401
