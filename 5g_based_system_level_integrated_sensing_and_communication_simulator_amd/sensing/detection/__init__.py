from .cfar2D import cfar2D, CFARDetector2D  # noqa: F401
