from .music import music  # noqa: F401
