from .basicRadarChannel import basicRadarChannel  # noqa: F401
