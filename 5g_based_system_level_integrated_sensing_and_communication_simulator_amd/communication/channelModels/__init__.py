from .updateCDLModels import updateCDLModels  # noqa: F401
