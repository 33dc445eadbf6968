from .updateCDLModels import updateCDLModels  # noqa: F401
from .cdl import CDLChannel, applyCDL  # noqa: F401
