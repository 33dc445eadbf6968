from .updateCDLModels import updateCDLModels  # noqa: F401
from .cdl import CDLChannel, applyCDL, applyCDLBatch  # noqa: F401
