from .updateCDLModels import updateCDLModels  # noqa: F401
from .cdl import CDLChannel, applyCDL, applyCDLBatch, csiEstimateBatch  # noqa: F401
