"""Mirror of the reference's ``+networkTopology`` package (LoS blockage seam only)."""
from . import blockages  # noqa: F401
