"""Mirror of the reference's ``+communication`` package (hot-path seams only)."""
from . import channelModels, phyLayer  # noqa: F401
