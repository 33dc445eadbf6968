"""Mirror of the reference's ``+communication`` package (hot-path seams only)."""
from . import channelModels  # noqa: F401
