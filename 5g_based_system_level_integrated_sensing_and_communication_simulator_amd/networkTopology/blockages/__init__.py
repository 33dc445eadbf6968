"""+networkTopology/+blockages: walls, buildings and the city-level LoS check."""
from .wallBlockage import wallBlockage  # noqa: F401
from .building import building  # noqa: F401
from .city import city  # noqa: F401
