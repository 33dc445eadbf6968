"""Hot-path seams of the reference's ``+communication/+phyLayer``: batched precodedSINR and the SINR->CQI lookup."""
from .precodedSINR import precodedSINR, getCQI, cqiFromChannel, DOWNLINK_SINR90PC, UPLINK_SINR90PC  # noqa: F401
