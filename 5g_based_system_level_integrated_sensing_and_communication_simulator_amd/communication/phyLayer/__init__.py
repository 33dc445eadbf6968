"""Hot-path seams of the reference's ``+communication/+phyLayer``: batched precodedSINR and the SINR->CQI lookup."""
from .precodedSINR import precodedSINR, getCQI, cqiFromChannel, DOWNLINK_SINR90PC, UPLINK_SINR90PC  # noqa: F401
from .senTx import SenTx, nrOFDMModulate, determineSlotType, signalAmp  # noqa: F401,E402
from .csiReport import cqiSelect, cqiSelectBatch, dlPMISelect, type1SinglePanelCodebook  # noqa: F401,E402
from .prgPrecode import prgPrecode, prgPrecodeGrid  # noqa: F401,E402
from .pmiSelect import pmiSelect, srsReportBatch, puschCodebook, maxPUSCHPrecodingMatrixIndicator  # noqa: F401,E402
from .riSelect import riSelect, riSelectBatch  # noqa: F401,E402
