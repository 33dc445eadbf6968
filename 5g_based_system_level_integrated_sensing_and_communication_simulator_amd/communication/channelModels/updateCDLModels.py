"""communication.channelModels.updateCDLModels (+communication/+channelModels/updateCDLModels.m:1-17)."""
from __future__ import annotations


def updateCDLModels(simuParams):
    """LoS -> 'CDL-D', NLoS -> 'CDL-A' per UE (updateCDLModels.m:9-14)."""
    los = list(getattr(simuParams, "ueLoSConditions"))
    n = int(getattr(simuParams, "numUEs", len(los)))
    return ["CDL-A" if los[i] == 0 else "CDL-D" for i in range(n)]
