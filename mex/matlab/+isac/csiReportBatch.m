function [CQI, i1, i2, subbandCQI, sinrPerSubband, riTotalSINR] = csiReportBatch(carrier, csirsInd, reportConfig, nLayers, Hests, nVars, SINRTable)
%CSIREPORTBATCH  isac.csiReport for all UEs of a cell at one CSI-RS occasion (uePhy.m:901-908 runs once per UE) in ONE library call: Hests is a
%   cell array of the UEs' channel estimates [K x L x nRx x P], nVars their noise variances.  Outputs have one COLUMN per UE
%   (CQI: wideband index, then the subband differential values; i1: [i11; i12; i13]; i2: one row per PMI subband; riTotalSINR: riSelect.m:253-276's
%   totalSINR of this rank per UE -- communication.phyLayer.riSelect loops the ranks over this call).
    K = carrier.NSizeGrid * 12;  L = carrier.SymbolsPerSlot;  n = numel(Hests);
    [k, l, p] = ind2sub([K L size(Hests{1}, 4)], double(csirsInd(:)));
    k = k(p == 1);  l = l(p == 1);
    Hre = zeros(numel(k), size(Hests{1}, 3), size(Hests{1}, 4), n);
    for u = 1:n
        for i = 1:numel(k), Hre(i, :, :, u) = Hests{u}(k(i), l(i), :, :); end
    end
    rc = struct('NSizeBWP', carrier.NSizeGrid, 'NStartBWP', 0, 'PanelDimensions', double(reportConfig.PanelDimensions(1, :)), ...
                'CodebookMode', reportConfig.CodebookMode, 'PMIMode', reportConfig.PMIMode, 'CQIMode', reportConfig.CQIMode, ...
                'SubbandSize', reportConfig.SubbandSize(1));
    [CQI, i1, i2, subbandCQI, sinrPerSubband, riTotalSINR] = isac_mex('csiReportBatch', complex(Hre), k, l, rc, nLayers, double(nVars(:).'), double(SINRTable(:)));
end
