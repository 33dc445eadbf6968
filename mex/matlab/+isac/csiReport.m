function [CQI, PMISet, CQIInfo] = csiReport(carrier, csirsInd, reportConfig, nLayers, Hest, nVar, SINRTable)
%CSIREPORT  The CSI report of uePhy.m:901-908 -- communication.phyLayer.cqiSelect(carrier, csirs, reportConfig, nLayers, Hest, nVar,
%   SINRTable) around dlPMISelect's exhaustive Type-I single-panel search -- with the per-RE x per-codebook-entry SINR
%   evaluation, the subband means and the totals on the MI355X.  csirsInd = nrCSIRSIndices(carrier, csirs) (toolbox, stays
%   on the MATLAB side); only the first port's resource elements are used (dlPMISelect.m:354-362).
%       [cqi, pmiSet] = isac.csiReport(carrier, nrCSIRSIndices(carrier, csirsInfo), obj.CSIReportConfig, rank, Hest, nVar, obj.SINRTable);
    K = carrier.NSizeGrid * 12;  L = carrier.SymbolsPerSlot;
    [k, l, p] = ind2sub([K L size(Hest, 4)], double(csirsInd(:)));
    k = k(p == 1);  l = l(p == 1);
    Hre = zeros(numel(k), size(Hest, 3), size(Hest, 4));
    for i = 1:numel(k), Hre(i, :, :) = Hest(k(i), l(i), :, :); end
    rc = struct('NSizeBWP', carrier.NSizeGrid, 'NStartBWP', 0, 'PanelDimensions', double(reportConfig.PanelDimensions(1, :)), ...
                'CodebookMode', reportConfig.CodebookMode, 'PMIMode', reportConfig.PMIMode, 'CQIMode', reportConfig.CQIMode, ...
                'SubbandSize', reportConfig.SubbandSize(1));
    [CQI, i1, i2, subCQI, sinrSB] = isac_mex('csiReport', complex(Hre), k, l, rc, nLayers, nVar, double(SINRTable(:)));
    PMISet  = struct('i1', i1, 'i2', i2);
    CQIInfo = struct('SubbandCQI', subCQI, 'SINRPerSubbandPerCW', sinrSB);
end
