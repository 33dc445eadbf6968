function [RI, PMISet] = riSelect(carrier, csirs, reportConfig, H, nVar)
%RISELECT  Drop-in for +communication/+phyLayer/riSelect.m:207-292 (uePhy.m:900), Type-I single panel: one CSI report per valid rank on the GPU
%   (isac.csiReportBatch: every rank's exhaustive PMI search; its sixth output is the rank's totalSINR of riSelect.m:253-276), the rank whose total
%   beats the best so far by more than 0.1.  RI = NaN when every total is NaN (no CSI-RS in the BWP).
    if nargin < 5, nVar = 1e-10; end
    csirsInd = nrCSIRSIndices(carrier, csirs);
    maxRank = min(size(H, 3), size(H, 4));
    restr = ones(1, 8);
    if isfield(reportConfig, 'RIRestriction') && ~isempty(reportConfig.RIRestriction), restr = reportConfig.RIRestriction; end
    validRanks = intersect(find(restr), 1:maxRank);
    best = -Inf;  RI = NaN;  PMISet = struct('i1', NaN(1, 3), 'i2', NaN);
    total = NaN(1, maxRank);
    for r = validRanks
        [~, i1, i2, ~, ~, tot] = isac.csiReportBatch(carrier, csirsInd, reportConfig, r, {H}, nVar, 0);
        total(r) = tot;
        if total(r) > best + 0.1
            best = total(r);  RI = r;  PMISet = struct('i1', i1(:).', 'i2', i2(:).');
        end
    end
    if all(isnan(total)), RI = NaN; end
end
