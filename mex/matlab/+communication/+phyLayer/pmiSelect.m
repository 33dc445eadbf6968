function [pmi, sinr, subbandIndices] = pmiSelect(nlayers, hest, noiseest, bandSize)
%PMISELECT  Drop-in for +communication/+phyLayer/pmiSelect.m:28-65 (the gNB's SRS measurement, gNBPhy.m:1033): the LMMSE SINR of every SRS resource
%   element for every TPMI of the PUSCH codebook and the subband means (sinrPerSubband.m:12-36) run on the GPU (isac_srs_pmi_select_batch_dev).
%   hest [K x L x R x P] (one or two SRS ports).  pmi: 0-based TPMI per subband of bandSize PRBs, NaN where the subband has no estimate;
%   sinr: the SINR of each subband's PMI (the reference returns the whole nSB x nTPMI table: only sinr(i, pmi(i) + 1) is used, gNBPhy.m:1041-1044).
    [K, L] = size(hest, [1 2]);
    have = sum(hest, 3:4) ~= 0;                                          % pmiSelect.m:36
    if ~any(have(:)) || noiseest == 0
        pmi = NaN; sinr = NaN; subbandIndices = NaN; return
    end
    [k, l] = find(have);
    Hre = zeros(numel(k), size(hest, 3), size(hest, 4));
    for i = 1:numel(k), Hre(i, :, :) = hest(k(i), l(i), :, :); end
    nrb = ceil(K / 12);
    [pmi, sinr] = isac_mex('srsReportBatch', complex(Hre), double(k), nrb, bandSize, nlayers, double(noiseest), zeros(0, 1));
    nSB = numel(pmi);
    present = false(nSB, 1);
    present(min(floor((k - 1) / (12 * bandSize)) + 1, nSB)) = true;      % subbands without an estimate stay NaN here (the gNB's fill is gNBPhy.m:1035-1040)
    pmi(~present) = NaN;  sinr(~present) = NaN;
    r = nrb / bandSize;  extra = ones(floor(r) ~= r);
    subbandIndices = [12 * bandSize * [0:r-1 floor(r) * extra]' + 1, 12 * bandSize * [1:r r * extra]'];
end
