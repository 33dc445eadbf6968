function [aziEst, eleEst] = digitalBF(numDets, radarEstParams, Ra)
%DIGITALBF  Drop-in replacement body for +sensing/+estimation/+doaEstimation/digitalBF.m (digitalBF.m:1, ULA branch).
    [aziEst, eleEst] = isac_mex('digitalBF', numDets, isac.estBlock(radarEstParams), Ra);
end
