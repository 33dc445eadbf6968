function [aziEst, eleEst] = mvdrBF(numDets, radarEstParams, Ra)
%MVDRBF  Drop-in replacement body for +sensing/+estimation/+doaEstimation/mvdrBF.m (mvdrBF.m:1, ULA branch).
    [aziEst, eleEst] = isac_mex('mvdrBF', numDets, isac.estBlock(radarEstParams), Ra);
end
