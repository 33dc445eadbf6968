function [L, aziEst, eleEst] = music(numDets, radarEstParams, Ra)
%MUSIC  Drop-in replacement body for +sensing/+estimation/+doaEstimation/music.m (same signature, ULA arrays).
    [L, aziEst, eleEst] = isac_mex('music', numDets, isac.estBlock(radarEstParams), Ra);
end
