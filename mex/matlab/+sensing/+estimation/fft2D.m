function estResults = fft2D(radarEstParams, cfar, rxGrid, txGrid)
%FFT2D  Drop-in replacement body for +sensing/+estimation/fft2D.m (same signature): range-Doppler map, per-antenna
%   2D CA-CFAR, range / velocity estimates, array covariance and MUSIC azimuths on the MI355X.
%   Returns the struct with fields rngEst, velEst, aziEst, eleEst.  Zero detections raise isac:NO_DETECTION (the
%   reference raises from findpeaks at the same place).
    estResults = isac_mex('fft2D', isac.estBlock(radarEstParams), cfar, rxGrid, txGrid);
end
