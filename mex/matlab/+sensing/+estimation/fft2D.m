function estResults = fft2D(radarEstParams, cfar, rxGrid, txGrid)
%FFT2D  Drop-in replacement body for +sensing/+estimation/fft2D.m (same signature): range-Doppler map, per-antenna
%   2D CA-CFAR, range / velocity estimates, array covariance and MUSIC azimuths on the MI355X.
%   Returns the struct with fields rngEst, velEst, aziEst, eleEst.  Zero detections raise isac:NO_DETECTION (the
%   reference raises from findpeaks at the same place).  The CUT rectangle, guard / training band sizes and Pfa are read
%   from cfar.CUTIdx and cfar.cfarDetector2D (sensing.detection.cfar2D stays the reference's own MATLAB code).
%   rxGrid / txGrid may be MATLAB arrays or uint64 device handles (isac_mex('toDevice', .), or the handle that the
%   device-resident sensing.monoStaticSensing returns): with handles nothing crosses PCIe but the estimates.
    estResults = isac_mex('fft2D', radarEstParams, cfar, rxGrid, txGrid);
end
