function estResults = music2D(rdrEstParams, bsParams, rxGrid, txGrid)
%MUSIC2D  Drop-in replacement body for +sensing/+estimation/music2D.m (same signature, music2D.m:1): MUSIC azimuth,
%   range and velocity estimates; struct with fields aziEst, eleEst, rngEst, velEst.  The 3276 x 3276 eigenproblem of
%   music2D.m:71,77 is solved through the nSym x nSym Gram matrix on the device (same non-zero spectrum).
    r = isac_mex('music2D', rdrEstParams, double(bsParams.scs), rxGrid, txGrid);
    estResults = struct('aziEst', r.aziEst, 'eleEst', r.eleEst, 'rngEst', r.rngEst, 'velEst', r.velEst);
end
