function echoGrid = monoStaticSensing(txWaveform, txDimension, carrierInfo, radarParams, targetLoSConditions)
%MONOSTATICSENSING  Drop-in replacement body for +sensing/monoStaticSensing.m of the reference:
%   same signature and result (echo resource grid [nSc x nSym x nAnts]), computed on the MI355X through isac_mex.
%   Put this folder BEFORE the reference on the MATLAB path.  Errors of the library arrive as MException with
%   identifier 'isac:<CODE>' (e.g. isac:NO_LOS when every target is blocked), so the try/catch around the sensing
%   call in simulation.cellSimulation keeps producing senResults = NaN.
%
%   The AWGN of basicRadarChannel is drawn here with randn (MATLAB's stream, as in the reference) and handed to the
%   library as unit-variance noise; drop the last argument to use the on-device Philox generator instead.
    noiseUnit = complex(randn(size(txWaveform)), randn(size(txWaveform)));
    echoGrid  = isac_mex('monoStaticSensing', txWaveform, double(txDimension), carrierInfo, ...
                         isac.channelBlock(radarParams), uint8(targetLoSConditions(:) == 1), noiseUnit);
end
