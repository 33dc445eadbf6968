function echoGrid = monoStaticSensing(txWaveform, txDimension, carrierInfo, radarParams, targetLoSConditions)
%MONOSTATICSENSING  Drop-in replacement body for +sensing/monoStaticSensing.m of the reference:
%   same signature and result (echo resource grid [nSc x nSym x nAnts]), computed on the MI355X through isac_mex.
%   Put this folder BEFORE the reference on the MATLAB path.  Errors of the library arrive as MException with
%   identifier 'isac:<CODE>' (e.g. isac:NO_LOS when every target is blocked), so the try/catch around the sensing
%   call in simulation.cellSimulation keeps producing senResults = NaN.
%
%   Host arrays in -> host array out, with the AWGN of basicRadarChannel drawn here with randn (MATLAB's stream, as in the
%   reference) and handed to the library as unit-variance noise.
%   Device-resident use (no PCIe round trip of the 0.75 GB echo grid): pass txWaveform as a uint64 handle
%   (h = isac_mex('toDevice', senTxWave), or the senTx accumulator of isac.SenTx) -- the result is then a handle that
%   sensing.estimation.fft2D accepts, and the noise is the library's Philox generator drawn on the demodulated grid
%   (same distribution; include/isac.h, isac_noise_mode).
    los = uint8(targetLoSConditions(:) == 1);
    if isa(txWaveform, 'uint64')
        echoGrid = isac_mex('monoStaticSensing', txWaveform, double(txDimension), carrierInfo, isac.channelBlock(radarParams), los, ...
                            [], randi(2^31), 'spectral');
    else
        noiseUnit = complex(randn(size(txWaveform)), randn(size(txWaveform)));
        echoGrid  = isac_mex('monoStaticSensing', txWaveform, double(txDimension), carrierInfo, isac.channelBlock(radarParams), los, noiseUnit);
    end
end
