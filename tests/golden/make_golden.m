function make_golden(name, refRoot)
%MAKE_GOLDEN  Reference-produced golden vectors for the sensing hot path (the ONLY thing that can pin parity).
%
%   make_golden('chain_small', '/path/to/5G_based_System_level_Integrated_Sensing_and_Communication_Simulator')
%
% Runs the UNMODIFIED reference functions
%     sensing.radarParams            (+sensing/radarParams.m:1)
%     sensing.monoStaticSensing      (+sensing/monoStaticSensing.m:1  ->  basicRadarChannel.m:1, nrOFDMDemodulate)
%     sensing.detection.cfar2D       (+sensing/+detection/cfar2D.m:1)
%     sensing.estimation.fft2D       (+sensing/+estimation/fft2D.m:1  ->  phased.CFARDetector2D, doaEstimation.music)
% on the inputs tests/golden/inputs_<name>.mat (written by `python tests/golden/export_inputs.py <name>`: the seeded scene of
% tests/conftest.make_scene -- txGrid, txWaveform, cell / carrier / waveform parameters) and writes tests/golden/ref_<name>.mat (v7).
% `python -m pytest tests/test_golden_cpu.py` then checks the repository's CPU oracle against that file, field by field.
%
% The AWGN of basicRadarChannel.m:67-69 is drawn inside the reference with randn: this script seeds the generator, records the very
% same draws (randn(size)+1j*randn(size): real part first), re-seeds, and lets the unmodified function draw them again -- the noise the
% reference used is therefore part of the fixture and the oracle is fed exactly that field.
%
% Needs: MATLAB R2022b or later (contents.m:28) with 5G Toolbox, Phased Array System Toolbox, Signal Processing Toolbox.
% Not runnable in the build container (no MATLAB): see tests/golden/README.md.

    here = fileparts(mfilename('fullpath'));
    if nargin < 1, name = 'chain_small'; end
    if nargin >= 2, addpath(refRoot); end
    in = load(fullfile(here, ['inputs_' name '.mat']));
    set(0, 'DefaultFigureVisible', 'off');                       % fft2D.m:119 / music.m:99 plot

    % ---- cellSimuParams exactly as scenarios/openStreetMapCity.m + gNBParameters.m fill it (fields radarParams.m reads)
    cellSimuParams                 = struct;
    cellSimuParams.numTargets      = double(in.numTargets);
    cellSimuParams.targetPosition  = double(in.targetPosition);     % [nTargets x 3]
    cellSimuParams.gNBPosition     = double(in.gNBPosition(:)');    % [1 x 3]
    cellSimuParams.tddPattern      = cellstr(in.tddPattern(:))';    % {'D','D','D','S','U'}
    cellSimuParams.numDLSlots      = double(in.numDLSlots);
    cellSimuParams.numSlots        = double(in.numSlots);
    cellSimuParams.gNBTxAnts       = double(in.gNBTxAnts);
    cellSimuParams.dlCarrierFreq   = double(in.dlCarrierFreq);
    cellSimuParams.gNBNoiseFigure  = double(in.gNBNoiseFigure);
    cellSimuParams.gNBTemperature  = double(in.gNBTemperature);
    cellSimuParams.gNBTxPower      = double(in.gNBTxPower);
    cellSimuParams.gNBRxGain       = double(in.gNBRxGain);
    cellSimuParams.rcs             = double(in.rcs(:)');
    cellSimuParams.velocity        = double(in.velocity(:)');
    cellSimuParams.Pfa             = double(in.Pfa);
    cellSimuParams.detectionArea   = double(in.detectionArea);      % [2 x 2]
    ant                            = parameters.baseStation.antenna.ula;
    ant.nV                         = double(in.antenna_nV);
    ant.p                          = double(in.antenna_p);
    ant.d                          = double(in.antenna_d);
    cellSimuParams.gNBSenAntenna   = ant;
    carrierInfo                    = struct('NRBsDL', double(in.NRBsDL), 'SubcarrierSpacing', double(in.SubcarrierSpacing));
    waveInfo                       = nrOFDMInfo(carrierInfo.NRBsDL, carrierInfo.SubcarrierSpacing);
    los                            = double(in.los(:)');

    radarParams = sensing.radarParams(cellSimuParams, carrierInfo, waveInfo);        % radarParams.m:1
    cfarConfig  = sensing.detection.cfar2D(radarParams);                              % cfar2D.m:1

    txGrid = in.tx_grid;                                                              % [K x L x A] complex double
    txWave = in.tx_wave;                                                              % [T x A]     complex double
    % toolbox cross-check of the repository's CP-OFDM modulator restatement (gNBPhy.m:599 calls nrOFDMModulate per slot)
    carrier = nrCarrierConfig; carrier.SubcarrierSpacing = carrierInfo.SubcarrierSpacing; carrier.NSizeGrid = carrierInfo.NRBsDL;
    nSlots = size(txGrid, 2) / 14; wv = cell(1, nSlots);
    for s = 1:nSlots
        carrier.NSlot = s - 1;
        wv{s} = nrOFDMModulate(carrier, txGrid(:, 14*(s-1)+1:14*s, :), 'Windowing', 0);
    end
    tbWave = cat(1, wv{:}) * double(in.signalAmp);
    modErr = max(abs(tbWave(:) - txWave(:))) / max(abs(txWave(:)));

    seed = double(in.noiseSeed);
    rng(seed, 'twister');
    nre = randn(size(txWave)); nim = randn(size(txWave));                             % basicRadarChannel.m:68, real part first
    noise_unit = complex(nre, nim);
    rng(seed, 'twister');
    echoGrid = sensing.monoStaticSensing(txWave, size(txGrid), carrierInfo, radarParams, los);   % monoStaticSensing.m:1
    estResults = sensing.estimation.fft2D(radarParams, cfarConfig, echoGrid, txGrid);             % fft2D.m:1

    % per-antenna detections + covariance, re-stated from fft2D.m:37-46,59-62,106-107 so that the fixture also pins the
    % intermediate integer outputs (the reference does not return them)
    [nSc, nSym, nAnts] = size(echoGrid);
    channelInfo = bsxfun(@times, echoGrid, pagectranspose(pagetranspose(txGrid)));
    rngWin = repmat(kaiser(nSc, 3), [1 nSym]); dopWin = repmat(kaiser(radarParams.nIFFT, 3), [1 nSym]);
    chlInfo = channelInfo .* rngWin;
    rngIFFT = ifftshift(ifft(chlInfo, radarParams.nIFFT, 1) .* sqrt(radarParams.nIFFT));
    rngIFFT = rngIFFT .* dopWin;
    rdm     = fftshift(fft(rngIFFT, radarParams.nFFT, 2) ./ sqrt(radarParams.nFFT));
    detIdx = cell(1, nAnts);
    for r = 1:nAnts
        detIdx{r} = rmmissing(cfarConfig.cfarDetector2D(abs(rdm(:, :, r)).^2, cfarConfig.CUTIdx), 2);   % fft2D.m:61-63, [2 x D], 1-based
    end
    det_counts = cellfun(@(d) size(d, 2), detIdx);
    det_idx    = cat(2, detIdx{:});
    rxGridReshaped = reshape(echoGrid, nSc*nSym, nAnts)';                               % fft2D.m:106  (conjugate transpose)
    Ra = rxGridReshaped*rxGridReshaped'./(nSc*nSym);                                    % fft2D.m:107
    rp = rmfield(radarParams, {'antennaType', 'targetRealPos'});
    matlabRelease = version; toolboxes = ver; toolboxes = strjoin(arrayfun(@(t) [t.Name ' ' t.Version], toolboxes, 'UniformOutput', false), '; ');
    rngEst = estResults.rngEst; velEst = estResults.velEst; aziEst = estResults.aziEst; eleEst = estResults.eleEst;
    CUTIdx = cfarConfig.CUTIdx;
    echo_grid = echoGrid;
    save(fullfile(here, ['ref_' name '.mat']), '-v7', 'rp', 'CUTIdx', 'noise_unit', 'echo_grid', 'rngEst', 'velEst', 'aziEst', 'eleEst', ...
         'det_idx', 'det_counts', 'Ra', 'modErr', 'matlabRelease', 'toolboxes', 'seed');
    fprintf('ref_%s.mat written: rng %s vel %s azi %s, %d detections, modulator restatement error %.2e\n', name, mat2str(rngEst, 6), ...
            mat2str(velEst, 6), mat2str(aziEst), size(det_idx, 2), modErr);
end
