function rxWaveforms = applyCDLBatch(channelModels, txWaveform)
%APPLYCDLBATCH  The per-UE loop around  rxWaveform = obj.ChannelModel(rxWaveform)  (uePhy.m:724-731: every UE of a cell applies its own
%   nrCDLChannel to the SAME gNB slot waveform) as ONE library call: channelModels is a cell array of the UEs' nrCDLChannel objects
%   (cdl.m:57-64, switched to ChannelFiltering = false as for applyCDL; one delay profile per call: the objects must share the number of
%   paths and the path filters), txWaveform the slot waveform [T x Nt] (with the MaxChannelDelay zero rows of uePhy.m:729 appended).
%   rxWaveforms is [T x Nr x nUE].  The toolbox keeps drawing every UE's path gains; the antenna contractions and delay filters of
%   all UEs run as one contraction launch + one filter launch on the MI355X (isac_cdl_apply_batch_dev).
    n = numel(channelModels);
    T = size(txWaveform, 1);
    for u = 1:n
        cdl = channelModels{u};
        cdl.NumTimeSamples = T;
        [pg, st] = cdl();                                     % [Ncs x Np x Nt x Nr], [Ncs x 1]
        if u == 1
            pathGains = zeros([size(pg, 1), size(pg, 2), size(pg, 3), size(pg, 4), n], 'like', complex(pg));
            sampleTimes = zeros(numel(st), n);
            pathFilters = getPathFilters(cdl).';              % [Nh x Np]
        end
        pathGains(:, :, :, :, u) = pg;
        sampleTimes(:, u) = st;
    end
    cdl = channelModels{1};
    rxWaveforms = isac_mex('applyCDLBatch', complex(double(txWaveform)), complex(double(pathGains)), double(sampleTimes), double(pathFilters), ...
                           cdl.SampleRate, cdl.NormalizeChannelOutputs);
end
