function rxWaveform = applyCDL(channelModel, txWaveform)
%APPLYCDL  Replacement for the toolbox step  rxWaveform = obj.ChannelModel(rxWaveform)  at uePhy.m:729-731 (downlink) and
%   gNBPhy.m:838-840 (uplink).  channelModel is the nrCDLChannel System object that +parameters/+channelModels/
%   +communication/cdl.m:57-64,78-85 configures, switched to ChannelFiltering = false: the toolbox keeps drawing the
%   TR 38.901 path gains (its own ray tables and random stream, so nothing about the channel statistics changes), the
%   delay filtering and the antenna contraction of step() run on the MI355X.
%
%       cdl = channelModel;  release(cdl);  cdl.ChannelFiltering = false;   % once, after cdl.m built the object
%       rx  = communication.channelModels.applyCDL(cdl, tx);
    channelModel.NumTimeSamples = size(txWaveform, 1);
    [pathGains, sampleTimes] = channelModel();               % [Ncs x Np x Nt x Nr], [Ncs x 1]
    pathFilters = getPathFilters(channelModel).';            % [Nh x Np]
    rxWaveform = isac_mex('applyCDL', complex(double(txWaveform)), complex(double(pathGains)), double(sampleTimes), double(pathFilters), ...
                          channelModel.SampleRate, channelModel.NormalizeChannelOutputs);
end
