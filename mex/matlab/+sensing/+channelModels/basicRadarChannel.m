function rxWaveform = basicRadarChannel(txWaveform, radarParams, targetLoSConditions)
%BASICRADARCHANNEL  Drop-in replacement body for +sensing/+channelModels/basicRadarChannel.m (same signature).
    noiseUnit  = complex(randn(size(txWaveform)), randn(size(txWaveform)));
    rxWaveform = isac_mex('basicRadarChannel', txWaveform, [], [], isac.channelBlock(radarParams), ...
                          uint8(targetLoSConditions(:) == 1), noiseUnit);
end
