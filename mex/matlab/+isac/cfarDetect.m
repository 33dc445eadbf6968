function detections = cfarDetect(cfar, P)
%CFARDETECT  detections = cfarDetector(P, CUTIdx) of fft2D.m:62 on the device for callers that keep the reference's own fft2D body:
%   cfar = sensing.detection.cfar2D(radarParams); returns [2 x D] detection indices in CUT order.
    detections = isac_mex('cfarDetector', double(P), double(cfar.CUTIdx), cfar.cfarDetector2D);
end
