function sinr = precodedSINR(H, sigma, W)
%PRECODEDSINR  Drop-in replacement body for +communication/+phyLayer/precodedSINR.m (precodedSINR.m:11-17): LMMSE SINR of
%   a precoded unit-power signal, summed over the layers.
    sinr = isac_mex('precodedSINR', complex(double(H)), double(sigma), complex(double(W)));
end
