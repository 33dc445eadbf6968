function [antsym, antind] = prgPrecode(siz, nstartgrid, portsym, portind, F)
%PRGPRECODE  Drop-in replacement body for +communication/+phyLayer/prgPrecode.m of the reference (prgPrecode.m:53-144; called for the PDSCH and its
%   DM-RS at gNBPhy.m:822-827): same signature and results -- portsym / portind [nRE x nu] (1-based linear indices into the [K x L x nu] port grid),
%   F [nu x P x NPRG] -> antenna symbols and indices [nRE x P].  The product  layers(k, l, :) * F(:, :, prg(k))  runs on the MI355X
%   (isac_prg_precode_dev), the PRG of every RE as getPRGSet assigns it (:93-99).
    [antsym, antind] = isac_mex('prgPrecode', double(siz(1:2)), double(nstartgrid), complex(double(portsym)), double(portind), complex(double(F)));
end
