classdef SenTx < handle
%SENTX  Device-resident senTxGrid / senTxWave of gNBPhy (gNBPhy.m:48-52): replaces the two cat() accumulations of
%   gNBPhy.m:604-612.  In phyTx, instead of
%       obj.senTxGrid = cat(2, obj.senTxGrid, txGrid);  obj.senTxWave = cat(1, obj.senTxWave, txWaveform);
%   call   obj.senTx.append(txGrid, obj.CurrSlot, currSlotType == 'D', signalAmp)   -- the slot is OFDM-modulated on the GPU
%   (nrOFDMModulate + windowing, gNBPhy.m:599), scaled and stored ('D') or stored as zeros (any other slot type).  After the slot
%   loop, cellSimulation.m:191-197 passes obj.senTx.wave / obj.senTx.grid (uint64 handles) straight to
%   sensing.monoStaticSensing / sensing.estimation.fft2D.
    properties
        grid; wave; carrierInfo; windowing; nSlots = 0;
    end
    methods
        function obj = SenTx(carrierInfo, nTxAnts, maxSlots, windowing)
            info = nrOFDMInfo(carrierInfo.NRBsDL, carrierInfo.SubcarrierSpacing);
            obj.carrierInfo = carrierInfo;  obj.windowing = windowing;
            slotLen = sum(info.SymbolLengths(1:info.SymbolsPerSlot));
            obj.grid = isac_mex('allocDevice', [12 * carrierInfo.NRBsDL, 14 * maxSlots, nTxAnts]);
            obj.wave = isac_mex('allocDevice', [slotLen * maxSlots, nTxAnts]);
        end
        function append(obj, txGrid, currSlot, isDL, signalAmp)
            isac_mex('senTxAppend', obj.grid, obj.wave, complex(double(txGrid)), currSlot, isDL, obj.carrierInfo, signalAmp, obj.windowing, obj.nSlots);
            obj.nSlots = obj.nSlots + 1;
        end
        function delete(obj)
            isac_mex('free', obj.grid);  isac_mex('free', obj.wave);
        end
    end
end
