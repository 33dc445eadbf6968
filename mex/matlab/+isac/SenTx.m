classdef SenTx < handle
%SENTX  Device-resident senTxGrid / senTxWave of gNBPhy (gNBPhy.m:48-52): replaces the two cat() accumulations of
%   gNBPhy.m:604-612.  In phyTx, instead of
%       obj.senTxGrid = cat(2, obj.senTxGrid, txGrid);  obj.senTxWave = cat(1, obj.senTxWave, txWaveform);
%   call   obj.senTx.append(txGrid, obj.CurrSlot, currSlotType == 'D', signalAmp)   -- the slot is OFDM-modulated on the GPU
%   (nrOFDMModulate + windowing, gNBPhy.m:599), scaled and stored ('D') or stored as zeros (any other slot type).  After the slot
%   loop, cellSimulation.m:191-197 passes [g, w] = obj.senTx.handles() (uint64 handles of exactly the accumulated size, as the
%   reference's cat() results are) straight to sensing.monoStaticSensing / sensing.estimation.fft2D.
%   The waveform offset of a slot is the running SAMPLE count, not slot index x slot length: at 60 / 120 kHz the slots of a
%   subframe differ in length (long cyclic prefix at symbols 0 and 7*2^mu only).
    properties
        grid; wave; carrierInfo; windowing; nSlots = 0; nSamples = 0; nSc; nAnts; maxSlots; waveRows;
    end
    methods
        function obj = SenTx(carrierInfo, nTxAnts, maxSlots, windowing)
            info = nrOFDMInfo(carrierInfo.NRBsDL, carrierInfo.SubcarrierSpacing);
            obj.carrierInfo = carrierInfo;  obj.windowing = windowing;
            obj.nSc = 12 * carrierInfo.NRBsDL;  obj.nAnts = nTxAnts;  obj.maxSlots = maxSlots;
            nSlotsSf = info.SlotsPerSubframe;  perSlot = info.SymbolsPerSlot;
            slotLens = arrayfun(@(s) sum(info.SymbolLengths((s-1)*perSlot + (1:perSlot))), 1:nSlotsSf);
            obj.waveRows = max(slotLens) * maxSlots;                     % capacity: the longest slot of a subframe, maxSlots times
            obj.grid = isac_mex('allocDevice', [obj.nSc, 14 * maxSlots, nTxAnts]);
            obj.wave = isac_mex('allocDevice', [obj.waveRows, nTxAnts]);
        end
        function append(obj, txGrid, currSlot, isDL, signalAmp)
            tLen = isac_mex('senTxAppend', obj.grid, obj.wave, complex(double(txGrid)), currSlot, isDL, obj.carrierInfo, signalAmp, obj.windowing, ...
                            obj.nSlots, obj.nSamples);
            obj.nSlots = obj.nSlots + 1;
            obj.nSamples = obj.nSamples + tLen;
        end
        function [g, w] = handles(obj)
            % arrays of exactly the accumulated size ([nSc x 14 nSlots x nAnts], [nSamples x nAnts]); the caller frees trimmed copies
            if obj.nSlots == obj.maxSlots, g = obj.grid; else, g = isac_mex('trim', obj.grid, obj.nSc, 14 * obj.nSlots); end
            if obj.nSamples == obj.waveRows, w = obj.wave; else, w = isac_mex('trim', obj.wave, obj.nSamples, obj.nAnts); end
        end
        function delete(obj)
            isac_mex('free', obj.grid);  isac_mex('free', obj.wave);
        end
    end
end
