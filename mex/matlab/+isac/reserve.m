function ms = reserve(waveformLength, txDimension, carrierInfo, radarParams, cfar, warmMs)
%RESERVE  Prepare the GPU context for the sensing chain of one cell BEFORE the call that matters.
%   ms = isac.reserve(size(senTxWave, 1), txDimension, carrierInfo, radarParams, cfar [, warmMs])
%   The reference runs  sensing.monoStaticSensing -> sensing.estimation.fft2D  ONCE per cell and simulation
%   (+simulation/cellSimulation.m:189-202; one parfor worker per cell: +simulation/networkSimulation.m:47-60), so a process's first call would pay
%   for loading the library's code objects, sizing its scratch buffers, building twiddle / Kaiser / sind tables and for the clocks of an idle device.
%   This runs that very chain dry -- on QPSK grids the library generates itself, with the caller's parameter structs -- until warmMs of wall time
%   have passed (default 100).  Call it where the scenario is set up (in front of the cell loop / at worker start).  Returns the wall time in ms.
    if nargin < 6, warmMs = 100; end
    ms = isac_mex('reserve', double(waveformLength), double(txDimension), carrierInfo, isac.channelBlock(radarParams), isac.estBlock(radarParams), cfar, warmMs);
end
