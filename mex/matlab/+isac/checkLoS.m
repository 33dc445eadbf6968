function losDecision = checkLoS(wallTable, uePos, antPos)
%CHECKLOS  Batched replacement for city.checkLoS(uePos, antPos): uePos [n x 3], antPos [1 x 3] or [n x 3] (row vectors as
%   simulation.networkSimulation passes them); returns logical [1 x n], true = line of sight.
    ue  = double(uePos).';
    ant = double(antPos).';
    if size(ant, 2) == 1
        ant = repmat(ant, 1, size(ue, 2));
    end
    losDecision = isac_mex('checkLoS', wallTable, ue, ant);
end
