function wt = packWallTable(city)
%PACKWALLTABLE  Flat wall table for isac_mex('checkLoS', ...): every wall of every building of a
%   networkTopology.blockages.city object (corner lists, plane normal and distance as computed by the reference's
%   wallBlockage constructor).  Build it once after the city has been constructed.
    walls   = [city.buildings.wallList];
    nWalls  = numel(walls);
    offsets = zeros(1, nWalls + 1, 'int32');
    for w = 1:nWalls
        offsets(w + 1) = offsets(w) + int32(size(walls(w).cornerList, 2));
    end
    wt = struct('corners', [walls.cornerList], 'offsets', offsets, ...
                'normals', [walls.normVec], 'normDist', [walls.normDist]);
end
