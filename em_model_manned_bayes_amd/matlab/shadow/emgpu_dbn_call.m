function [initial, events] = emgpu_dbn_call(parms, di, dt, T, start, flags)
% One trajectory through emgpu_mex('sample_uncor') without the UncorEncounterModel rejection test (idxV = idxDH = 0).
    h = emgpu_mex('from_struct', parms); cleanup = onCleanup(@() emgpu_mex('free', h));
    emgpu_mex('set_alpha', h, di, dt);
    if ~isempty(start), emgpu_mex('set_start', h, emgpu_start_vector(start, parms.n_initial)); end
    seed = randi(2^31 - 1); cap = 256;
    while true
        try
            [initial, cnt, E] = emgpu_mex('sample_uncor', h, 1, T, seed, 0, flags, 0, 0, 0, [], cap);
            break
        catch err
            if ~strcmp(err.identifier, 'emgpu:eventcap'), rethrow(err); end
            cap = 2 * cap;
        end
    end
    events = E(1:cnt(1), :, 1);
end
