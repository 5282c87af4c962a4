function v = emgpu_start_vector(start, n)
% start cell ([] or NaN = unset, bn_sample.m:45 in both copies) -> numeric vector with NaN for unset
    v = nan(1, n);
    for ii = 1:min(n, numel(start))
        if ~isempty(start{ii}) && ~isnan(start{ii}(1)), v(ii) = start{ii}(1); end
    end
end
