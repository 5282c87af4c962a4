function S = bn_sample(G, r, N, alpha, num_samples, start, order) %#ok<INUSD>
% BN_SAMPLE  Same signature as the reference's bn_sample (bn_sample.m:1): num_samples x n matrix of bins, drawn on the GPU.
% `order` is recomputed by the library (bn_sort.m semantics).  The key comes from MATLAB's global stream, so rng(s) before
% the call makes it reproducible; the draws themselves are the library's Philox slots.  UNTESTED (no MATLAB here).
    n = numel(N);
    parms = struct('G_initial', logical(G), 'N_initial', {N(:)}, 'n_initial', n);
    h = emgpu_mex('from_struct', parms); cleanup = onCleanup(@() emgpu_mex('free', h));
    emgpu_mex('set_alpha', h, alpha, {});
    if nargin >= 6 && ~isempty(start), emgpu_mex('set_start', h, emgpu_start_vector(start, n)); end
    S = emgpu_mex('bn_sample', h, num_samples, randi(2^31 - 1), 0);
    if numel(r) ~= n, error('bn_sample:size', 'r must have one entry per variable'); end
end
