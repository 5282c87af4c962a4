function [initial, events] = dbn_hierarchical_sample(parms, dirichlet_initial, dirichlet_transition, sample_time, dediscretize_parameters, zero_bins, resample_rates, start)
% DBN_HIERARCHICAL_SAMPLE  Same signature as the reference's (dbn_hierarchical_sample.m:1): dbn_sample + terminator row +
% resample_events + dediscretize, on the GPU.  UNTESTED (no MATLAB here).
    if nargin < 8, start = {}; end
    parms.boundaries = dediscretize_parameters; parms.zero_bins = zero_bins; parms.resample_rates = resample_rates;
    [initial, events] = emgpu_dbn_call(parms, dirichlet_initial, dirichlet_transition, sample_time, start, 0);
end
