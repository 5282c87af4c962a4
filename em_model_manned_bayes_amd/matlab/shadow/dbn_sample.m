function [initial, events] = dbn_sample(parms, dirichlet_initial, dirichlet_transition, t_max, start)
% DBN_SAMPLE  Same signature as the reference's dbn_sample (dbn_sample.m:1,25-33): one trajectory of the dynamic Bayesian
% network, raw events [dt var bin] without terminator, resampling or dediscretisation.  UNTESTED (no MATLAB here).
    if nargin < 5, start = {}; end
    [initial, events] = emgpu_dbn_call(parms, dirichlet_initial, dirichlet_transition, t_max, start, 2 + 4 + 8);
end
