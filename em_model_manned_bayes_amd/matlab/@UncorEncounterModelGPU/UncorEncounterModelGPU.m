classdef UncorEncounterModelGPU < UncorEncounterModel
    % UncorEncounterModelGPU  Drop-in subclass whose .sample and .track run on MI355X GPUs through libemgpu.
    % UNTESTED (no MATLAB in the build image; emgpu_mex.c is only type-checked).  Same signatures and outputs as
    % UncorEncounterModel.sample / .track (@UncorEncounterModel/UncorEncounterModel.m:192-313, 318-471); every property,
    % setter and helper is inherited from the reference class unchanged.
    %
    %   mdl = UncorEncounterModelGPU('parameters_filename', f);
    %   [out_inits, out_events, out_samples, out_EME] = mdl.sample(n, T, 'seed', 1);
    %   out_results = mdl.track(n, T, 'initialSeed', 1);          % coordSys 'NEU'
    %   emgpu_mex('use_devices', 0:7);                            % optional: split every .sample call over 8 GPUs
    %
    % The native model is rebuilt from struct(self) at every call (emgpu_mex('from_struct')), so edits of N_initial,
    % N_transition, prior, start, boundaries, zero_bins and resample_rates made through the reference's own setters
    % (EncounterModel.m:156-210) are what gets sampled.

    methods
        function self = UncorEncounterModelGPU(varargin)
            self@UncorEncounterModel(varargin{:});
        end

        function [out_inits, out_events, out_samples, out_EME] = sample(self, n_samples, sample_time, varargin)
            p = inputParser;
            addParameter(p, 'seed', nan, @isnumeric);
            addParameter(p, 'isQuantize500', false, @islogical);
            addParameter(p, 'layers', [], @isnumeric);
            parse(p, varargin{:});
            seed = p.Results.seed;
            if isnan(seed) || isempty(seed), seed = randi(2^31 - 1); end   % no global stream on the device: derive a key from it
            idx = @(name) max([0 find(strcmp(self.labels_initial, name))]);
            idxL = idx('"L"'); idxV = idx('"v"'); idxDV = idx('"\dot v"'); idxDH = idx('"\dot h"'); idxDPsi = idx('"\dot \psi"');
            if idxDV == 0 || idxDH == 0 || idxDPsi == 0                   % UncorEncounterModel.m:231-234
                error('dynvar:empty', 'Model does not have a dynamic variable for either acceleration, vertical rate, or turn rate');
            end
            s = self.struct;
            h = emgpu_mex('from_struct', s); cleanup = onCleanup(@() emgpu_mex('free', h));
            emgpu_mex('set_alpha', h, s.dirichlet_initial, s.dirichlet_transition);
            cap = 256;
            while true                                                    % the longest event list must fit: double and retry
                try
                    [out_inits, ev_count, E] = emgpu_mex('sample_uncor', h, n_samples, sample_time, seed, 0, double(p.Results.isQuantize500), ...
                                                         idxL, idxV, idxDH, p.Results.layers, cap);
                    break
                catch err
                    if ~strcmp(err.identifier, 'emgpu:eventcap'), rethrow(err); end
                    cap = 2 * cap;
                end
            end
            out_events = cell(n_samples, 1); out_samples = cell(n_samples, 1); out_EME(n_samples, 1) = EncounterModelEvents;
            idxEME = [find(s.temporal_map(:, 1) == idxDH), find(s.temporal_map(:, 1) == idxDPsi), find(s.temporal_map(:, 1) == idxDV)] + 1;   % :291
            for ii = 1:n_samples
                events = E(1:ev_count(ii), :, ii);
                out_events{ii} = events;
                out_samples{ii} = events2samples(out_inits(ii, :), events);          % the reference's own helpers
                controls = events2controls(out_inits(ii, :), events, s);
                controls = controls(:, [1 idxEME]);
                controls(:, 2) = controls(:, 2) / 60; controls(:, 3) = deg2rad(controls(:, 3)); controls(:, 4) = controls(:, 4) * 1.68780972222222;   % :295-297
                out_EME(ii) = EncounterModelEvents('event', controls);
            end
        end

        function out_results = track(self, nSamples, sample_time, varargin)
            % coordSys 'NEU' only; anything else falls through to the reference's own .track (em-core, DEM, placeTrack).
            p = inputParser; p.KeepUnmatched = true;
            addParameter(p, 'initialSeed', nan, @isnumeric);
            addParameter(p, 'isQuantize500', false, @islogical);
            addParameter(p, 'coordSys', 'NEU', @ischar);
            parse(p, varargin{:});
            if ~strcmpi(p.Results.coordSys, 'NEU')
                out_results = track@UncorEncounterModel(self, nSamples, sample_time, varargin{:});
                return
            end
            seed = p.Results.initialSeed;
            if isnan(seed) || isempty(seed), seed = randi(2^31 - 1); end
            idx = @(name) max([0 find(strcmp(self.labels_initial, name))]);
            idx7 = [idx('"G"') idx('"A"') idx('"L"') idx('"v"') idx('"\dot v"') idx('"\dot h"') idx('"\dot \psi"')];   % :385-391
            s = self.struct;
            h = emgpu_mex('from_struct', s); cleanup = onCleanup(@() emgpu_mex('free', h));
            emgpu_mex('set_alpha', h, s.dirichlet_initial, s.dirichlet_transition);
            tracks = emgpu_mex('track_uncor', h, nSamples, sample_time, seed, 0, double(p.Results.isQuantize500), double(self.isRotorcraft), idx7, 1, 1000);
            out_results = cell(nSamples, 1);
            for ii = 1:nSamples                                              % :473-476
                r = tracks(:, :, ii)';
                out_results{ii} = timetable(r(:, 2), r(:, 3), r(:, 4), r(:, 5), r(:, 6), r(:, 7), r(:, 8), ...
                                            'VariableNames', {'north_ft', 'east_ft', 'up_ft', 'speed_ft_s', 'phi_rad', 'theta_rad', 'psi_rad'}, ...
                                            'RowTimes', seconds(r(:, 1)));
            end
        end
    end
end
