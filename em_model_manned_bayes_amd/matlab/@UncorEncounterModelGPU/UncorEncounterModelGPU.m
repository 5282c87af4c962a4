classdef UncorEncounterModelGPU < UncorEncounterModel
    % UncorEncounterModelGPU  Drop-in subclass whose .sample runs on an MI355X through libemgpu.
    % UNTESTED (no MATLAB in the build image).  Same signature and outputs as
    % UncorEncounterModel.sample (@UncorEncounterModel/UncorEncounterModel.m:192-313); everything
    % else (.track, properties) is inherited from the reference class unchanged.
    %
    %   mdl = UncorEncounterModelGPU('parameters_filename', f);
    %   [out_inits, out_events, out_samples, out_EME] = mdl.sample(n, T, 'seed', 1);

    properties (Access = private)
        h uint64 = uint64(0)   % emgpu_model handle
    end

    methods
        function self = UncorEncounterModelGPU(varargin)
            self@UncorEncounterModel(varargin{:});
            p = inputParser; p.KeepUnmatched = true;
            addParameter(p, 'parameters_filename', [getenv('AEM_DIR_BAYES') filesep 'model' filesep 'uncor_1200only_fwse_v1p2.txt']);
            addParameter(p, 'idxZeroBoundaries', [1 2 3]);
            addParameter(p, 'isOverwriteZeroBoundaries', false);
            parse(p, varargin{:});
            self.h = emgpu_mex('load_txt', p.Results.parameters_filename, p.Results.idxZeroBoundaries, p.Results.isOverwriteZeroBoundaries);
        end

        function delete(self)
            if self.h ~= 0, emgpu_mex('free', self.h); end
        end

        function [out_inits, out_events, out_samples, out_EME] = sample(self, n_samples, sample_time, varargin)
            p = inputParser;
            addParameter(p, 'seed', nan, @isnumeric);
            addParameter(p, 'isQuantize500', false, @islogical);
            addParameter(p, 'layers', [], @isnumeric);
            parse(p, varargin{:});
            seed = p.Results.seed;
            if isnan(seed), seed = randi(2^31 - 1); end   % no global stream on the device: derive a key
            idx = @(name) max([0 find(strcmp(self.labels_initial, name))]);
            idxL = idx('"L"'); idxV = idx('"v"'); idxDV = idx('"\dot v"'); idxDH = idx('"\dot h"'); idxDPsi = idx('"\dot \psi"');
            if idxDV == 0 || idxDH == 0 || idxDPsi == 0
                error('dynvar:empty', 'Model does not have a dynamic variable for either acceleration, vertical rate, or turn rate');
            end
            emgpu_mex('set_prior', self.h, self.prior);
            st = nan(1, self.n_initial);
            for ii = 1:self.n_initial, if ~isempty(self.start{ii}), st(ii) = self.start{ii}; end, end
            emgpu_mex('set_start', self.h, st);
            [out_inits, ev_count, E] = emgpu_mex('sample_uncor', self.h, n_samples, sample_time, seed, 0, double(p.Results.isQuantize500), ...
                                                 idxL, idxV, idxDH, p.Results.layers, 512);
            s = struct('temporal_map', self.temporal_map);
            out_events = cell(n_samples, 1); out_samples = cell(n_samples, 1); out_EME(n_samples, 1) = EncounterModelEvents;
            idxEME = [find(s.temporal_map(:, 1) == idxDH), find(s.temporal_map(:, 1) == idxDPsi), find(s.temporal_map(:, 1) == idxDV)] + 1;
            for ii = 1:n_samples
                events = E(1:ev_count(ii), :, ii);
                out_events{ii} = events;
                out_samples{ii} = events2samples(out_inits(ii, :), events);          % unchanged reference helpers
                controls = events2controls(out_inits(ii, :), events, s);
                controls = controls(:, [1 idxEME]);
                controls(:, 2) = controls(:, 2) / 60; controls(:, 3) = deg2rad(controls(:, 3)); controls(:, 4) = controls(:, 4) * 1.68780972222222;
                out_EME(ii) = EncounterModelEvents('event', controls);
            end
        end
    end
end
