classdef CorTerminalModelGPU < CorTerminalModel
    % CorTerminalModelGPU  Drop-in subclass of the reference's CorTerminalModel whose geometry draw (.sample,
    % @CorTerminalModel/sample.m:1-82) and trajectory propagation (createEncounter.m:93-265, all four tracks of many
    % encounters in one launch) run on an MI355X through libemgpu.  UNTESTED (no MATLAB in the build image).
    %
    %   mdl = CorTerminalModelGPU('srcData', 'terminalradar');
    %   [outInits, outSamples] = mdl.sample(1000, 'seed', 1);
    %   traj = mdl.createEncounterGPU(outSamples, 120, 'seed', 1);      % traj{k}(1:2): ownship / intruder structs
    %
    % em-core's local_smooth of speed and altitude (createEncounter.m:88-89) is not applied.

    methods
        function self = CorTerminalModelGPU(varargin)
            self@CorTerminalModel(varargin{:});
        end

        function [outInits, outSamples] = sample(self, nSamples, varargin)
            p = inputParser;
            addParameter(p, 'seed', nan, @isnumeric);
            parse(p, varargin{:});
            seed = p.Results.seed;
            if isnan(seed) || isempty(seed), seed = randi(2^31 - 1); end
            s = self.struct;
            h = emgpu_mex('from_struct', s); cleanup = onCleanup(@() emgpu_mex('free', h));
            emgpu_mex('set_alpha', h, s.dirichlet_initial, {});
            idxOwn = find(strcmp(self.labels_initial, '"own_speed"')); idxInt = find(strcmp(self.labels_initial, '"int_speed"'));   % sample.m:64-70
            d1 = self.dynLimits1; d2 = self.dynLimits2;
            bs = self.bounds_sample; if all(isinf(bs(:))), bs = []; end
            outInits = emgpu_mex('geom_sample', h, nSamples, seed, 0, bs, idxOwn, idxInt, [d1.minVel_ft_s d1.maxVel_ft_s], [d2.minVel_ft_s d2.maxVel_ft_s]);
            names = matlab.lang.makeValidName(erase(self.labels_initial, '"'));
            outSamples = cell2struct(num2cell(outInits), names, 2);          % sample.m:72-76
        end

        function [out_results, gen_time_s] = track(self, nSamples, varargin)
            % Same signature and outputs as @CorTerminalModel/track.m; the rejection loop of track.m:55-150 runs in device rounds
            % (emgpu_mex('track_terminal')).  em-core's computeVerticalRate / computeHeadingRate are forward differences there.
            p = inputParser; p.KeepUnmatched = true;
            addParameter(p, 'initialSeed', nan, @isnumeric);
            addParameter(p, 'firstID', 1, @isnumeric);
            addParameter(p, 'minEncTime_s', 30, @isnumeric);
            addParameter(p, 'thresDist_ft', 2.5 * 6076, @isnumeric);
            addParameter(p, 'thresAltLow_ft', 750, @isnumeric);
            addParameter(p, 'thresVertRate_ft_s', 300 / 60, @isnumeric);
            parse(p, varargin{:});
            seed = p.Results.initialSeed;
            if isnan(seed) || isempty(seed), seed = randi(2^31 - 1); end
            tStart = tic;
            [hGeom, handles] = self.nativeModels(); cleanup = onCleanup(@() arrayfun(@(x) emgpu_mex('free', x), [hGeom handles]));
            lab = @(name) find(strcmp(self.labels_initial, ['"' name '"']));
            f = {'distance', 'bearing', 'alt', 'speed', 'heading', 'intent'};
            idx12 = [cellfun(@(x) lab(['own_' x]), f), cellfun(@(x) lab(['int_' x]), f)];
            d1 = self.dynLimits1; d2 = self.dynLimits2;
            dl = [d1.minVel_ft_s d1.maxVel_ft_s d1.maxTurnRate_deg_s d1.maxAltitude_ft d1.maxVertRate_ft_s; ...
                  d2.minVel_ft_s d2.maxVel_ft_s d2.maxTurnRate_deg_s d2.maxAltitude_ft d2.maxVertRate_ft_s]';
            bs = self.bounds_sample; if all(isinf(bs(:))), bs = []; end
            [S, TR, LEN, META] = emgpu_mex('track_terminal', hGeom, handles, nSamples, seed, 0, dl, ...
                                           [d1.maxCumTurn_deg d2.maxCumTurn_deg d1.pitch_deg d2.pitch_deg], ...
                                           [p.Results.minEncTime_s p.Results.thresDist_ft p.Results.thresAltLow_ft p.Results.thresVertRate_ft_s], idx12, bs, 5000);
            names = matlab.lang.makeValidName(erase(self.labels_initial, '"'));
            out_results = struct('sample', [], 'traj', []);
            fields = {'t_s', 'x_nm', 'y_nm', 'z_ft', 'heading_deg', 'v_ft_s'};
            for ii = 1:nSamples
                sample_geo = cell2struct(num2cell(S(:, ii)), names(:), 1);
                for a = 1:2
                    for k = 1:6, traj(a).(fields{k}) = squeeze(TR(k, 1:LEN(a, ii), a, ii))'; end %#ok<AGROW>
                end
                [trajFrmt, tcpa_adjusted] = self.reformatTrajFiles(traj, META(1, ii));             % the reference's own helper
                if tcpa_adjusted == 0, tcpa_adjusted = 1; end                                       % track.m:160-163
                out_results(ii).sample = sample_geo; out_results(ii).traj = trajFrmt;
                out_results(ii).sample.id = ii + (p.Results.firstID - 1);
                out_results(ii).sample.tcpa = tcpa_adjusted;
                out_results(ii).sample.hmd_ft = META(2, ii); out_results(ii).sample.vmd_ft = META(3, ii);
                out_results(ii).sample.nmac = abs(META(2, ii)) < 500 & abs(META(3, ii)) < 100;
            end
            gen_time_s = repmat(toc(tStart) / nSamples, nSamples, 1);
        end

        function [hGeom, handles] = nativeModels(self)
            s = self.struct;
            hGeom = emgpu_mex('from_struct', s); emgpu_mex('set_alpha', hGeom, s.dirichlet_initial, {});
            mdls = {self.mdlFwd1_1, self.mdlBck1_1, self.mdlFwd1_2, self.mdlBck1_2, self.mdlFwd2_1, self.mdlBck2_1, ...
                    self.mdlFwd2_2, self.mdlBck2_2, self.mdlFwd2_3, self.mdlBck2_3};       % CorTerminalModel.m:84-100
            handles = zeros(1, 10, 'uint64');
            for k = 1:10
                s = mdls{k}.struct;
                s.r_transition = cellfun(@(N) size(N, 1), s.N_transition);                  % from the tables, not EncounterModel.m:313-323
                s.r_transition(1:s.n_initial) = s.r_initial;
                handles(k) = emgpu_mex('from_struct', s);
                a = setTransitionPriors(s.G_transition, s.r_transition, s.temporal_map, 1);  % createEncounter.m:128-129
                emgpu_mex('set_alpha', handles(k), s.dirichlet_initial, a);
            end
        end

        function traj = createEncounterGPU(self, sample_geo, tmax_s, varargin)
            p = inputParser;
            addParameter(p, 'seed', nan, @isnumeric);
            parse(p, varargin{:});
            seed = p.Results.seed;
            if isnan(seed) || isempty(seed), seed = randi(2^31 - 1); end
            [hGeom, handles] = self.nativeModels(); emgpu_mex('free', hGeom);
            cleanup = onCleanup(@() arrayfun(@(x) emgpu_mex('free', x), handles));
            n = numel(sample_geo);
            geo = zeros(12, n); model_of = zeros(4, n);
            for e = 1:n                                                                       % createEncounter.m:21-49
                g = sample_geo(e);
                geo(:, e) = [g.own_distance * cosd(g.own_bearing); g.own_distance * sind(g.own_bearing); g.own_alt; g.own_speed; g.own_heading; g.own_intent; ...
                             g.int_distance * cosd(g.int_bearing); g.int_distance * sind(g.int_bearing); g.int_alt; g.int_speed; g.int_heading; g.int_intent];
                model_of(:, e) = [2 * (g.own_intent - 1); 2 * (g.own_intent - 1) + 1; 4 + 2 * (g.int_intent - 1); 4 + 2 * (g.int_intent - 1) + 1];
            end
            d1 = self.dynLimits1; d2 = self.dynLimits2;
            dl = [d1.minVel_ft_s d1.maxVel_ft_s d1.maxTurnRate_deg_s d1.maxAltitude_ft d1.maxVertRate_ft_s; ...
                  d2.minVel_ft_s d2.maxVel_ft_s d2.maxTurnRate_deg_s d2.maxAltitude_ft d2.maxVertRate_ft_s]';
            [out, rows] = emgpu_mex('propagate_terminal', handles, geo, model_of, seed, 0, tmax_s, dl);
            fields = {'t_s', 'x_nm', 'y_nm', 'z_ft', 'heading_deg', 'v_ft_s'};
            traj = cell(n, 1);
            for e = 1:n
                for a = 1:2
                    lf = 4 * (e - 1) + 2 * (a - 1) + 1; lb = lf + 1;
                    fwd = squeeze(out(lf, 1:rows(lf), :)); bck = squeeze(out(lb, 2:rows(lb), :));   % bck(1, 2:end), createEncounter.m:77
                    both = sortrows([fwd; bck], 1);                                            % :81-84
                    for f = 1:6, traj{e}(a).(fields{f}) = both(:, f)'; end
                end
            end
        end
    end
end
