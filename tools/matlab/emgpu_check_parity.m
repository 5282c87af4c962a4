function ok = emgpu_check_parity(repo_dir)
% EMGPU_CHECK_PARITY  Pin the emgpu CPU oracle (and through it the HIP kernels) against REAL MATLAB output.
%
%   Run once on a machine that has MATLAB and a checkout of Airspace-Encounter-Models/em-model-manned-bayes
%   (startup_bayes executed, AEM_DIR_BAYES set):
%
%       >> addpath('<emgpu repo>/tools/matlab'); emgpu_check_parity('<emgpu repo>')
%
%   Neither MATLAB nor Octave exists in the image this repository was built in, so the CPU oracle
%   (oracle/em_oracle.c) that every GPU parity test compares against is pinned only by a second
%   independent restatement, known answers and statistics -- "parity unpinned" in DESIGN.md section 8.
%   This function is the missing pin: it runs the REFERENCE code and compares it with the oracle's
%   MT19937-mode answers committed as text under tests/golden/matlab/ (written by
%   tests/golden/make_matlab_goldens.py; rng(seed,'twister') is the stream the oracle replays).
%
%   Checks (each prints PASS / FAIL with the first difference):
%     1. config 1: mdl = UncorEncounterModel(uncor_1200code_v2p1); mdl.sample(100, 120, 'seed', 1)
%        (UncorEncounterModel.m:192-313; RUN_uncor.m:43-50 shape) -- out_inits and every out_events{i}.
%     2. dependent branch: 40 successive dbn_hierarchical_sample calls on em_read(uncor_1200code_v1.txt)
%        after rng(7,'twister') (dbn_sample.m:65-93; order_transition is not the identity).
%     3. correlated model through em_read: 25 calls on cor_v1.txt after rng(11,'twister')
%        (16 initial / 4 dynamic variables; order_initial is not the identity; r_transition from the file).
%     4. bn_sort (toposort 'stable', bn_sort.m:17-20) on every shipped graph that is not upper-triangular.
%     5. sind / cosd as restated for createEncounter.m (report only: max difference in ulps).
%     6. the terminal geometry network: CorTerminalModel('srcData', s).sample(500, 'seed', 1) for s = terminalradar, opensky, without a
%        start and with mdl.start = {2, 1, 3, [], ...} (@CorTerminalModel/sample.m:29-77, RUN_terminal.m:36-39; acType GENERIC / GENERIC).
%   Discrete fields (dt, var, bins) must be equal; dediscretised doubles must be bit-equal (same IEEE
%   operations in the same order: a + (b - a) * rand), reported as max relative difference otherwise.
%
%   Returns true when checks 1-4 and 6 pass.  Please attach the printed output to an issue either way.

    if nargin < 1, repo_dir = fileparts(fileparts(fileparts(mfilename('fullpath')))); end
    gold = fullfile(repo_dir, 'tests', 'golden', 'matlab');
    mdir = fullfile(getenv('AEM_DIR_BAYES'), 'model');
    assert(isfolder(mdir), 'AEM_DIR_BAYES is not set to a checkout of em-model-manned-bayes');
    ok = true;

    %% 1. config 1 -----------------------------------------------------------------------------------------
    mdl = UncorEncounterModel('parameters_filename', fullfile(mdir, 'uncor_1200code_v2p1.txt'));
    [out_inits, out_events] = mdl.sample(100, 120, 'seed', 1);
    ok = compare('1 config1 UncorEncounterModel.sample(100,120,''seed'',1)', out_inits, out_events, ...
                 readmatrix(fullfile(gold, 'config1_inits.csv'), 'CommentStyle', '%'), ...
                 readmatrix(fullfile(gold, 'config1_events.csv'), 'CommentStyle', '%')) && ok;

    %% 2, 3. plain dbn_hierarchical_sample loops ------------------------------------------------------------
    cases = {'uncor_1200code_v1', 40, 60, 7; 'cor_v1', 25, 50, 11};
    for c = 1:size(cases, 1)
        [name, n, T, seed] = cases{c, :};
        parms = em_read(fullfile(mdir, [name '.txt']));
        a_i = bn_dirichlet_prior(parms.N_initial, 0);
        a_t = bn_dirichlet_prior(parms.N_transition, 0);
        rng(seed, 'twister');
        inits = zeros(n, parms.n_initial); events = cell(n, 1);
        for ii = 1:n
            [initial, ev] = dbn_hierarchical_sample(parms, a_i, a_t, T, parms.boundaries, parms.zero_bins, parms.resample_rates);
            inits(ii, :) = initial(:)'; events{ii} = ev;
        end
        ok = compare(sprintf('%d dbn_hierarchical_sample x%d on %s after rng(%d)', c + 1, n, name, seed), inits, events, ...
                     readmatrix(fullfile(gold, ['hier_' name '_inits.csv']), 'CommentStyle', '%'), ...
                     readmatrix(fullfile(gold, ['hier_' name '_events.csv']), 'CommentStyle', '%')) && ok;
    end

    %% 4. bn_sort -------------------------------------------------------------------------------------------
    fid = fopen(fullfile(gold, 'bn_sort_orders.csv')); fgetl(fid); good = true;
    while true
        ln = fgetl(fid); if ~ischar(ln), break; end
        parts = strsplit(ln, ','); want = str2double(parts(3:end)); want = want(want > 0);
        parms = em_read(fullfile(mdir, [parts{1} '.txt']));
        if str2double(parts{2}) == 1, G = parms.G_initial; else, G = parms.G_transition; end
        got = bn_sort(G);
        if ~isequal(got(:)', want(:)')
            good = false;
            fprintf('   %s graph %s: MATLAB %s, emgpu %s\n', parts{1}, parts{2}, mat2str(got(:)'), mat2str(want(:)'));
        end
    end
    fclose(fid);
    report('4 bn_sort on the non-upper-triangular graphs', good); ok = ok && good;

    %% 5. trig (report only) --------------------------------------------------------------------------------
    tt = readmatrix(fullfile(gold, 'trig_table.csv'), 'CommentStyle', '%');
    if ~isempty(tt)
        ds = abs(sind(tt(:, 1)) - tt(:, 2)) ./ eps(max(abs(tt(:, 2)), realmin));
        dc = abs(cosd(tt(:, 1)) - tt(:, 3)) ./ eps(max(abs(tt(:, 3)), realmin));
        fprintf('5 sind/cosd restatement: max difference %.1f / %.1f ulp over %d angles (0 = bit-equal)\n', max(ds), max(dc), size(tt, 1));
    end
    %% 6. terminal geometry network ---------------------------------------------------------------------------
    srcs = {'terminalradar', 'opensky'};
    for c = 1:numel(srcs)
        for withStart = [false true]
            mdl = CorTerminalModel('srcData', srcs{c});
            tag = '';
            if withStart
                st = cell(1, mdl.n_initial); st(1:3) = {2, 1, 3};
                mdl.start = st; tag = '_start';
            end
            outInits = mdl.sample(500, 'seed', 1);
            g = readmatrix(fullfile(gold, ['geom_' srcs{c} tag '_inits.csv']), 'CommentStyle', '%');
            good = isequal(size(outInits), size(g)) && isequal(outInits, g);
            msg = '';
            if ~good && isequal(size(outInits), size(g))
                [r, cc] = find(outInits ~= g, 1);
                msg = sprintf('outInits(%d,%d): MATLAB %.17g, emgpu %.17g; max rel diff %.3g', r, cc, outInits(r, cc), g(r, cc), ...
                              max(abs(outInits(:) - g(:)) ./ max(abs(g(:)), realmin)));
            end
            report(sprintf('6 CorTerminalModel(''srcData'',''%s'')%s.sample(500,''seed'',1)', srcs{c}, tag), good, msg); ok = ok && good;
        end
    end
    if ok, fprintf('ALL PASS: the emgpu oracle reproduces this MATLAB bit for bit on the checked paths.\n');
    else, fprintf('FAILURES above: please report them with this output.\n'); end
end

function ok = compare(tag, inits, events, g_inits, g_events)
    ok = true; msg = '';
    if ~isequal(size(inits), size(g_inits))
        ok = false; msg = sprintf('initial size %s vs %s', mat2str(size(inits)), mat2str(size(g_inits)));
    elseif ~isequal(inits, g_inits)
        [r, c] = find(inits ~= g_inits, 1);
        ok = false; msg = sprintf('initial(%d,%d): MATLAB %.17g, emgpu %.17g; max rel diff %.3g', r, c, inits(r, c), g_inits(r, c), ...
                                  max(abs(inits(:) - g_inits(:)) ./ max(abs(g_inits(:)), realmin)));
    end
    for ii = 1:numel(events)
        if ~ok, break; end
        g = g_events(g_events(:, 1) == ii, 2:4);
        e = events{ii};
        if ~isequal(size(e), size(g))
            ok = false; msg = sprintf('sample %d: %d event rows vs %d', ii, size(e, 1), size(g, 1));
        elseif ~isequal(e(:, 1:2), g(:, 1:2))
            r = find(any(e(:, 1:2) ~= g(:, 1:2), 2), 1);
            ok = false; msg = sprintf('sample %d row %d: [dt var] MATLAB %s, emgpu %s', ii, r, mat2str(e(r, 1:2)), mat2str(g(r, 1:2)));
        elseif ~isequal(e(:, 3), g(:, 3))
            r = find(e(:, 3) ~= g(:, 3), 1);
            ok = false; msg = sprintf('sample %d row %d: value MATLAB %.17g, emgpu %.17g', ii, r, e(r, 3), g(r, 3));
        end
    end
    report(tag, ok, msg);
end

function report(tag, ok, msg)
    if nargin < 3, msg = ''; end
    if ok, fprintf('PASS  %s\n', tag); else, fprintf('FAIL  %s  -- %s\n', tag, msg); end
end
