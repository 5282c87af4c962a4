/*
 * emgpu_mex.c -- MATLAB gateway to libemgpu.so (include/emgpu.h).
 *
 * NOT RUN: neither MATLAB nor mex.h exists in the build image (tests only type-check this file against a
 * declaration-only stand-in, tests/stubs/mex.h); this is the binding a maintainer compiles on a machine
 * that has both:
 *     mex -R2018a emgpu_mex.c -I<repo>/include -L<repo>/em_model_manned_bayes_amd -lemgpu
 *
 * One entry point, dispatched on a command string (MATLAB calls are single threaded).  Handles are uint64 scalars.
 *
 *   model
 *     h = emgpu_mex('load_txt', filename, idxZeroBoundaries, isOverwriteZeroBoundaries)            em_read.m:1,41-42
 *     s = emgpu_mex('em_read', filename, idxZeroBoundaries, isOverwriteZeroBoundaries)             the struct em_read.m returns
 *     h = emgpu_mex('from_struct', parms)          em_read's struct or struct(EncounterModel): dbn_sample.m:25-33 contract
 *         emgpu_mex('set_prior', h, prior)         numeric or 'dbe'                                EncounterModel.m:194-203
 *         emgpu_mex('set_alpha', h, dirichlet_initial, dirichlet_transition)   cell arrays as passed to dbn_sample.m:1
 *         emgpu_mex('set_start', h, start)         double vector, 0 / NaN = unset                  bn_sample.m:44-50
 *         emgpu_mex('save_bin', h, filename);  h = emgpu_mex('load_bin', filename)   the binary model cache (parsed model + compiled plan)
 *         emgpu_mex('free', h)
 *   devices
 *     n = emgpu_mex('device_count');   emgpu_mex('use_devices', [0 1 ...])   later sample_uncor calls are split over them
 *   sampling
 *     S = emgpu_mex('bn_sample', h, num_samples, seed, first_index)                                bn_sample.m:1 (bins)
 *     [initial, ev_count, events] = emgpu_mex('sample_uncor', h, n, T, seed, first_index, flags, idxL, idxV, idxDH, layers, event_cap)
 *         flags: EMGPU_FLAG_* ; idxV = idxDH = 0 => dbn_hierarchical_sample.m:1 ; with NO_RESAMPLE|NO_DEDISC|NO_TERMINATOR => dbn_sample.m:1
 *         initial n x n_initial double; ev_count n x 1; events event_cap x 3 x n rows [dt var value]
 *     [outInits, attempts, logWeight] = emgpu_mex('geom_sample', h, n, seed, first_index, bounds_sample, idxOwnSpeed, idxIntSpeed, lim1, lim2, startGrid)
 *         startGrid (optional): n x n_initial presets, one row per sample (InitStartTerminal.m:57-90), drawn in ONE launch; logWeight n x 1
 *                                                                                                  @CorTerminalModel/sample.m:29-77
 *     [out, rows] = emgpu_mex('propagate_terminal', handles, geo, model_of, seed, first_index, tmax_s, dyn_limits)
 *         handles 1 x 10 uint64 (stay prior applied); geo 12 x n; model_of 4 x n (0-based); dyn_limits 5 x 2
 *         out 4n x cap x 6 double (lane, second, [t_s x_nm y_nm z_ft heading_deg v_ft_s]); rows 4n x 1     createEncounter.m:93-265
 *     [tracks, limits, attempts] = emgpu_mex('track_uncor', h, n, T, seed, first_index, isQuantize500, isRotorcraft, idx7, stride, max_track_attempts)
 *         tracks 8 x S x n [time north east up speed phi theta psi]                                UncorEncounterModel.m:318-471
 *     [sample, traj, len, meta, attempts] = emgpu_mex('track_terminal', hGeom, handles, n, seed, first_index, dyn_limits, cumturn_pitch, thresholds, idx12, bounds_sample, max_track_attempts)
 *         CorTerminalModel.track (track.m:45-150) in device rounds: dyn_limits 5 x 2, cumturn_pitch = [maxCumTurn1 maxCumTurn2 pitch1 pitch2],
 *         thresholds = [minEncTime_s thresDist_ft thresAltLow_ft thresVertRate_ft_s], idx12 = variable ids of own / int {distance bearing alt
 *         speed heading intent}; sample n_i x n, traj 6 x cap2 x 2 x n, len 2 x n, meta 4 x n [tcpa_s hmd_ft vmd_ft enc_time_s]
 *     [xyz, flags, vminmax] = emgpu_mex('sample2track', alt0, speed0, updates, ur, min_speed, max_speed)   sample2track.m:182-243
 *
 * Errors keep the reference's identifiers where it has them: prior:notdbe / prior:unknown (bn_dirichlet_prior.m:28,37);
 * the free-text errors of bn_sample.m:47, bn_sort.m:23 and em_read.m:104 keep their text under emgpu:preset / emgpu:sort /
 * emgpu:parse; emgpu:eventcap and emgpu:rejectcap are the two caps the reference does not have.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "mex.h"
#include "emgpu.h"

#define MAX_DEV 16
#define EMGPU_MEX_MAX_VARS 32 /* initial variables a model may have on this side (the library's own limit is lower) */
static emgpu_ctx *g_ctx[MAX_DEV];
static int g_nctx = 0;

static void shutdown_all(void) {
    for (int d = 0; d < g_nctx; d++) { emgpu_ctx_free(g_ctx[d]); g_ctx[d] = NULL; }
    g_nctx = 0;
}

static void check(int rc) {
    if (rc >= 0) return;
    const char *id = "emgpu:error";
    const char *msg = emgpu_last_error();
    switch (rc) {
    case EMGPU_ERR_PRIOR: id = strstr(msg, "Second argument") ? "prior:unknown" : "prior:notdbe"; break;
    case EMGPU_ERR_PRESET: id = "emgpu:preset"; break;      /* 'Attempt to preset a dependent variable' */
    case EMGPU_ERR_SORT: id = "emgpu:sort"; break;          /* 'Network could not be hierarchically sorted' */
    case EMGPU_ERR_PARSE: id = "emgpu:parse"; break;        /* 'Unknown field: ...' */
    case EMGPU_ERR_IO: id = "emgpu:io"; break;
    case EMGPU_ERR_EVENT_CAP: id = "emgpu:eventcap"; break; /* the .m layer doubles event_cap and retries */
    case EMGPU_ERR_REJECT_CAP: id = "emgpu:rejectcap"; break;
    case EMGPU_ERR_NO_DEVICE: id = "emgpu:nodevice"; break;
    case EMGPU_ERR_UNSUPPORTED: id = "emgpu:unsupported"; break;
    case EMGPU_ERR_ARG: id = strstr(msg, "dynvar:empty") ? "dynvar:empty" : "emgpu:arg"; break;
    default: break;
    }
    mexErrMsgIdAndTxt(id, "%s", msg);
}

static void need(int nrhs, int n, const char *usage) {
    if (nrhs < n) mexErrMsgIdAndTxt("emgpu:usage", "%s", usage);
}

static emgpu_model *handle_of(const mxArray *a) {
    if (mxGetNumberOfElements(a) < 1) mexErrMsgIdAndTxt("emgpu:usage", "empty model handle");
    return (emgpu_model *)(uintptr_t)(*(uint64_t *)mxGetData(a));
}

static void get_string(const mxArray *a, char *buf, size_t n) {
    if (!mxIsChar(a) || mxGetString(a, buf, n)) mexErrMsgIdAndTxt("emgpu:usage", "expected a character vector of at most %d characters", (int)n - 1);
}

static emgpu_ctx *ctx0(void) {
    if (g_nctx == 0) { check(emgpu_ctx_create(0, &g_ctx[0])); g_nctx = 1; mexAtExit(shutdown_all); }
    return g_ctx[0];
}

static int32_t n_initial_of(emgpu_model *m) { emgpu_model_info_t info; check(emgpu_model_info(m, &info)); return info.n_initial; }

/* cell {i} of r_i x q_i doubles -> concatenated column-major (em_read.m:191-198) */
static double *concat_cells(const mxArray *cell, int first, int last, int64_t *total) {
    int64_t n = 0;
    for (int i = first; i < last; i++) { const mxArray *c = mxGetCell(cell, i); if (c) n += (int64_t)mxGetNumberOfElements(c); }
    double *out = (double *)mxMalloc(sizeof(double) * (size_t)(n ? n : 1));
    int64_t o = 0;
    for (int i = first; i < last; i++) {
        const mxArray *c = mxGetCell(cell, i);
        if (!c) continue;
        const size_t k = mxGetNumberOfElements(c);
        memcpy(out + o, mxGetPr(c), k * sizeof(double));
        o += (int64_t)k;
    }
    *total = n;
    return out;
}

/* G (n x n logical or double, MATLAB column-major [parent][child] = G(parent, child)) -> row-major uint8 */
static uint8_t *graph_rows(const mxArray *G, int n) {
    uint8_t *out = (uint8_t *)mxMalloc((size_t)n * n + 1);
    for (int p = 0; p < n; p++)
        for (int c = 0; c < n; c++)
            out[(size_t)p * n + c] = mxIsLogical(G) ? (uint8_t)mxGetLogicals(G)[(size_t)c * n + p] : (uint8_t)(mxGetPr(G)[(size_t)c * n + p] != 0);
    return out;
}

static mxArray *cells_from_model(emgpu_model *m, int field, int first, int last, int total, const int32_t *r) {
    mxArray *cell = mxCreateCellMatrix((mwSize)total, 1);
    for (int v = first; v < last; v++) {
        const int64_t cnt = emgpu_model_get_f64(m, field, v + 1, NULL, 0);
        if (cnt <= 0) continue;
        mxArray *a = mxCreateDoubleMatrix((mwSize)r[v], (mwSize)(cnt / r[v]), mxREAL);
        check((int)emgpu_model_get_f64(m, field, v + 1, mxGetPr(a), cnt));
        mxSetCell(cell, (mwSize)v, a);
    }
    return cell;
}

static mxArray *labels_cell(emgpu_model *m, int field) {
    const int64_t len = emgpu_model_get_text(m, field, NULL, 0);
    char *buf = (char *)mxMalloc((size_t)(len > 0 ? len : 1) + 1);
    buf[0] = 0;
    if (len > 0) check((int)emgpu_model_get_text(m, field, buf, len));
    int n = 0;
    if (buf[0]) { n = 1; for (char *p = buf; *p; p++) n += (*p == '\n'); }
    mxArray *cell = mxCreateCellMatrix(1, (mwSize)n);
    char *p = buf;
    for (int i = 0; i < n; i++) {
        char *e = strchr(p, '\n');
        if (e) *e = 0;
        mxSetCell(cell, (mwSize)i, mxCreateString(p));
        p = e ? e + 1 : p;
    }
    mxFree(buf);
    return cell;
}

static void events_out(mxArray *plhs[], int nlhs, size_t n, size_t ni, size_t cap, const float *iv, const uint32_t *ec, const emgpu_event *ev) {
    plhs[0] = mxCreateDoubleMatrix((mwSize)n, (mwSize)ni, mxREAL);
    for (size_t v = 0; v < ni; v++) for (size_t i = 0; i < n; i++) mxGetPr(plhs[0])[v * n + i] = iv[v * n + i];
    if (nlhs < 2) return;
    plhs[1] = mxCreateDoubleMatrix((mwSize)n, 1, mxREAL);
    for (size_t i = 0; i < n; i++) mxGetPr(plhs[1])[i] = ec[i];
    if (nlhs < 3) return;
    mwSize dims[3];
    dims[0] = (mwSize)cap; dims[1] = 3; dims[2] = (mwSize)n;
    plhs[2] = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxREAL);
    double *E = mxGetPr(plhs[2]);
    for (size_t i = 0; i < n; i++)
        for (size_t e = 0; e < ec[i] && e < cap; e++) {
            const emgpu_event *r = &ev[i * cap + e];
            E[i * cap * 3 + e] = r->dt; E[i * cap * 3 + cap + e] = r->var; E[i * cap * 3 + 2 * cap + e] = r->value;
        }
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    char cmd[64];
    if (nrhs < 1) mexErrMsgIdAndTxt("emgpu:usage", "first argument must be a command string");
    get_string(prhs[0], cmd, sizeof cmd);

    if (!strcmp(cmd, "load_txt") || !strcmp(cmd, "em_read")) {
        need(nrhs, 2, "emgpu_mex('load_txt' | 'em_read', filename, idxZeroBoundaries, isOverwriteZeroBoundaries)");
        char path[4096];
        get_string(prhs[1], path, sizeof path);
        int32_t idx[EMGPU_MEX_MAX_VARS];
        int32_t n_idx = 0;
        if (nrhs > 2) {
            n_idx = (int32_t)mxGetNumberOfElements(prhs[2]);
            if (n_idx > EMGPU_MEX_MAX_VARS) mexErrMsgIdAndTxt("emgpu:usage", "idxZeroBoundaries has more than %d entries", EMGPU_MEX_MAX_VARS);
            for (int i = 0; i < n_idx; i++) idx[i] = (int32_t)mxGetPr(prhs[2])[i];
        }
        emgpu_model *m = NULL;
        check(emgpu_model_load_txt(path, n_idx ? idx : NULL, n_idx, nrhs > 3 && mxIsLogicalScalarTrue(prhs[3]), &m));
        if (!strcmp(cmd, "load_txt")) {
            plhs[0] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
            *(uint64_t *)mxGetData(plhs[0]) = (uint64_t)(uintptr_t)m;
            return;
        }
        /* the struct of em_read.m:47-141 */
        emgpu_model_info_t info;
        check(emgpu_model_info(m, &info));
        const int ni = info.n_initial, nt = info.n_transition;
        static const char *fields[] = {"labels_initial", "n_initial", "G_initial", "order_initial", "r_initial", "N_initial", "labels_transition",
                                       "n_transition", "G_transition", "order_transition", "r_transition", "N_transition", "boundaries",
                                       "resample_rates", "temporal_map", "zero_bins", "bounds_initial", "cutpoints_initial"};
        mxArray *s = mxCreateStructMatrix(1, 1, 18, fields);
        int32_t r[2 * EMGPU_MEX_MAX_VARS], tmp[4 * EMGPU_MEX_MAX_VARS * EMGPU_MEX_MAX_VARS];
        mxSetField(s, 0, "labels_initial", labels_cell(m, EMGPU_F_LABELS_INITIAL));
        mxSetField(s, 0, "n_initial", mxCreateDoubleScalar(ni));
        for (int pass = 0; pass < (nt ? 2 : 1); pass++) {
            const int n = pass ? nt : ni;
            check((int)emgpu_model_get_i32(m, pass ? EMGPU_F_G_TRANSITION : EMGPU_F_G_INITIAL, tmp, 4 * EMGPU_MEX_MAX_VARS * EMGPU_MEX_MAX_VARS));
            mxArray *G = mxCreateLogicalMatrix((mwSize)n, (mwSize)n);
            for (int p = 0; p < n; p++) for (int c = 0; c < n; c++) mxGetLogicals(G)[(size_t)c * n + p] = tmp[(size_t)p * n + c] != 0;
            mxSetField(s, 0, pass ? "G_transition" : "G_initial", G);
            check((int)emgpu_model_get_i32(m, pass ? EMGPU_F_ORDER_TRANSITION : EMGPU_F_ORDER_INITIAL, tmp, 2 * EMGPU_MEX_MAX_VARS));
            mxArray *o = mxCreateDoubleMatrix(1, (mwSize)n, mxREAL);
            for (int i = 0; i < n; i++) mxGetPr(o)[i] = tmp[i];
            mxSetField(s, 0, pass ? "order_transition" : "order_initial", o);
            check((int)emgpu_model_get_i32(m, pass ? EMGPU_F_R_TRANSITION : EMGPU_F_R_INITIAL, r, 2 * EMGPU_MEX_MAX_VARS));
            mxArray *rr = mxCreateDoubleMatrix((mwSize)n, 1, mxREAL);
            for (int i = 0; i < n; i++) mxGetPr(rr)[i] = r[i];
            mxSetField(s, 0, pass ? "r_transition" : "r_initial", rr);
            mxSetField(s, 0, pass ? "N_transition" : "N_initial",
                       cells_from_model(m, pass ? EMGPU_F_N_TRANSITION : EMGPU_F_N_INITIAL, pass ? ni : 0, n, n, r));
        }
        if (nt) {
            mxSetField(s, 0, "labels_transition", labels_cell(m, EMGPU_F_LABELS_TRANSITION));
            mxSetField(s, 0, "n_transition", mxCreateDoubleScalar(nt));
            const int nd = info.n_dyn;
            check((int)emgpu_model_get_i32(m, EMGPU_F_TEMPORAL_MAP, tmp, 2 * EMGPU_MEX_MAX_VARS));
            mxArray *tm = mxCreateDoubleMatrix((mwSize)nd, 2, mxREAL);
            for (int k = 0; k < nd; k++) { mxGetPr(tm)[k] = tmp[2 * k]; mxGetPr(tm)[nd + k] = tmp[2 * k + 1]; }
            mxSetField(s, 0, "temporal_map", tm);
        }
        check((int)emgpu_model_get_i32(m, EMGPU_F_R_INITIAL, r, 2 * EMGPU_MEX_MAX_VARS));
        check((int)emgpu_model_get_i32(m, EMGPU_F_ZERO_BINS, tmp, 2 * EMGPU_MEX_MAX_VARS));
        mxArray *bnd = mxCreateCellMatrix(1, (mwSize)ni), *zb = mxCreateCellMatrix(1, (mwSize)ni), *cut = mxCreateCellMatrix(1, (mwSize)ni);
        mxArray *bounds = mxCreateDoubleMatrix((mwSize)ni, 2, mxREAL);
        for (int v = 0; v < ni; v++) {
            const int64_t nb = emgpu_model_get_f64(m, EMGPU_F_BOUNDARIES, v + 1, NULL, 0);
            mxArray *b = mxCreateDoubleMatrix((mwSize)(nb > 0 ? nb : 0), nb > 0 ? 1 : 0, mxREAL);      /* '*' -> empty (em_read.m:97-99) */
            mxArray *c;
            if (nb > 0) {
                check((int)emgpu_model_get_f64(m, EMGPU_F_BOUNDARIES, v + 1, mxGetPr(b), nb));
                double lo = mxGetPr(b)[0], hi = mxGetPr(b)[0];
                for (int64_t i = 1; i < nb; i++) { lo = fmin(lo, mxGetPr(b)[i]); hi = fmax(hi, mxGetPr(b)[i]); }
                mxGetPr(bounds)[v] = lo; mxGetPr(bounds)[ni + v] = hi;                                 /* em_read.m:133-134 */
                c = mxCreateDoubleMatrix(1, (mwSize)(nb - 2), mxREAL);
                for (int64_t i = 1; i + 1 < nb; i++) mxGetPr(c)[i - 1] = mxGetPr(b)[i];
            } else {
                c = mxCreateDoubleMatrix(1, (mwSize)(r[v] - 1), mxREAL);                              /* 2:n, em_read.m:130-131 */
                for (int i = 0; i + 1 < r[v]; i++) mxGetPr(c)[i] = 2 + i;
            }
            mxSetCell(bnd, (mwSize)v, b);
            mxSetCell(cut, (mwSize)v, c);
            mxArray *z = mxCreateDoubleMatrix(tmp[v] ? 1 : 0, tmp[v] ? 1 : 0, mxREAL);
            if (tmp[v]) mxGetPr(z)[0] = tmp[v];
            mxSetCell(zb, (mwSize)v, z);
        }
        mxSetField(s, 0, "boundaries", bnd); mxSetField(s, 0, "zero_bins", zb);
        mxSetField(s, 0, "cutpoints_initial", cut); mxSetField(s, 0, "bounds_initial", bounds);
        mxArray *rates = mxCreateDoubleMatrix((mwSize)ni, 1, mxREAL);
        check((int)emgpu_model_get_f64(m, EMGPU_F_RESAMPLE_RATES, 0, mxGetPr(rates), ni));
        mxSetField(s, 0, "resample_rates", rates);
        emgpu_model_free(m);
        plhs[0] = s;
    } else if (!strcmp(cmd, "save_bin")) {
        need(nrhs, 3, "emgpu_mex('save_bin', h, filename)");
        char path[4096];
        get_string(prhs[2], path, sizeof path);
        check(emgpu_model_save_bin((const emgpu_model *)(uintptr_t)(*(uint64_t *)mxGetData(prhs[1])), path));
    } else if (!strcmp(cmd, "load_bin")) {
        need(nrhs, 2, "h = emgpu_mex('load_bin', filename)");
        char path[4096];
        get_string(prhs[1], path, sizeof path);
        emgpu_model *m = NULL;
        check(emgpu_model_load_bin(path, &m));
        plhs[0] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
        *(uint64_t *)mxGetData(plhs[0]) = (uint64_t)(uintptr_t)m;
    } else if (!strcmp(cmd, "from_struct")) {
        need(nrhs, 2, "h = emgpu_mex('from_struct', parms)");
        const mxArray *P = prhs[1];
        if (!mxIsStruct(P)) mexErrMsgIdAndTxt("emgpu:usage", "parms must be a struct (em_read's, or struct(EncounterModel))");
        const mxArray *Gi = mxGetField(P, 0, "G_initial"), *Ni = mxGetField(P, 0, "N_initial");
        if (!Gi || !Ni || !mxIsCell(Ni)) mexErrMsgIdAndTxt("emgpu:usage", "parms needs G_initial and the cell N_initial");
        const int ni = (int)mxGetM(Gi);
        const mxArray *Gt = mxGetField(P, 0, "G_transition"), *Nt = mxGetField(P, 0, "N_transition"), *rt = mxGetField(P, 0, "r_transition");
        const mxArray *tm = mxGetField(P, 0, "temporal_map"), *bnd = mxGetField(P, 0, "boundaries"), *rates = mxGetField(P, 0, "resample_rates");
        const mxArray *zb = mxGetField(P, 0, "zero_bins");
        const int nt = (Gt && !mxIsEmpty(Gt)) ? (int)mxGetM(Gt) : 0;
        if (ni < 1 || ni > EMGPU_MEX_MAX_VARS || nt > 2 * EMGPU_MEX_MAX_VARS) mexErrMsgIdAndTxt("emgpu:usage", "model size outside 1..%d initial variables", EMGPU_MEX_MAX_VARS);
        emgpu_model_desc d;
        memset(&d, 0, sizeof d);
        int32_t r_i[EMGPU_MEX_MAX_VARS], r_t[2 * EMGPU_MEX_MAX_VARS], tmap[2 * EMGPU_MEX_MAX_VARS], blen[EMGPU_MEX_MAX_VARS], zbin[EMGPU_MEX_MAX_VARS];
        for (int i = 0; i < ni; i++) { const mxArray *c = mxGetCell(Ni, i); r_i[i] = c ? (int32_t)mxGetM(c) : 0; }   /* r = rows of N{i} */
        d.n_initial = ni; d.G_initial = graph_rows(Gi, ni); d.r_initial = r_i;
        d.N_initial = concat_cells(Ni, 0, ni, &d.n_N_initial);
        if (nt) {
            if (!Nt || !mxIsCell(Nt) || !rt) mexErrMsgIdAndTxt("emgpu:usage", "parms needs N_transition and r_transition (dbn_sample.m:25-33)");
            for (int i = 0; i < nt; i++) r_t[i] = (int32_t)mxGetPr(rt)[i];
            d.n_transition = nt; d.G_transition = graph_rows(Gt, nt); d.r_transition = r_t;
            d.N_transition = concat_cells(Nt, ni, nt, &d.n_N_transition);                              /* em_read.m:92 */
            if (tm && !mxIsEmpty(tm)) {
                d.n_dyn = (int32_t)mxGetM(tm);
                if (d.n_dyn > EMGPU_MEX_MAX_VARS) mexErrMsgIdAndTxt("emgpu:usage", "temporal_map has too many rows");
                for (int k = 0; k < d.n_dyn; k++) { tmap[2 * k] = (int32_t)mxGetPr(tm)[k]; tmap[2 * k + 1] = (int32_t)mxGetPr(tm)[d.n_dyn + k]; }
                d.temporal_map = tmap;
            }
        }
        double *bflat = NULL;
        if (bnd && mxIsCell(bnd)) {
            int64_t tot = 0;
            bflat = concat_cells(bnd, 0, ni, &tot);
            for (int i = 0; i < ni; i++) { const mxArray *c = mxGetCell(bnd, i); blen[i] = c ? (int32_t)mxGetNumberOfElements(c) : 0; }
            d.boundaries = bflat; d.bnd_len = blen;
        }
        if (zb && mxIsCell(zb)) {
            for (int i = 0; i < ni; i++) { const mxArray *c = mxGetCell(zb, i); zbin[i] = (c && !mxIsEmpty(c)) ? (int32_t)mxGetPr(c)[0] : 0; }
            d.zero_bins = zbin;
        }
        if (rates && (int)mxGetNumberOfElements(rates) == ni) d.resample_rates = mxGetPr(rates);
        emgpu_model *m = NULL;
        const int rc = emgpu_model_from_arrays(&d, &m);
        mxFree((void *)d.G_initial); mxFree((void *)d.N_initial);
        if (nt) { mxFree((void *)d.G_transition); mxFree((void *)d.N_transition); }
        if (bflat) mxFree(bflat);
        check(rc);
        const mxArray *st = mxGetField(P, 0, "start");
        if (st && mxIsCell(st) && (int)mxGetNumberOfElements(st) == ni) {
            int32_t s32[EMGPU_MEX_MAX_VARS];
            for (int i = 0; i < ni; i++) {
                const mxArray *c = mxGetCell(st, i);
                const double v = (c && !mxIsEmpty(c)) ? mxGetPr(c)[0] : 0.0;
                s32[i] = (v != v) ? 0 : (int32_t)v;                                                    /* [] or NaN = unset */
            }
            check(emgpu_model_set_start(m, s32, ni));
        }
        plhs[0] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
        *(uint64_t *)mxGetData(plhs[0]) = (uint64_t)(uintptr_t)m;
    } else if (!strcmp(cmd, "set_prior")) {
        need(nrhs, 3, "emgpu_mex('set_prior', h, prior)");
        if (mxIsChar(prhs[2])) {
            char s[16];
            get_string(prhs[2], s, sizeof s);
            if (!(s[0] == 'd' || s[0] == 'D') || !(s[1] == 'b' || s[1] == 'B') || !(s[2] == 'e' || s[2] == 'E') || s[3])
                mexErrMsgIdAndTxt("prior:notdbe", "Unknown prior of %s, if char expecting prior = 'dbe'", s);   /* bn_dirichlet_prior.m:28 */
            check(emgpu_model_set_prior(handle_of(prhs[1]), 1, 0.0));
        } else if (mxIsDouble(prhs[2])) {
            check(emgpu_model_set_prior(handle_of(prhs[1]), 0, mxGetScalar(prhs[2])));
        } else {
            mexErrMsgIdAndTxt("prior:unknown", "Second argument must be a char or double");                     /* bn_dirichlet_prior.m:37 */
        }
    } else if (!strcmp(cmd, "set_alpha")) {
        need(nrhs, 3, "emgpu_mex('set_alpha', h, dirichlet_initial, dirichlet_transition)");
        emgpu_model *m = handle_of(prhs[1]);
        for (int pass = 0; pass < 2 && 2 + pass < nrhs; pass++) {
            const mxArray *cell = prhs[2 + pass];
            if (!mxIsCell(cell)) continue;
            const int n = (int)mxGetNumberOfElements(cell);
            for (int i = 0; i < n; i++) {
                const mxArray *c = mxGetCell(cell, i);
                if (!c || mxIsEmpty(c)) continue;
                check(emgpu_model_set_f64(m, pass ? EMGPU_F_ALPHA_TRANSITION : EMGPU_F_ALPHA_INITIAL, i + 1, mxGetPr(c), (int64_t)mxGetNumberOfElements(c)));
            }
        }
    } else if (!strcmp(cmd, "set_start")) {
        need(nrhs, 3, "emgpu_mex('set_start', h, start)");
        emgpu_model *m = handle_of(prhs[1]);
        int32_t st[EMGPU_MEX_MAX_VARS];
        const int n = (int)mxGetNumberOfElements(prhs[2]);
        if (n != n_initial_of(m) || n > EMGPU_MEX_MAX_VARS) mexErrMsgIdAndTxt("emgpu:usage", "start needs n_initial entries");
        for (int i = 0; i < n; i++) { const double v = mxGetPr(prhs[2])[i]; st[i] = (v != v) ? 0 : (int32_t)v; }
        check(emgpu_model_set_start(m, st, n));
    } else if (!strcmp(cmd, "device_count")) {
        int32_t c = 0;
        check(emgpu_device_count(&c));
        plhs[0] = mxCreateDoubleScalar(c);
    } else if (!strcmp(cmd, "use_devices")) {
        need(nrhs, 2, "emgpu_mex('use_devices', [0 1 ...])");
        const int n = (int)mxGetNumberOfElements(prhs[1]);
        if (n < 1 || n > MAX_DEV) mexErrMsgIdAndTxt("emgpu:usage", "1..%d devices", MAX_DEV);
        shutdown_all();
        for (int d = 0; d < n; d++) { check(emgpu_ctx_create((int32_t)mxGetPr(prhs[1])[d], &g_ctx[d])); g_nctx = d + 1; }
        mexAtExit(shutdown_all);
    } else if (!strcmp(cmd, "bn_sample")) {
        need(nrhs, 5, "S = emgpu_mex('bn_sample', h, num_samples, seed, first_index)");
        emgpu_model *m = handle_of(prhs[1]);
        const size_t ni = (size_t)n_initial_of(m);
        emgpu_bn_params p;
        memset(&p, 0, sizeof p);
        p.n = (int64_t)mxGetScalar(prhs[2]); p.seed = (uint64_t)mxGetScalar(prhs[3]); p.first_index = (uint64_t)mxGetScalar(prhs[4]);
        p.flags = EMGPU_FLAG_NO_DEDISC; p.max_attempts = 1;
        const size_t n = (size_t)(p.n > 0 ? p.n : 0);
        uint8_t *ob = (uint8_t *)mxMalloc(ni * n + 1);
        check(emgpu_sample_bn_host(ctx0(), m, &p, ob, NULL, NULL));
        plhs[0] = mxCreateDoubleMatrix((mwSize)n, (mwSize)ni, mxREAL);                               /* S: num_samples x n (bn_sample.m:39) */
        for (size_t v = 0; v < ni; v++) for (size_t i = 0; i < n; i++) mxGetPr(plhs[0])[v * n + i] = ob[v * n + i];
        mxFree(ob);
    } else if (!strcmp(cmd, "sample_uncor")) {
        need(nrhs, 10, "[initial, ev_count, events] = emgpu_mex('sample_uncor', h, n, T, seed, first_index, flags, idxL, idxV, idxDH, layers, event_cap)");
        emgpu_model *m = handle_of(prhs[1]);
        emgpu_sample_params p;
        memset(&p, 0, sizeof p);
        p.n = (int64_t)mxGetScalar(prhs[2]); p.sample_time = (int32_t)mxGetScalar(prhs[3]);
        p.seed = (uint64_t)mxGetScalar(prhs[4]); p.first_index = (uint64_t)mxGetScalar(prhs[5]);
        p.flags = (uint32_t)mxGetScalar(prhs[6]); p.max_attempts = 1000;
        p.idx_L = (int32_t)mxGetScalar(prhs[7]); p.idx_v = (int32_t)mxGetScalar(prhs[8]); p.idx_dh = (int32_t)mxGetScalar(prhs[9]);
        double *layers_rm = NULL;
        if (nrhs > 10 && !mxIsEmpty(prhs[10])) {            /* MATLAB is column-major: transpose to rows [lo hi] */
            const int r = (int)mxGetM(prhs[10]);
            layers_rm = (double *)mxMalloc(sizeof(double) * 2 * (size_t)r);
            for (int i = 0; i < r; i++) { layers_rm[2 * i] = mxGetPr(prhs[10])[i]; layers_rm[2 * i + 1] = mxGetPr(prhs[10])[r + i]; }
            p.layers = layers_rm; p.n_layers = r;
        }
        p.event_cap = nrhs > 11 ? (int32_t)mxGetScalar(prhs[11]) : 512;
        const size_t n = (size_t)(p.n > 0 ? p.n : 0), ni = (size_t)n_initial_of(m), cap = (size_t)p.event_cap;
        float *iv = (float *)mxMalloc(sizeof(float) * (ni * n + 1));
        uint32_t *ec = (uint32_t *)mxMalloc(sizeof(uint32_t) * (n + 1));
        emgpu_event *ev = (emgpu_event *)mxMalloc(sizeof(emgpu_event) * (cap * n + 1));
        emgpu_sample_out o;
        memset(&o, 0, sizeof o);
        o.init_val = iv; o.ev_count = ec; o.events = ev;
        (void)ctx0();
        check(g_nctx > 1 ? emgpu_sample_dbn_multi_host(g_ctx, g_nctx, m, &p, &o) : emgpu_sample_dbn_host(g_ctx[0], m, &p, &o));
        events_out(plhs, nlhs, n, ni, cap, iv, ec, ev);
        mxFree(iv); mxFree(ec); mxFree(ev);
        if (layers_rm) mxFree(layers_rm);
    } else if (!strcmp(cmd, "geom_sample")) {
        need(nrhs, 5, "[outInits, attempts] = emgpu_mex('geom_sample', h, n, seed, first_index, bounds_sample, idxOwn, idxInt, lim1, lim2)");
        emgpu_model *m = handle_of(prhs[1]);
        const size_t ni = (size_t)n_initial_of(m);
        emgpu_bn_params p;
        memset(&p, 0, sizeof p);
        p.n = (int64_t)mxGetScalar(prhs[2]); p.seed = (uint64_t)mxGetScalar(prhs[3]); p.first_index = (uint64_t)mxGetScalar(prhs[4]);
        p.max_attempts = 100000;
        double *bs = NULL;
        if (nrhs > 5 && !mxIsEmpty(prhs[5])) {              /* n_initial x 2 column-major -> row-major (sample.m:45-53) */
            if (mxGetM(prhs[5]) != ni || mxGetN(prhs[5]) != 2) mexErrMsgIdAndTxt("emgpu:usage", "bounds_sample must be n_initial x 2");
            bs = (double *)mxMalloc(sizeof(double) * 2 * ni);
            for (size_t v = 0; v < ni; v++) { bs[2 * v] = mxGetPr(prhs[5])[v]; bs[2 * v + 1] = mxGetPr(prhs[5])[ni + v]; }
            p.bounds_sample = bs;
        }
        p.min_vel1 = p.min_vel2 = 0; p.max_vel1 = p.max_vel2 = INFINITY;
        if (nrhs > 9) {
            p.idx_own_speed = (int32_t)mxGetScalar(prhs[6]); p.idx_int_speed = (int32_t)mxGetScalar(prhs[7]);
            p.min_vel1 = mxGetPr(prhs[8])[0]; p.max_vel1 = mxGetPr(prhs[8])[1]; p.min_vel2 = mxGetPr(prhs[9])[0]; p.max_vel2 = mxGetPr(prhs[9])[1];
        }
        const size_t n = (size_t)(p.n > 0 ? p.n : 0);
        /* a start GRID (InitStartTerminal.m:57-90: one row of presets per sample, n x n_initial, 0 / NaN = unset): one launch for the lot */
        int32_t *grid = NULL;
        double *lw = NULL;
        if (nrhs > 10 && !mxIsEmpty(prhs[10])) {
            if (mxGetM(prhs[10]) != n || mxGetN(prhs[10]) != ni) mexErrMsgIdAndTxt("emgpu:usage", "start_grid must be n x n_initial");
            grid = (int32_t *)mxMalloc(sizeof(int32_t) * (n * ni + 1));
            for (size_t i = 0; i < n; i++) for (size_t v = 0; v < ni; v++) { const double x = mxGetPr(prhs[10])[v * n + i]; grid[i * ni + v] = (x != x) ? 0 : (int32_t)x; }
            p.start = grid;
        }
        if (nlhs > 2) { lw = (double *)mxMalloc(sizeof(double) * (n + 1)); p.log_weight = lw; }
        float *ov = (float *)mxMalloc(sizeof(float) * (ni * n + 1));
        int32_t *att = (int32_t *)mxMalloc(sizeof(int32_t) * (n + 1));
        check(emgpu_sample_bn_host(ctx0(), m, &p, NULL, ov, att));
        plhs[0] = mxCreateDoubleMatrix((mwSize)n, (mwSize)ni, mxREAL);
        for (size_t v = 0; v < ni; v++) for (size_t i = 0; i < n; i++) mxGetPr(plhs[0])[v * n + i] = ov[v * n + i];
        if (nlhs > 1) { plhs[1] = mxCreateDoubleMatrix((mwSize)n, 1, mxREAL); for (size_t i = 0; i < n; i++) mxGetPr(plhs[1])[i] = att[i]; }
        if (nlhs > 2) { plhs[2] = mxCreateDoubleMatrix((mwSize)n, 1, mxREAL); for (size_t i = 0; i < n; i++) mxGetPr(plhs[2])[i] = lw[i]; mxFree(lw); }
        mxFree(ov); mxFree(att);
        if (bs) mxFree(bs);
        if (grid) mxFree(grid);
    } else if (!strcmp(cmd, "propagate_terminal")) {
        need(nrhs, 8, "[out, rows] = emgpu_mex('propagate_terminal', handles, geo, model_of, seed, first_index, tmax_s, dyn_limits)");
        const int nm = (int)mxGetNumberOfElements(prhs[1]);
        if (nm < 1 || nm > 64) mexErrMsgIdAndTxt("emgpu:usage", "1..64 trajectory models");
        const emgpu_model *models[64];
        for (int i = 0; i < nm; i++) models[i] = (const emgpu_model *)(uintptr_t)((uint64_t *)mxGetData(prhs[1]))[i];
        emgpu_term_params p;
        memset(&p, 0, sizeof p);
        p.n = (int64_t)mxGetN(prhs[2]);                      /* geo 12 x n == [n][12] row-major */
        if (mxGetM(prhs[2]) != 12 || mxGetNumberOfElements(prhs[3]) != (size_t)(4 * p.n) || mxGetNumberOfElements(prhs[7]) != 10)
            mexErrMsgIdAndTxt("emgpu:usage", "geo must be 12 x n, model_of 4 x n, dyn_limits 5 x 2");
        p.seed = (uint64_t)mxGetScalar(prhs[4]); p.first_index = (uint64_t)mxGetScalar(prhs[5]); p.tmax_s = mxGetScalar(prhs[6]);
        p.max_resample = 100000; p.cap = (int32_t)p.tmax_s + 3;
        for (int a = 0; a < 2; a++) for (int k = 0; k < 5; k++) p.dyn_limits[a][k] = mxGetPr(prhs[7])[5 * a + k];
        const size_t nl = 4 * (size_t)p.n, cap = (size_t)p.cap;
        int32_t *mo = (int32_t *)mxMalloc(sizeof(int32_t) * (nl + 1)), *rows = (int32_t *)mxMalloc(sizeof(int32_t) * (nl + 1));
        for (size_t i = 0; i < nl; i++) mo[i] = (int32_t)mxGetPr(prhs[3])[i];
        /* the library's layout: the joined track of aircraft 2e + a, [2n][W][5], row C + t = second t (include/emgpu.h) */
        const size_t C0 = (size_t)EMGPU_TERMINAL_T0_ROW(p.cap), W = (size_t)EMGPU_TERMINAL_BLOCK_ROWS(p.cap);
        float *out = (float *)mxCalloc((nl / 2) * W * 5 + 1, sizeof(float));
        check(emgpu_propagate_terminal_host(ctx0(), models, nm, &p, mxGetPr(prhs[2]), mo, out, rows));
        mwSize dims[3];
        dims[0] = (mwSize)nl; dims[1] = (mwSize)cap; dims[2] = 6;                                     /* out(lane, second, field) */
        plhs[0] = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxREAL);
        for (size_t l = 0; l < nl; l++) {                          /* lane 4e + 2a + backward: row r of it is row C +- r of aircraft l / 2 */
            const size_t r_l = (size_t)(rows[l] < 0 ? -rows[l] - 1 : rows[l]);
            const int back = (int)(l & 1);
            for (size_t t = 0; t < cap && t < r_l; t++) {
                const float *q = out + ((l >> 1) * W + (back ? C0 - t : C0 + t)) * 5;
                mxGetPr(plhs[0])[(0 * cap + t) * nl + l] = back ? -(double)t : (double)t;            /* t_s */
                for (size_t f = 1; f < 6; f++) mxGetPr(plhs[0])[(f * cap + t) * nl + l] = q[f - 1];
            }
        }
        if (nlhs > 1) { plhs[1] = mxCreateDoubleMatrix((mwSize)nl, 1, mxREAL); for (size_t l = 0; l < nl; l++) mxGetPr(plhs[1])[l] = rows[l]; }
        mxFree(mo); mxFree(rows); mxFree(out);
    } else if (!strcmp(cmd, "track_uncor")) {
        need(nrhs, 9, "[tracks, limits, attempts] = emgpu_mex('track_uncor', h, n, T, seed, first_index, isQuantize500, isRotorcraft, idx7, stride, max_track_attempts)");
        emgpu_model *m = handle_of(prhs[1]);
        emgpu_utrack_params p;
        memset(&p, 0, sizeof p);
        p.n = (int64_t)mxGetScalar(prhs[2]); p.sample_time = (int32_t)mxGetScalar(prhs[3]);
        p.seed = (uint64_t)mxGetScalar(prhs[4]); p.first_index = (uint64_t)mxGetScalar(prhs[5]);
        p.flags = mxGetScalar(prhs[6]) != 0 ? EMGPU_FLAG_QUANTIZE500 : 0u; p.is_rotorcraft = mxGetScalar(prhs[7]) != 0;
        if (mxGetNumberOfElements(prhs[8]) != 7) mexErrMsgIdAndTxt("emgpu:usage", "idx7 = [idx_G idx_A idx_L idx_V idx_DV idx_DH idx_DPsi], 0 = absent");
        const double *ix = mxGetPr(prhs[8]);
        p.idx_G = (int32_t)ix[0]; p.idx_A = (int32_t)ix[1]; p.idx_L = (int32_t)ix[2]; p.idx_v = (int32_t)ix[3];
        p.idx_dv = (int32_t)ix[4]; p.idx_dh = (int32_t)ix[5]; p.idx_dpsi = (int32_t)ix[6];
        p.record_stride = nrhs > 9 ? (int32_t)mxGetScalar(prhs[9]) : 1;
        p.max_track_attempts = nrhs > 10 ? (int32_t)mxGetScalar(prhs[10]) : 200; p.max_attempts = 1000;
        if (p.record_stride < 1 || p.sample_time < 1) mexErrMsgIdAndTxt("emgpu:usage", "stride and T must be >= 1");
        const size_t n = (size_t)(p.n > 0 ? p.n : 0), S = (size_t)(10 * p.sample_time / p.record_stride + 1);
        mwSize dims[3];
        dims[0] = 8; dims[1] = (mwSize)S; dims[2] = (mwSize)n;                                        /* 8 x S x n == [n][S][8] */
        plhs[0] = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxREAL);
        mxArray *lim = mxCreateDoubleMatrix(3, (mwSize)n, mxREAL);
        int32_t *att = (int32_t *)mxMalloc(sizeof(int32_t) * (n + 1));
        check(emgpu_track_uncor_host(ctx0(), m, &p, mxGetPr(plhs[0]), mxGetPr(lim), att));
        if (nlhs > 1) plhs[1] = lim;
        if (nlhs > 2) { plhs[2] = mxCreateDoubleMatrix((mwSize)n, 1, mxREAL); for (size_t i = 0; i < n; i++) mxGetPr(plhs[2])[i] = att[i]; }
        mxFree(att);
    } else if (!strcmp(cmd, "track_terminal")) {
        need(nrhs, 10, "[sample, traj, len, meta, attempts] = emgpu_mex('track_terminal', hGeom, handles, n, seed, first_index, dyn_limits, cumturn_pitch, thresholds, idx12, bounds_sample, max_track_attempts)");
        emgpu_model *gm = handle_of(prhs[1]);
        if (mxGetNumberOfElements(prhs[2]) != 10) mexErrMsgIdAndTxt("emgpu:usage", "handles: the 10 trajectory models in CorTerminalModel.m:84-100 order");
        const emgpu_model *models[10];
        for (int i = 0; i < 10; i++) models[i] = (const emgpu_model *)(uintptr_t)((uint64_t *)mxGetData(prhs[2]))[i];
        emgpu_ttrack_params p;
        memset(&p, 0, sizeof p);
        p.n = (int64_t)mxGetScalar(prhs[3]); p.seed = (uint64_t)mxGetScalar(prhs[4]); p.first_index = (uint64_t)mxGetScalar(prhs[5]);
        p.tmax_s = 120.0; p.max_resample = 100000; p.max_attempts = 100000;                       /* track.m:33 */
        p.max_track_attempts = nrhs > 11 ? (int32_t)mxGetScalar(prhs[11]) : 2000;
        if (mxGetNumberOfElements(prhs[6]) != 10 || mxGetNumberOfElements(prhs[7]) != 4 || mxGetNumberOfElements(prhs[8]) != 4 || mxGetNumberOfElements(prhs[9]) != 12)
            mexErrMsgIdAndTxt("emgpu:usage", "dyn_limits 5 x 2, cumturn_pitch 1 x 4, thresholds 1 x 4, idx12 1 x 12");
        for (int a = 0; a < 2; a++) for (int k = 0; k < 5; k++) p.dyn_limits[a][k] = mxGetPr(prhs[6])[5 * a + k];
        p.max_cum_turn_deg[0] = mxGetPr(prhs[7])[0]; p.max_cum_turn_deg[1] = mxGetPr(prhs[7])[1]; p.pitch_deg[0] = mxGetPr(prhs[7])[2]; p.pitch_deg[1] = mxGetPr(prhs[7])[3];
        p.min_enc_time_s = mxGetPr(prhs[8])[0]; p.thres_dist_ft = mxGetPr(prhs[8])[1]; p.thres_alt_low_ft = mxGetPr(prhs[8])[2]; p.thres_vertrate_ft_s = mxGetPr(prhs[8])[3];
        for (int k = 0; k < 12; k++) p.idx[k] = (int32_t)mxGetPr(prhs[9])[k];
        const size_t ni = (size_t)n_initial_of(gm), n = (size_t)(p.n > 0 ? p.n : 0);
        double *bs = NULL;
        if (nrhs > 10 && !mxIsEmpty(prhs[10])) {
            if (mxGetM(prhs[10]) != ni || mxGetN(prhs[10]) != 2) mexErrMsgIdAndTxt("emgpu:usage", "bounds_sample must be n_initial x 2");
            bs = (double *)mxMalloc(sizeof(double) * 2 * ni);
            for (size_t v = 0; v < ni; v++) { bs[2 * v] = mxGetPr(prhs[10])[v]; bs[2 * v + 1] = mxGetPr(prhs[10])[ni + v]; }
            p.bounds_sample = bs;
        }
        const int32_t cap2 = 2 * ((int32_t)p.tmax_s + 3);
        mwSize d4[4];
        d4[0] = 6; d4[1] = (mwSize)cap2; d4[2] = 2; d4[3] = (mwSize)n;                           /* 6 x cap2 x 2 x n == [n][2][cap2][6] */
        plhs[0] = mxCreateDoubleMatrix((mwSize)ni, (mwSize)n, mxREAL);
        mxArray *tr = mxCreateNumericArray(4, d4, mxDOUBLE_CLASS, mxREAL), *mt = mxCreateDoubleMatrix(4, (mwSize)n, mxREAL);
        int32_t *ln = (int32_t *)mxMalloc(sizeof(int32_t) * (2 * n + 1)), *att = (int32_t *)mxMalloc(sizeof(int32_t) * (n + 1));
        const int rc = emgpu_track_terminal_host(ctx0(), gm, models, 10, &p, mxGetPr(plhs[0]), mxGetPr(tr), cap2, ln, mxGetPr(mt), att);
        if (bs) mxFree(bs);
        if (rc != EMGPU_ERR_REJECT_CAP) check(rc);           /* at the cap the accepted encounters are still returned: attempts = -1 marks the rest */
        if (nlhs > 1) plhs[1] = tr;
        if (nlhs > 2) { plhs[2] = mxCreateDoubleMatrix(2, (mwSize)n, mxREAL); for (size_t i = 0; i < 2 * n; i++) mxGetPr(plhs[2])[i] = ln[i]; }
        if (nlhs > 3) plhs[3] = mt;
        if (nlhs > 4) { plhs[4] = mxCreateDoubleMatrix((mwSize)n, 1, mxREAL); for (size_t i = 0; i < n; i++) mxGetPr(plhs[4])[i] = att[i]; }
        mxFree(ln); mxFree(att);
    } else if (!strcmp(cmd, "sample2track")) {
        need(nrhs, 7, "sample2track needs alt0, speed0, updates, ur, min_speed, max_speed");
        emgpu_track_params tp;
        memset(&tp, 0, sizeof tp);
        tp.n = (int64_t)mxGetNumberOfElements(prhs[1]);
        tp.T = tp.n ? (int32_t)(mxGetNumberOfElements(prhs[3]) / (3 * (size_t)tp.n)) : 1;
        tp.ur_speed = mxGetPr(prhs[4])[0]; tp.ur_vertrate = mxGetPr(prhs[4])[1]; tp.ur_heading = mxGetPr(prhs[4])[2];
        tp.min_speed = mxGetScalar(prhs[5]); tp.max_speed = mxGetScalar(prhs[6]);
        /* column-major 3 x T x n == row-major [n][T][3], 3 x (T+1) x n == [n][T+1][3], 2 x n == [n][2]: no transposes */
        mwSize dx[3];
        dx[0] = 3; dx[1] = (mwSize)tp.T + 1; dx[2] = (mwSize)tp.n;
        plhs[0] = mxCreateNumericArray(3, dx, mxDOUBLE_CLASS, mxREAL);
        mxArray *fl = mxCreateNumericMatrix((mwSize)tp.n, 1, mxUINT8_CLASS, mxREAL), *vm = mxCreateDoubleMatrix(2, (mwSize)tp.n, mxREAL);
        check(emgpu_sample2track_host(ctx0(), &tp, mxGetPr(prhs[1]), mxGetPr(prhs[2]), mxGetPr(prhs[3]), mxGetPr(plhs[0]),
                                      (uint8_t *)mxGetData(fl), mxGetPr(vm)));
        if (nlhs > 1) plhs[1] = fl;
        if (nlhs > 2) plhs[2] = vm;
    } else if (!strcmp(cmd, "free")) {
        need(nrhs, 2, "emgpu_mex('free', h)");
        emgpu_model_free(handle_of(prhs[1]));
    } else if (!strcmp(cmd, "shutdown")) {
        shutdown_all();
    } else {
        mexErrMsgIdAndTxt("emgpu:usage", "unknown command %s", cmd);
    }
}
