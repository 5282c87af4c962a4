/*
 * emgpu_mex.c -- MATLAB gateway to libemgpu.so (include/emgpu.h).
 *
 * NOT RUN: neither MATLAB nor mex.h exists in the build image (tests only type-check this file against a
 * declaration-only stand-in, tests/stubs/mex.h); this is the binding a maintainer compiles on a machine
 * that has both:
 *     mex -R2018a emgpu_mex.c -I<repo>/include -L<repo>/em_model_manned_bayes_amd -lemgpu
 *
 * One entry point, dispatched on a command string (MATLAB calls are single threaded):
 *   h   = emgpu_mex('load_txt', filename, idxZeroBoundaries, isOverwriteZeroBoundaries)  % em_read.m:1
 *         emgpu_mex('set_prior', h, prior)           % numeric or 'dbe'   (EncounterModel.m:194-203)
 *         emgpu_mex('set_start', h, start)           % double vector, 0/NaN = unset  (bn_sample.m:44-50)
 *   [init_val, ev_count, events] = emgpu_mex('sample_uncor', h, n, T, seed, first_index, flags,
 *                                            idxL, idxV, idxDH, layers, event_cap)
 *         init_val : n x n_initial double        (out_inits, UncorEncounterModel.m:303)
 *         ev_count : n x 1 double
 *         events   : event_cap x 3 x n double    rows [dt var value] (out_events{i}, :304)
 *   [xyz, flags, vminmax] = emgpu_mex('sample2track', alt0, speed0, updates, ur, min_speed, max_speed)
 *         the loop of sample2track.m:182-243 for n trajectories: alt0, speed0 n x 1; updates 3 x T x n
 *         (vertical rate, acceleration, turn rate per second); ur = [ur_speed ur_vertrate ur_heading];
 *         xyz 3 x (T+1) x n feet; flags n x 1 (bit 0 CFIT, bit 1 speed); vminmax 2 x n
 *         emgpu_mex('free', h)
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "mex.h"
#include "emgpu.h"

static emgpu_ctx *g_ctx = NULL;

static void check(int rc) {
    if (rc < 0) {
        /* same identifiers the reference raises where it has them */
        const char *id = rc == EMGPU_ERR_PRIOR ? "prior:notdbe" : (rc == EMGPU_ERR_PRESET ? "emgpu:preset" : "emgpu:error");
        mexErrMsgIdAndTxt(id, "%s", emgpu_last_error());
    }
}

static emgpu_model *handle_of(const mxArray *a) { return (emgpu_model *)(uintptr_t)(*(uint64_t *)mxGetData(a)); }

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    char cmd[64];
    if (nrhs < 1 || mxGetString(prhs[0], cmd, sizeof cmd)) mexErrMsgIdAndTxt("emgpu:usage", "first argument must be a command string");
    if (!g_ctx) check(emgpu_ctx_create(0, &g_ctx));

    if (!strcmp(cmd, "load_txt")) {
        char path[4096];
        mxGetString(prhs[1], path, sizeof path);
        int32_t idx[64], n_idx = 0;
        if (nrhs > 2) { n_idx = (int32_t)mxGetNumberOfElements(prhs[2]); for (int i = 0; i < n_idx; i++) idx[i] = (int32_t)mxGetPr(prhs[2])[i]; }
        emgpu_model *m = NULL;
        check(emgpu_model_load_txt(path, n_idx ? idx : NULL, n_idx, nrhs > 3 && mxIsLogicalScalarTrue(prhs[3]), &m));
        plhs[0] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
        *(uint64_t *)mxGetData(plhs[0]) = (uint64_t)(uintptr_t)m;
    } else if (!strcmp(cmd, "set_prior")) {
        if (mxIsChar(prhs[2])) {
            char s[16]; mxGetString(prhs[2], s, sizeof s);
            check(emgpu_model_set_prior(handle_of(prhs[1]), (s[0] == 'd' || s[0] == 'D') ? 1 : 2, 0.0));
        } else check(emgpu_model_set_prior(handle_of(prhs[1]), 0, mxGetScalar(prhs[2])));
    } else if (!strcmp(cmd, "set_start")) {
        int32_t st[64]; int n = (int)mxGetNumberOfElements(prhs[2]);
        for (int i = 0; i < n; i++) { double v = mxGetPr(prhs[2])[i]; st[i] = (v != v) ? 0 : (int32_t)v; }
        check(emgpu_model_set_start(handle_of(prhs[1]), st, n));
    } else if (!strcmp(cmd, "sample_uncor")) {
        emgpu_model *m = handle_of(prhs[1]);
        emgpu_model_info_t info; check(emgpu_model_info(m, &info));
        emgpu_sample_params p; memset(&p, 0, sizeof p);
        p.n = (int64_t)mxGetScalar(prhs[2]); p.sample_time = (int32_t)mxGetScalar(prhs[3]);
        p.seed = (uint64_t)mxGetScalar(prhs[4]); p.first_index = (uint64_t)mxGetScalar(prhs[5]);
        p.flags = (uint32_t)mxGetScalar(prhs[6]); p.max_attempts = 1000;
        p.idx_L = (int32_t)mxGetScalar(prhs[7]); p.idx_v = (int32_t)mxGetScalar(prhs[8]); p.idx_dh = (int32_t)mxGetScalar(prhs[9]);
        double *layers_rm = NULL;
        if (nrhs > 10 && !mxIsEmpty(prhs[10])) {            /* MATLAB is column-major: transpose to rows [lo hi] */
            int r = (int)mxGetM(prhs[10]); layers_rm = (double *)mxMalloc(sizeof(double) * 2 * r);
            for (int i = 0; i < r; i++) { layers_rm[2 * i] = mxGetPr(prhs[10])[i]; layers_rm[2 * i + 1] = mxGetPr(prhs[10])[r + i]; }
            p.layers = layers_rm; p.n_layers = r;
        }
        p.event_cap = nrhs > 11 ? (int32_t)mxGetScalar(prhs[11]) : 512;
        const size_t n = (size_t)p.n, ni = info.n_initial, cap = p.event_cap;
        float *iv = (float *)mxMalloc(sizeof(float) * ni * n);
        uint32_t *ec = (uint32_t *)mxMalloc(sizeof(uint32_t) * n);
        emgpu_event *ev = (emgpu_event *)mxMalloc(sizeof(emgpu_event) * cap * n);
        emgpu_sample_out o; memset(&o, 0, sizeof o);
        o.init_val = iv; o.ev_count = ec; o.events = ev;
        check(emgpu_sample_dbn_host(g_ctx, m, &p, &o));
        plhs[0] = mxCreateDoubleMatrix(n, ni, mxREAL);
        for (size_t v = 0; v < ni; v++) for (size_t i = 0; i < n; i++) mxGetPr(plhs[0])[v * n + i] = iv[v * n + i];
        plhs[1] = mxCreateDoubleMatrix(n, 1, mxREAL);
        mwSize dims[3] = {cap, 3, n};
        plhs[2] = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxREAL);
        double *E = mxGetPr(plhs[2]);
        for (size_t i = 0; i < n; i++) {
            mxGetPr(plhs[1])[i] = ec[i];
            for (size_t e = 0; e < ec[i] && e < cap; e++) {
                const emgpu_event *r = &ev[i * cap + e];
                E[i * cap * 3 + e] = r->dt; E[i * cap * 3 + cap + e] = r->var; E[i * cap * 3 + 2 * cap + e] = r->value;
            }
        }
        mxFree(iv); mxFree(ec); mxFree(ev); if (layers_rm) mxFree(layers_rm);
    } else if (!strcmp(cmd, "sample2track")) {
        if (nrhs < 7) mexErrMsgIdAndTxt("emgpu:usage", "sample2track needs alt0, speed0, updates, ur, min_speed, max_speed");
        emgpu_track_params tp; memset(&tp, 0, sizeof tp);
        tp.n = (int64_t)mxGetNumberOfElements(prhs[1]);
        tp.T = tp.n ? (int32_t)(mxGetNumberOfElements(prhs[3]) / (3 * (size_t)tp.n)) : 1;
        tp.ur_speed = mxGetPr(prhs[4])[0]; tp.ur_vertrate = mxGetPr(prhs[4])[1]; tp.ur_heading = mxGetPr(prhs[4])[2];
        tp.min_speed = mxGetScalar(prhs[5]); tp.max_speed = mxGetScalar(prhs[6]);
        /* column-major 3 x T x n == row-major [n][T][3], 3 x (T+1) x n == [n][T+1][3], 2 x n == [n][2]: no transposes */
        mwSize dx[3] = {3, (mwSize)tp.T + 1, (mwSize)tp.n};
        plhs[0] = mxCreateNumericArray(3, dx, mxDOUBLE_CLASS, mxREAL);
        plhs[1] = mxCreateNumericMatrix((mwSize)tp.n, 1, mxUINT8_CLASS, mxREAL);
        plhs[2] = mxCreateDoubleMatrix(2, (mwSize)tp.n, mxREAL);
        check(emgpu_sample2track_host(g_ctx, &tp, mxGetPr(prhs[1]), mxGetPr(prhs[2]), mxGetPr(prhs[3]), mxGetPr(plhs[0]),
                                      (uint8_t *)mxGetData(plhs[1]), mxGetPr(plhs[2])));
    } else if (!strcmp(cmd, "free")) {
        emgpu_model_free(handle_of(prhs[1]));
    } else {
        mexErrMsgIdAndTxt("emgpu:usage", "unknown command %s", cmd);
    }
}
