/* examples/c_api_demo.c -- the C ABI used from plain C, the way a mex gateway or any other host binds it:
 *   gcc -O2 -Iinclude examples/c_api_demo.c -Lem_model_manned_bayes_amd -lemgpu -Wl,-rpath,$PWD/em_model_manned_bayes_amd -o c_api_demo
 *   ./c_api_demo model.txt [n] [T] [seed]
 * Loads a model file (em_read.m), draws n trajectories of T seconds like
 * UncorEncounterModel.sample (UncorEncounterModel.m:192-313) into host buffers, runs the
 * sample2track dead reckoning (sample2track.m:183-243) on the same values and prints a summary. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "emgpu.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != EMGPU_OK) {                                                       \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, emgpu_last_error());       \
            return 1;                                                                \
        }                                                                            \
    } while (0)

static int label_index(const char *labels, const char *want) { /* 1-based position of "want" in a '\n'-separated list, 0 = absent */
    int idx = 1;
    const char *p = labels;
    size_t n = strlen(want);
    while (*p) {
        const char *e = strchr(p, '\n');
        size_t len = e ? (size_t)(e - p) : strlen(p);
        if (len == n && strncmp(p, want, n) == 0) return idx;
        if (!e) break;
        p = e + 1;
        idx++;
    }
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s model.txt [n] [T] [seed]\n", argv[0]);
        return 2;
    }
    const int64_t n = argc > 2 ? atoll(argv[2]) : 1000;
    const int T = argc > 3 ? atoi(argv[3]) : 120;
    const uint64_t seed = argc > 4 ? strtoull(argv[4], NULL, 10) : 1;

    emgpu_model *m = NULL;
    emgpu_ctx *ctx = NULL;
    CHECK(emgpu_model_load_txt(argv[1], NULL, 0, 0, &m));
    emgpu_model_info_t info;
    CHECK(emgpu_model_info(m, &info));
    char labels[4096];
    if (emgpu_model_get_text(m, EMGPU_F_LABELS_INITIAL, labels, sizeof labels) < 0) return 1;
    CHECK(emgpu_ctx_create(0, &ctx));

    const int ni = info.n_initial, nd = info.n_dyn, G4 = (T + 3) / 4;
    emgpu_sample_params p;
    memset(&p, 0, sizeof p);
    p.seed = seed; p.n = n; p.sample_time = T; p.max_attempts = 1000;
    p.idx_L = label_index(labels, "\"L\"");
    p.idx_v = label_index(labels, "\"v\"");
    p.idx_dh = label_index(labels, "\"\\dot h\"");
    emgpu_sample_out o;
    memset(&o, 0, sizeof o);
    o.init_val = (float *)calloc((size_t)ni * n, sizeof(float));
    o.dyn_val = (float *)calloc((size_t)G4 * nd * n * 4, sizeof(float));
    o.dyn_bin = (uint32_t *)calloc((size_t)G4 * nd * n, sizeof(uint32_t));
    o.attempts = (int32_t *)calloc((size_t)n, sizeof(int32_t));
    CHECK(emgpu_sample_dbn_host(ctx, m, &p, &o));

    /* dense trace -> the three update columns of sample2track (rows of the temporal map, ascending id:
     * \dot v, \dot h, \dot\psi for the uncorrelated models) */
    int changes = 0;
    double *upd = (double *)calloc((size_t)n * T * 3, sizeof(double));
    double *alt0 = (double *)calloc((size_t)n, sizeof(double)), *v0 = (double *)calloc((size_t)n, sizeof(double));
    for (int64_t i = 0; i < n; i++) {
        alt0[i] = p.idx_L ? o.init_val[(size_t)(p.idx_L - 1) * n + i] : 0.0;
        v0[i] = p.idx_v ? o.init_val[(size_t)(p.idx_v - 1) * n + i] : 0.0;
        for (int c = 0; c < T; c++) {
            for (int k = 0; k < nd && k < 3; k++) {
                const size_t w = ((size_t)(c / 4) * nd + k) * n + i;
                const float v = o.dyn_val[w * 4 + c % 4];
                const int col = k == 0 ? 1 : (k == 1 ? 0 : 2); /* temporal map row -> (vertrate, acc, turnrate) */
                upd[((size_t)i * T + c) * 3 + col] = v;
                if (c > 0 && ((o.dyn_bin[w] >> (8 * (c % 4))) & 0xFF) != ((o.dyn_bin[((size_t)((c - 1) / 4) * nd + k) * n + i] >> (8 * ((c - 1) % 4))) & 0xFF))
                    changes++;
            }
        }
    }
    double bnd[64];
    const int64_t nb = p.idx_v ? emgpu_model_get_f64(m, EMGPU_F_BOUNDARIES, p.idx_v, bnd, 64) : 0;
    emgpu_track_params tp;
    memset(&tp, 0, sizeof tp);
    tp.n = n; tp.T = T;
    tp.ur_speed = (1852.0 / 0.3048) / 3600.0; tp.ur_vertrate = 1.0 / 60.0; tp.ur_heading = 1.0;
    tp.min_speed = nb > 0 ? bnd[0] : 0.0;
    tp.max_speed = nb > 0 ? bnd[nb - 1] : 1e9;
    uint8_t *flags = (uint8_t *)calloc((size_t)n, 1);
    double *xyz = (double *)calloc((size_t)n * (T + 1) * 3, sizeof(double));
    CHECK(emgpu_sample2track_host(ctx, &tp, alt0, v0, upd, xyz, flags, NULL));
    int64_t good = 0, retried = 0;
    for (int64_t i = 0; i < n; i++) { good += flags[i] == 0; retried += o.attempts[i] > 1; }
    printf("model: %d initial / %d dynamic variables; %lld trajectories x %d s (kernel %s)\n", ni, nd, (long long)n, T, emgpu_last_kernel_name(ctx));
    printf("bin changes per trajectory: %.3f; rejection retries: %lld; tracks accepted by sample2track: %lld\n",
           (double)changes / (double)n, (long long)retried, (long long)good);
    printf("trajectory 0: altitude %.1f ft, speed %.1f kt; after %d s: x %.0f ft, y %.0f ft, z %.0f ft\n", alt0[0], v0[0], T,
           xyz[(size_t)T * 3], xyz[(size_t)T * 3 + 1], xyz[(size_t)T * 3 + 2]);
    emgpu_ctx_free(ctx);
    emgpu_model_free(m);
    return 0;
}
