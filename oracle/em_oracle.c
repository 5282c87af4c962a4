/*
 * oracle/em_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * Scalar, f64, single-threaded restatement of the sampling hot path of
 * Airspace-Encounter-Models/em-model-manned-bayes (MATLAB).  It is the CHECKER
 * for the HIP kernels in em_model_manned_bayes_amd/csrc and is never linked,
 * imported or executed by the product path: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may use it.
 *
 * PARITY PINNING: the reference is MATLAB-only, ships no tests / golden vectors
 * and cannot be executed in the build container (no MATLAB, no Octave).
 * ==> "parity unpinned" against real MATLAB output.  What pins this file:
 *   (1) the RNG-free known answers derived from the cited lines (SURVEY.md
 *       Appendix D), (2) an independently written numpy restatement
 *       (oracle/pyref.py) that uses numpy's MT19937 (== MATLAB rng(s,'twister'))
 *       and must agree with this file draw-for-draw, (3) the Random123
 *       Philox4x32-10 known-answer vectors, (4) chi-square tests against the CPTs.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * code/matlab/ of the reference).  All indices are 1-based like the reference
 * unless a comment says otherwise.
 *
 * Two uniform sources:
 *   EM_RNG_MT19937 : one global MT19937 stream, genrand_res53 doubles, consumed in
 *                    exactly the order the reference calls rand.
 *   EM_RNG_PHILOX  : counter-based Philox4x32-R (R = EM_PHILOX_ROUNDS); every reference rand call site
 *                    has a fixed SLOT (section, a, idx) so that the result does not
 *                    depend on draw order, GPU count or launch shape.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#if defined(__GNUC__)
#pragma STDC FP_CONTRACT OFF
#endif

/* ------------------------------------------------------------------------- */
/* MT19937 (Matsumoto & Nishimura 1998/2002): init_genrand + genrand_res53.   */
/* MATLAB rng(seed,'twister'); rand  ==  this stream.         */
/* ------------------------------------------------------------------------- */
#define MT_N 624
#define MT_M 397
typedef struct { uint32_t mt[MT_N]; int mti; } em_mt_t;

static void mt_seed(em_mt_t *s, uint32_t seed) {
    s->mt[0] = seed;
    for (int i = 1; i < MT_N; i++)
        s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->mti = MT_N;
}
static uint32_t mt_u32(em_mt_t *s) {
    static const uint32_t mag01[2] = {0u, 0x9908b0dfu};
    uint32_t y;
    if (s->mti >= MT_N) {
        int kk;
        for (kk = 0; kk < MT_N - MT_M; kk++) {
            y = (s->mt[kk] & 0x80000000u) | (s->mt[kk + 1] & 0x7fffffffu);
            s->mt[kk] = s->mt[kk + MT_M] ^ (y >> 1) ^ mag01[y & 1u];
        }
        for (; kk < MT_N - 1; kk++) {
            y = (s->mt[kk] & 0x80000000u) | (s->mt[kk + 1] & 0x7fffffffu);
            s->mt[kk] = s->mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ mag01[y & 1u];
        }
        y = (s->mt[MT_N - 1] & 0x80000000u) | (s->mt[0] & 0x7fffffffu);
        s->mt[MT_N - 1] = s->mt[MT_M - 1] ^ (y >> 1) ^ mag01[y & 1u];
        s->mti = 0;
    }
    y = s->mt[s->mti++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
static double mt_res53(em_mt_t *s) {
    uint32_t a = mt_u32(s) >> 5, b = mt_u32(s) >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}

/* ------------------------------------------------------------------------- */
/* Philox4x32-R (Salmon et al., SC'11; Random123 v1.14 philox.h constants).    */
/* The generator runs EM_PHILOX_ROUNDS rounds -- the same number as the kernels' */
/* EMGPU_PHILOX_ROUNDS (csrc/emgpu_plan.h); em_philox4x32_10 is kept for the     */
/* published Random123 known-answer vectors, which pin the round function.       */
#ifndef EM_PHILOX_ROUNDS
#define EM_PHILOX_ROUNDS 7
#endif
void em_philox4x32_r(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < rounds; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
void em_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { em_philox4x32_r(ctr, key, 10, out); }
int em_philox_rounds(void) { return EM_PHILOX_ROUNDS; }
static void em_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { em_philox4x32_r(ctr, key, EM_PHILOX_ROUNDS, out); }

/* Slot sections.  ctr = {gidx_lo, gidx_hi, attempt,
 * section<<28 | a<<20 | (idx>>2)}, word = idx & 3, key = {seed_lo, seed_hi}. */
enum {
    EM_SEC_INIT = 1,         /* bn_sample draw of initial node: a=0, idx = var-1            */
    EM_SEC_DEDISC_INIT = 2,  /* dediscretize of initial var:    a=0, idx = var-1            */
    /* per-second slots are keyed by the ABSOLUTE EVENT TIME at = 1..T: the time (seconds since
     * the start) at which the row the draw produces appears in the event list; a transition
     * at `at` produces column c = at of events2samples (0-based), a resample hit of the at-th
     * second takes effect from column at. */
    EM_SEC_TRANS = 3,        /* transition draw:                    a = tvar-1, idx = at    */
    EM_SEC_RES = 4,          /* resample Bernoulli:                 a = var-1,  idx = at    */
    EM_SEC_DEDISC_RES = 5,   /* dediscretize of a resample event:   a = var-1,  idx = at    */
    EM_SEC_DEDISC_TRANS = 6, /* dediscretize of a transition event: a = var-1,  idx = at    */
    EM_SEC_LAYER = 7,        /* UncorEncounterModel.sample 'layers' draw: a=0, idx=0        */
    EM_SEC_GEOM_DEDISC = 8,  /* CorTerminalModel.sample dediscretize: a=0, idx = var-1      */
    /* TRANS and RES are SPLIT slots: the 32-bit draw is x = (H << 16) | L where H is halfword
     * (idx & 7) of block (section, a, idx >> 3) and L is halfword (idx & 7) of block
     * (section + 6, a, idx >> 3).  Halfword h of a block: word h >> 1, upper 16 bits when h is
     * odd, lower 16 bits when h is even.  (A kernel can decide almost every compare from H alone
     * and fetch L lazily; the value drawn is the same full 32-bit uniform either way.) */
    EM_SEC_TRANS_LO = 9,
    EM_SEC_RES_LO = 10,
    /* terminal trajectory propagation (createEncounter.m:93-265): counter word 2 ("attempt") carries
     * role + 4*resample_attempt with role = 2*(aircraft-1) + (direction == backward); idx = step ii */
    EM_SEC_TERM_TRANS = 11,  /* dbn_sample(...,2,start) transition draw: a = 0, idx = 4 ii + row: block ii holds the step's draws of all  */
                             /* three dynamic variables (word = the variable's row of the temporal map); round 5: word 3 of the block  */
                             /* is the FIRST dediscretize draw the attempt makes (one Philox call per attempt for almost every step)   */
    EM_SEC_TERM_DEDISC = 12  /* an attempt's SECOND and THIRD dediscretize draw (two events in one step): idx = 4 ii + row as above    */
};

enum { EM_RNG_MT19937 = 0, EM_RNG_PHILOX = 1 };

typedef struct {
    int32_t mode;
    uint32_t key[2];
    uint64_t gidx;
    uint32_t attempt;
    em_mt_t mt;
    uint64_t n_draws; /* statistics: number of uniforms consumed */
    uint32_t cache_ctr[4], cache_out[4]; /* last Philox block (one block = 4 consecutive idx) */
    int32_t cache_valid;
} em_rng_t;

void em_rng_init(em_rng_t *g, int mode, uint64_t seed) {
    memset(g, 0, sizeof *g);
    g->mode = mode;
    g->key[0] = (uint32_t)seed;
    g->key[1] = (uint32_t)(seed >> 32);
    if (mode == EM_RNG_MT19937) mt_seed(&g->mt, (uint32_t)seed);
}

/* uniform32: x' = min(x, 2^32-2); u = (x' + 0.5) * 2^-32  in (0,1)   */
double em_uniform32(uint32_t x) {
    if (x > 0xFFFFFFFEu) x = 0xFFFFFFFEu;
    return ((double)x + 0.5) * (1.0 / 4294967296.0);
}

uint32_t em_philox_word(em_rng_t *g, uint32_t section, uint32_t a, uint32_t idx) {
    uint32_t ctr[4];
    ctr[0] = (uint32_t)g->gidx;
    ctr[1] = (uint32_t)(g->gidx >> 32);
    ctr[2] = g->attempt;
    ctr[3] = (section << 28) | (a << 20) | (idx >> 2);
    if (!g->cache_valid || memcmp(ctr, g->cache_ctr, sizeof ctr) != 0) {
        em_philox4x32(ctr, g->key, g->cache_out);
        memcpy(g->cache_ctr, ctr, sizeof ctr);
        g->cache_valid = 1;
    }
    return g->cache_out[idx & 3u];
}

static uint32_t em_philox_half(em_rng_t *g, uint32_t section, uint32_t a, uint32_t idx) {
    uint32_t ctr[4], out[4];
    const uint32_t h = idx & 7u;
    ctr[0] = (uint32_t)g->gidx;
    ctr[1] = (uint32_t)(g->gidx >> 32);
    ctr[2] = g->attempt;
    ctr[3] = (section << 28) | (a << 20) | (idx >> 3);
    em_philox4x32(ctr, g->key, out);
    return (h & 1u) ? (out[h >> 1] >> 16) : (out[h >> 1] & 0xFFFFu);
}

/* One MATLAB `rand` call.  In MT mode the slot is ignored. */
static double em_rand(em_rng_t *g, uint32_t section, uint32_t a, uint32_t idx) {
    g->n_draws++;
    if (g->mode == EM_RNG_MT19937) return mt_res53(&g->mt);
    if (section == EM_SEC_TRANS || section == EM_SEC_RES) { /* split slot */
        const uint32_t hi = em_philox_half(g, section, a, idx);
        const uint32_t lo = em_philox_half(g, section + 6u, a, idx);
        return em_uniform32((hi << 16) | lo);
    }
    return em_uniform32(em_philox_word(g, section, a, idx));
}

/* ------------------------------------------------------------------------- */
/* Model container (flat arrays; filled by oracle/oracle.py)                  */
/* ------------------------------------------------------------------------- */
typedef struct {
    int32_t n_initial, n_transition, n_dyn, _pad;
    const uint8_t *G_initial;        /* n_i x n_i row-major, [parent][child] (em_read.m:204) */
    const uint8_t *G_transition;     /* n_t x n_t row-major                                  */
    const int32_t *r_initial;        /* n_i                                                  */
    const int32_t *r_transition;     /* n_t                                                  */
    const int32_t *order_initial;    /* n_i, 1-based variable ids (bn_sort)                  */
    const int32_t *order_transition; /* n_t                                                  */
    const int32_t *temporal_map;     /* n_dyn x 2 row-major, 1-based (var@t, var@t+1)        */
    const double *N_initial;         /* concatenated r_i x q_i column-major (em_read.m:191)  */
    const double *A_initial;         /* alpha, same layout (bn_dirichlet_prior.m)            */
    const int64_t *off_initial;      /* n_i offsets into N_initial / A_initial               */
    const double *N_transition;      /* nodes n_i+1..n_t (em_read.m:92)                      */
    const double *A_transition;
    const int64_t *off_transition;   /* n_t offsets; -1 for nodes without a table            */
    const double *boundaries;        /* concatenated per initial var                         */
    const int32_t *bnd_off;          /* n_i                                                  */
    const int32_t *bnd_len;          /* n_i ; 0 => empty ('*' in the file, em_read.m:97-99)  */
    const int32_t *zero_bins;        /* n_i ; 0 => none                                      */
    const double *resample_rates;    /* n_i                                                  */
    const int32_t *start;            /* n_i ; 0 => unset                                     */
} em_model_t;

typedef struct {
    double dt;      /* seconds since the previous row                                        */
    int32_t var;    /* 1-based initial-network variable id; 0 = terminator                   */
    int32_t bin;    /* discrete value                                                        */
    double val;     /* dediscretised value (== bin until dediscretize ran)                   */
    int32_t kind;   /* 0 = transition event, 1 = resample event, 2 = terminator              */
    int32_t atime;  /* absolute time of the row in seconds since the start (1..T)            */
} em_event_t;

/* ------------------------------------------------------------------------- */
/* a1  asub2ind.m:13-14                                                       */
/* ------------------------------------------------------------------------- */
int64_t em_asub2ind(const int32_t *siz, const int32_t *x, int n) {
    /* k = [1 cumprod(siz(1:end-1))]; ndx = k*(x-1) + 1 */
    int64_t k = 1, ndx = 1;
    for (int i = 0; i < n; i++) {
        ndx += k * (int64_t)(x[i] - 1);
        k *= siz[i];
    }
    return ndx;
}

/* ------------------------------------------------------------------------- */
/* a2  select_random.m:14-20 with the uniform injected                        */
/* ------------------------------------------------------------------------- */
int em_select_random_r(const double *weights, int n, double r) {
    /* s = cumsum(weights); sthres = s(end)*r; index = find(s >= sthres, 1, 'first') */
    double s_end = 0.0;
    for (int i = 0; i < n; i++) s_end += weights[i];   /* sequential sum == cumsum(end) */
    double sthres = s_end * r;
    double s = 0.0;
    for (int i = 0; i < n; i++) {
        s += weights[i];
        if (s >= sthres) return i + 1;
    }
    return n; /* unreachable for r < 1 */
}

/* weights = N{i}(:,j) + alpha{i}(:,j)  (bn_sample.m:55, dbn_sample.m:77,123-127) */
static void column_weights(const double *N, const double *A, int64_t off, int r, int64_t j, double *w) {
    const double *n = N + off + (j - 1) * r;
    const double *a = A + off + (j - 1) * r;
    for (int k = 0; k < r; k++) w[k] = n[k] + a[k];
}

/* j = asub2ind(r(parents), x(parents)) with parents = logical column G(:,i)
 * (bn_sample.m:42,53; dbn_sample.m:72-75): parents ascend in variable index. */
static int64_t parent_config(const uint8_t *G, int n, const int32_t *r, const int32_t *x, int child) {
    int32_t siz[64], sub[64];
    int np = 0;
    for (int p = 0; p < n; p++)
        if (G[p * n + (child - 1)]) { siz[np] = r[p]; sub[np] = x[p]; np++; }
    if (np == 0) return 1;
    return em_asub2ind(siz, sub, np);
}

#define EM_MAX_R 64

/* ------------------------------------------------------------------------- */
/* a4  bn_sample.m:39-57 (one sample).  Returns 0, or -1 for                  */
/* 'Attempt to preset a dependent variable' (bn_sample.m:45-47).              */
/* r must be indexable by every node of G (the reference passes r_transition  */
/* from dbn_sample.m:36, r_initial from @CorTerminalModel/sample.m:34).       */
/* ------------------------------------------------------------------------- */
int em_bn_sample(const em_model_t *m, em_rng_t *g, const int32_t *r, int32_t *S /* n_initial */) {
    int n = m->n_initial;
    double w[EM_MAX_R];
    for (int i = 0; i < n; i++) S[i] = 0;
    for (int oi = 0; oi < n; oi++) {
        int i = m->order_initial[oi]; /* 1-based */
        int has_par = 0, n_par = 0, n_par_set = 0;
        for (int p = 0; p < n; p++)
            if (m->G_initial[p * n + (i - 1)]) { has_par = 1; n_par++; if (m->start[p] != 0) n_par_set++; }
        if (m->start[i - 1] != 0) {
            if (has_par && n_par_set < n_par) return -1;
            S[i - 1] = m->start[i - 1];
        } else {
            int64_t j = 1;
            if (has_par) j = parent_config(m->G_initial, n, r, S, i);
            column_weights(m->N_initial, m->A_initial, m->off_initial[i - 1], r[i - 1], j, w);
            double u = em_rand(g, EM_SEC_INIT, 0, (uint32_t)(i - 1));
            S[i - 1] = em_select_random_r(w, r[i - 1], u);
        }
    }
    return 0;
}

static int is_dynamic(const em_model_t *m, int var) {
    for (int k = 0; k < m->n_dyn; k++) if (m->temporal_map[2 * k + 1] == var) return 1;
    return 0;
}

/* dbn_sample.m:55  is_dynvar_depend = any(G_transition(dyn,dyn),'all') */
int em_is_dynvar_depend(const em_model_t *m) {
    int nt = m->n_transition;
    for (int a = 0; a < m->n_dyn; a++)
        for (int b = 0; b < m->n_dyn; b++)
            if (m->G_transition[(m->temporal_map[2 * a + 1] - 1) * nt + (m->temporal_map[2 * b + 1] - 1)]) return 1;
    return 0;
}

/* ------------------------------------------------------------------------- */
/* a5/a6  dbn_sample.m:36-166.  per_step: 0 = REFERENCE_AUTO (branch chosen   */
/* by is_dynvar_depend, dbn_sample.m:55), 1 = force the dependent branch      */
/* (true per-timestep DBN; not reference behaviour for "fast" models).        */
/* events: rows (dt, var, bin); returns row count, or <0 on error.            */
/* ------------------------------------------------------------------------- */
int em_dbn_sample(const em_model_t *m, em_rng_t *g, int t_max, int per_step,
                  int32_t *initial, em_event_t *events, int cap) {
    int ni = m->n_initial, nt = m->n_transition, nd = m->n_dyn;
    int32_t x[128], x_old[128];
    double w[EM_MAX_R];
    if (em_bn_sample(m, g, m->r_transition ? m->r_transition : m->r_initial, initial) != 0) return -1;
    if (nd == 0 || t_max < 2) return 0;
    for (int i = 0; i < ni; i++) x[i] = initial[i];
    for (int i = ni; i < nt; i++) x[i] = 0;      /* x = [initial zeros(...)]  dbn_sample.m:40 */
    double delta_t = 0;
    int counter = 0;
    int depend = per_step ? 1 : em_is_dynvar_depend(m);

    /* thresholds of the fast branch: s{ii}, sthres(:,ii)  dbn_sample.m:104-135 */
    double *scum = NULL, *sthres = NULL;
    if (!depend) {
        scum = (double *)calloc((size_t)nt * EM_MAX_R, sizeof(double));
        sthres = (double *)calloc((size_t)nt * (size_t)t_max, sizeof(double));
        for (int oi = 0; oi < nt; oi++) {
            int ii = m->order_transition[oi];
            if (!is_dynamic(m, ii)) continue;
            int64_t j = parent_config(m->G_transition, nt, m->r_transition, x, ii); /* :111-120 */
            int r = m->r_transition[ii - 1];
            column_weights(m->N_transition, m->A_transition, m->off_transition[ii - 1], r, j, w);
            double s = 0;
            for (int k = 0; k < r; k++) { s += w[k]; scum[(ii - 1) * EM_MAX_R + k] = s; }  /* :130 */
            /* sthres(:,ii) = s{ii}(end) * rand(t_max,1)  :133.  Row 1 is drawn but unused. */
            for (int t = 1; t <= t_max; t++) {
                double u;
                if (g->mode == EM_RNG_MT19937 || t >= 2) u = em_rand(g, EM_SEC_TRANS, (uint32_t)(ii - 1), (uint32_t)(t - 1));
                else u = 0.0; /* Philox mode: slot (ii, c=0) is never used, skip the draw */
                sthres[(size_t)(ii - 1) * t_max + (t - 1)] = s * u;
            }
        }
    }

    for (int t = 2; t <= t_max; t++) {                                   /* :66 / :138 */
        delta_t += 1;
        memcpy(x_old, x, sizeof(int32_t) * nt);
        for (int oi = 0; oi < nt; oi++) {
            int i = m->order_transition[oi];
            if (!is_dynamic(m, i)) continue;
            int r = m->r_transition[i - 1];
            if (depend) {
                int64_t j = parent_config(m->G_transition, nt, m->r_transition, x, i);   /* :72-75 */
                column_weights(m->N_transition, m->A_transition, m->off_transition[i - 1], r, j, w);
                double u = em_rand(g, EM_SEC_TRANS, (uint32_t)(i - 1), (uint32_t)(t - 1));
                x[i - 1] = em_select_random_r(w, r, u);                                   /* :77 */
            } else {
                /* x(ii) = find(s{ii} >= sthres(t,ii), 1, 'first')  :144 */
                double th = sthres[(size_t)(i - 1) * t_max + (t - 1)];
                int k = 0;
                while (k < r - 1 && !(scum[(i - 1) * EM_MAX_R + k] >= th)) k++;
                x[i - 1] = k + 1;
            }
        }
        for (int k = 0; k < nd; k++) x[m->temporal_map[2 * k] - 1] = x[m->temporal_map[2 * k + 1] - 1]; /* :82/:149 */
        for (int i = 0; i < ni; i++) {                                                    /* :84-91 / :151-161 */
            if (x[i] != x_old[i]) {
                if (counter >= cap) { free(scum); free(sthres); return -2; }
                events[counter].dt = delta_t;
                events[counter].var = i + 1;
                events[counter].bin = x[i];
                events[counter].val = x[i];
                events[counter].kind = 0;
                events[counter].atime = t - 1;
                counter++;
                delta_t = 0;
            }
        }
    }
    free(scum); free(sthres);
    return counter;
}

/* ------------------------------------------------------------------------- */
/* a8  resample_events.m:16-37                                                */
/* ------------------------------------------------------------------------- */
int em_resample_events(const em_model_t *m, em_rng_t *g, const int32_t *initial,
                       const em_event_t *ev, int n, em_event_t *out, int cap) {
    int ni = m->n_initial;
    int32_t x[128];
    int k = 0, sec = 0; /* sec: 0-based absolute second of the next draw */
    for (int i = 0; i < ni; i++) x[i] = initial[i];
    for (int ii = 0; ii < n; ii++) {
        int holdtime = (int)ev[ii].dt;
        if (holdtime == 0) {
            if (k >= cap) return -2;
            out[k++] = ev[ii];                                                  /* :19-20 */
        } else {
            double delta_t = 0;
            for (int j = 1; j <= holdtime; j++) {
                /* changes = find(rand(size(rates)) < rates)  :24 -- n_initial uniforms */
                int first = 1;
                uint8_t ch[128];
                for (int v = 0; v < ni; v++) {
                    double rate = m->resample_rates[v];
                    double u;
                    if (g->mode == EM_RNG_MT19937 || rate > 0.0) u = em_rand(g, EM_SEC_RES, (uint32_t)v, (uint32_t)(sec + 1));
                    else u = 1.0; /* Philox mode: a draw compared with rate 0 never hits; skip it */
                    ch[v] = (u < rate);
                }
                delta_t += 1;                                                   /* :25 */
                for (int v = 0; v < ni; v++) {
                    if (!ch[v]) continue;
                    if (k >= cap) return -2;
                    out[k].dt = first ? delta_t : 0;                            /* :27 */
                    out[k].var = v + 1;
                    out[k].bin = x[v];
                    out[k].val = x[v];
                    out[k].kind = 1;
                    out[k].atime = sec + 1;
                    k++;
                    if (first) { first = 0; }
                }
                if (!first) delta_t = 0;                                        /* :28 */
                sec++;
            }
            if (k >= cap) return -2;
            out[k] = ev[ii];                                                    /* :31 */
            out[k].dt = delta_t;
            k++;
        }
        if (ev[ii].var > 0) x[ev[ii].var - 1] = ev[ii].bin;                     /* :33-35 */
    }
    return k;
}

/* ------------------------------------------------------------------------- */
/* a9  dediscretize.m:7-40 (scalar d; wrap is never requested by any caller)  */
/* ------------------------------------------------------------------------- */
double em_dediscretize_u(int d, const double *params, int n_params, int zero_bin, double u, int *used) {
    *used = 0;
    if (n_params == 0) return (double)d;          /* :7-10 */
    if (zero_bin != 0 && zero_bin == d) return 0; /* :24-25 */
    double a = params[d - 1];
    double b = params[d];
    *used = 1;
    return a + (b - a) * u;                       /* :39 */
}

static double dedisc(const em_model_t *m, em_rng_t *g, int var, int d, uint32_t section, uint32_t idx) {
    int np = m->bnd_len[var - 1];
    if (np == 0) return (double)d;
    if (m->zero_bins[var - 1] != 0 && m->zero_bins[var - 1] == d) return 0.0;
    const double *p = m->boundaries + m->bnd_off[var - 1];
    double u = em_rand(g, section, (section == EM_SEC_DEDISC_INIT || section == EM_SEC_GEOM_DEDISC) ? 0u : (uint32_t)(var - 1), idx);
    double a = p[d - 1], b = p[d];
    return a + (b - a) * u;
}

/* ------------------------------------------------------------------------- */
/* a7  dbn_hierarchical_sample.m:9-37                                         */
/* initial_bin: discrete initial sample; initial_val: dediscretised.          */
/* Returns number of event rows (incl. terminator) or <0.                     */
/* ------------------------------------------------------------------------- */
int em_dbn_hierarchical_sample(const em_model_t *m, em_rng_t *g, int sample_time, int per_step,
                               int32_t *initial_bin, double *initial_val,
                               em_event_t *events, int cap) {
    int ni = m->n_initial;
    em_event_t *raw = (em_event_t *)malloc(sizeof(em_event_t) * (size_t)cap);
    int n = em_dbn_sample(m, g, sample_time, per_step, initial_bin, raw, cap - 1);
    if (n < 0) { free(raw); return n; }
    /* terminator row [sample_time - sum(events(:,1)) 0 0]  :15-19 */
    double sum = 0;
    for (int i = 0; i < n; i++) sum += raw[i].dt;
    raw[n].dt = sample_time - sum; raw[n].var = 0; raw[n].bin = 0; raw[n].val = 0; raw[n].kind = 2; raw[n].atime = sample_time;
    n++;
    int k = em_resample_events(m, g, initial_bin, raw, n, events, cap);          /* :22 */
    free(raw);
    if (k < 0) return k;
    for (int ii = 1; ii <= ni; ii++) {                                           /* :25-31 */
        int r_rows = m->r_initial[ii - 1]; /* size(parms.N_initial{ii},1) */
        if (m->bnd_len[ii - 1] == r_rows - 2) {
            initial_val[ii - 1] = initial_bin[ii - 1];  /* :26 branch body is empty: left as the bin */
        } else {
            initial_val[ii - 1] = dedisc(m, g, ii, initial_bin[ii - 1], EM_SEC_DEDISC_INIT, (uint32_t)(ii - 1));
        }
    }
    for (int ii = 0; ii < k - 1; ii++) {                                         /* :33-37 */
        em_event_t *e = &events[ii];
        if (e->kind == 1) e->val = dedisc(m, g, e->var, e->bin, EM_SEC_DEDISC_RES, (uint32_t)e->atime);
        else              e->val = dedisc(m, g, e->var, e->bin, EM_SEC_DEDISC_TRANS, (uint32_t)e->atime);
    }
    return k;
}

/* ------------------------------------------------------------------------- */
/* a10 events2samples.m:9-26 on (value) and, in parallel, on (bin)            */
/* d: n_initial x T column-major (like MATLAB). Returns columns filled.       */
/* ------------------------------------------------------------------------- */
int em_events2samples(int n_initial, const double *initial_val, const int32_t *initial_bin,
                      const em_event_t *ev, int n, double *d_val, int32_t *d_bin, int T_cap) {
    double xv[128]; int32_t xb[128];
    for (int i = 0; i < n_initial; i++) { xv[i] = initial_val[i]; xb[i] = initial_bin ? initial_bin[i] : 0; }
    int t = 0, filled = 0;
    for (int e = 0; e < n; e++) {
        int delta_t = (int)ev[e].dt;
        int c0, c1;
        if (ev[e].var == 0) {
            t = t + 1;                       /* :17-18  d(:, t:t+delta_t-1) = x */
            c0 = t; c1 = t + delta_t - 1;
        } else {
            c0 = t + 1; c1 = t + delta_t;    /* :20-23  d(:, t+1:t+delta_t) = x; t = t+delta_t */
            if (delta_t > 0) t = t + delta_t;
        }
        for (int c = c0; c <= c1; c++) {
            if (c > T_cap) return -2;
            for (int i = 0; i < n_initial; i++) {
                d_val[(size_t)(c - 1) * n_initial + i] = xv[i];
                if (d_bin) d_bin[(size_t)(c - 1) * n_initial + i] = xb[i];
            }
            if (c > filled) filled = c;
        }
        if (ev[e].var != 0) {                /* :25 */
            xv[ev[e].var - 1] = ev[e].val;
            xb[ev[e].var - 1] = ev[e].bin;
        }
    }
    return filled;
}

/* ------------------------------------------------------------------------- */
/* a10 events2controls.m:11-31.  controls: rows x (1+n_dyn) row-major.        */
/* ------------------------------------------------------------------------- */
int em_events2controls(const em_model_t *m, const double *initial_val, const em_event_t *ev, int n, double *controls) {
    double x[128];
    int nd = m->n_dyn;
    for (int i = 0; i < m->n_initial; i++) x[i] = initial_val[i];
    double t = 0; int counter = 0;
    for (int e = 0; e < n; e++) {
        double delta_t = ev[e].dt;
        if (delta_t > 0) {
            controls[(size_t)counter * (1 + nd)] = t;
            for (int k = 0; k < nd; k++) controls[(size_t)counter * (1 + nd) + 1 + k] = x[m->temporal_map[2 * k] - 1];
            counter++;
            t += delta_t;
        }
        if (ev[e].var > 0) x[ev[e].var - 1] = ev[e].val;
    }
    return counter;
}

/* ------------------------------------------------------------------------- */
/* a11 @UncorEncounterModel/UncorEncounterModel.m:244-281, one sample with    */
/* its rejection loop.  idx*: 1-based variable ids (0 = absent).              */
/* layers: r_L x 2 row-major or NULL.  Returns event rows; *attempts_out =    */
/* number of attempts used (>=1); -3 if max_attempts exceeded.                */
/* ------------------------------------------------------------------------- */
typedef struct {
    int32_t idxL, idxV, idxDH, is_quantize500;
    const double *layers;
    int32_t max_attempts, per_step;
} em_uncor_opts_t;

static double round500(double num) { /* UncorEncounterModel.m:196 */
    return 500.0 * (floor(num / 500.0) + ((fmod(num, 500.0) > 250.0) ? 1.0 : 0.0));
}

int em_uncor_sample_one(const em_model_t *m, em_rng_t *g, int sample_time, const em_uncor_opts_t *o,
                        int32_t *initial_bin, double *initial_val, em_event_t *events, int cap, int32_t *attempts_out) {
    for (uint32_t attempt = 0; attempt < (uint32_t)o->max_attempts; attempt++) {
        g->attempt = attempt;
        int k = em_dbn_hierarchical_sample(m, g, sample_time, o->per_step, initial_bin, initial_val, events, cap); /* :253 */
        if (k < 0) return k;
        double h_ft;
        if (o->layers && o->idxL > 0) {                                             /* :259-260 */
            int b = (int)initial_val[o->idxL - 1];
            double lo = o->layers[2 * (b - 1)], hi = o->layers[2 * (b - 1) + 1];
            h_ft = lo + em_rand(g, EM_SEC_LAYER, 0, 0) * (hi - lo);
        } else {
            h_ft = o->idxL > 0 ? initial_val[o->idxL - 1] : 0.0;
        }
        if (o->idxDH > 0 && initial_val[o->idxDH - 1] == 0 && o->is_quantize500) h_ft = round500(h_ft); /* :266-268 */
        if ((o->layers || o->is_quantize500) && o->idxL > 0) initial_val[o->idxL - 1] = h_ft;          /* :270-272 */
        int good = 1;
        if (o->idxV > 0 && o->idxDH > 0)
            good = (initial_val[o->idxV - 1] * 1.68781 > fabs(initial_val[o->idxDH - 1]) / 60.0);      /* :275 */
        if (good) { *attempts_out = (int32_t)attempt + 1; return k; }
    }
    return -3;
}

/* ------------------------------------------------------------------------- */
/* Batch driver used by tests and by bench.py's cpu_baseline leg.             */
/* Philox: trajectory ii uses gidx = first_index + ii.  MT: one stream seeded */
/* once (UncorEncounterModel.m:213-216) that runs on across samples.          */
/* Outputs (any may be NULL):                                                 */
/*   init_bin  int32 [n][n_i], init_val f64 [n][n_i]                          */
/*   ev_*: event lists, ev_cap rows per sample, ev_count[n]                   */
/*   dense_bin uint8 [n][T][n_dyn], dense_val f64 [n][T][n_dyn]: the columns  */
/*     of events2samples restricted to temporal_map(:,1), bins alongside.     */
/* ------------------------------------------------------------------------- */
int64_t em_uncor_sample_batch(const em_model_t *m, int mode, uint64_t seed, uint64_t first_index, int64_t n,
                              int sample_time, const em_uncor_opts_t *o,
                              int32_t *init_bin, double *init_val,
                              double *ev_dt, int32_t *ev_var, int32_t *ev_bin, double *ev_val, int32_t *ev_count, int ev_cap,
                              uint8_t *dense_bin, double *dense_val, int32_t *attempts, uint64_t *n_draws_out) {
    em_rng_t g;
    em_rng_init(&g, mode, seed);
    int ni = m->n_initial, nd = m->n_dyn, T = sample_time;
    int cap = (ni + nd + 1) * T + 8;
    if (cap < 64) cap = 64;
    em_event_t *ev = (em_event_t *)malloc(sizeof(em_event_t) * (size_t)cap);
    int32_t ib[128]; double iv[128];
    double *dv = (double *)malloc(sizeof(double) * (size_t)ni * (size_t)(T + 1));
    int32_t *db = (int32_t *)malloc(sizeof(int32_t) * (size_t)ni * (size_t)(T + 1));
    int64_t rc = 0;
    for (int64_t ii = 0; ii < n; ii++) {
        g.gidx = first_index + (uint64_t)ii;
        int32_t att = 0;
        int k = em_uncor_sample_one(m, &g, T, o, ib, iv, ev, cap, &att);
        if (k < 0) { rc = k; break; }
        if (attempts) attempts[ii] = att;
        for (int i = 0; i < ni; i++) {
            if (init_bin) init_bin[ii * ni + i] = ib[i];
            if (init_val) init_val[ii * ni + i] = iv[i];
        }
        if (ev_count) {
            ev_count[ii] = k;
            if (k > ev_cap) { rc = -2; break; }
            for (int e = 0; e < k; e++) {
                ev_dt[ii * ev_cap + e] = ev[e].dt; ev_var[ii * ev_cap + e] = ev[e].var;
                ev_bin[ii * ev_cap + e] = ev[e].bin; ev_val[ii * ev_cap + e] = ev[e].val;
            }
        }
        if (dense_bin || dense_val) {
            /* the value row of L may have been overwritten by layers/quantize; dense covers dynamic vars only */
            int Tt = em_events2samples(ni, iv, ib, ev, k, dv, db, T);
            if (Tt != T) { rc = -4; break; }
            for (int c = 0; c < T; c++)
                for (int q = 0; q < nd; q++) {
                    int v = m->temporal_map[2 * q] - 1;
                    if (dense_bin) dense_bin[((size_t)ii * T + c) * nd + q] = (uint8_t)db[(size_t)c * ni + v];
                    if (dense_val) dense_val[((size_t)ii * T + c) * nd + q] = dv[(size_t)c * ni + v];
                }
        }
    }
    if (n_draws_out) *n_draws_out = g.n_draws;
    free(ev); free(dv); free(db);
    return rc;
}

/* The same batch split over `threads` OpenMP threads (Philox mode only: trajectories are keyed by
 * their global index, so any split gives identical results).  Used by bench.py's all-cores
 * cpu_baseline leg.  Dense outputs only. */
int64_t em_uncor_sample_batch_mt(const em_model_t *m, uint64_t seed, uint64_t first_index, int64_t n,
                                 int sample_time, const em_uncor_opts_t *o, int threads,
                                 uint8_t *dense_bin, double *dense_val, int32_t *init_bin, double *init_val) {
    int64_t rc_all = 0;
    const int ni = m->n_initial, nd = m->n_dyn, T = sample_time;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int t = 0; t < threads; t++) {
        const int64_t lo = n * t / threads, hi = n * (t + 1) / threads;
        int64_t rc = em_uncor_sample_batch(m, EM_RNG_PHILOX, seed, first_index + (uint64_t)lo, hi - lo, T, o,
                                           init_bin ? init_bin + lo * ni : NULL, init_val ? init_val + lo * ni : NULL,
                                           NULL, NULL, NULL, NULL, NULL, 0,
                                           dense_bin ? dense_bin + (size_t)lo * T * nd : NULL,
                                           dense_val ? dense_val + (size_t)lo * T * nd : NULL, NULL, NULL);
        if (rc != 0) {
#pragma omp critical
            rc_all = rc;
        }
    }
    return rc_all;
}

/* Throughput form of the same batch for bench.py's all-cores cpu_baseline leg: every thread samples chunks of `chunk`
 * trajectories into ITS OWN dense buffers (allocated and first touched by the thread, reused chunk after chunk), so the
 * threads do not contend on page faults of one shared multi-gigabyte output the way em_uncor_sample_batch_mt's callers made
 * them (256 threads scaled 7.6x).  The arithmetic per trajectory is the same; *checksum (sum of all dense bins) shows the
 * work was done and equals the sum over em_uncor_sample_batch's dense_bin for the same range. */
int64_t em_uncor_sample_throughput_mt(const em_model_t *m, uint64_t seed, uint64_t first_index, int64_t n,
                                      int sample_time, const em_uncor_opts_t *o, int threads, int64_t chunk, uint64_t *checksum) {
    int64_t rc_all = 0;
    uint64_t sum_all = 0;
    const int nd = m->n_dyn, T = sample_time;
    if (chunk < 1) chunk = 256;
    const int64_t n_chunks = (n + chunk - 1) / chunk;
#pragma omp parallel num_threads(threads) reduction(+ : sum_all)
    {
        uint8_t *db = (uint8_t *)malloc((size_t)chunk * (size_t)T * (size_t)nd);
        double *dv = (double *)malloc(sizeof(double) * (size_t)chunk * (size_t)T * (size_t)nd);
#pragma omp for schedule(dynamic, 1)
        for (int64_t c = 0; c < n_chunks; c++) {
            const int64_t lo = c * chunk, cnt = (lo + chunk <= n) ? chunk : n - lo;
            int64_t rc = em_uncor_sample_batch(m, EM_RNG_PHILOX, seed, first_index + (uint64_t)lo, cnt, T, o, NULL, NULL,
                                               NULL, NULL, NULL, NULL, NULL, 0, db, dv, NULL, NULL);
            if (rc != 0) {
#pragma omp critical
                rc_all = rc;
            } else {
                uint64_t s = 0;
                for (size_t q = 0; q < (size_t)cnt * (size_t)T * (size_t)nd; q++) s += db[q];
                sum_all += s;
            }
        }
        free(db); free(dv);
    }
    if (checksum) *checksum = sum_all;
    return rc_all;
}

/* ------------------------------------------------------------------------- */
/* a14 @CorTerminalModel/sample.m:29-77 -- geometry BN, one sample with its   */
/* rejection loop: bn_sample + dediscretize (:34-42), bounds box (:45-53),    */
/* speed limits (:64-70).  bounds_sample: n_i x 2 row-major or NULL.          */
/* ------------------------------------------------------------------------- */
typedef struct {
    const double *bounds_sample;
    int32_t idx_own_speed, idx_int_speed;   /* 1-based, 0 = no speed check */
    double min1, max1, min2, max2;
    int32_t max_attempts, _pad;
} em_geom_opts_t;

int64_t em_geom_sample_batch(const em_model_t *m, int mode, uint64_t seed, uint64_t first_index, int64_t n,
                             const em_geom_opts_t *o, int32_t *out_bin, double *out_val, int32_t *attempts) {
    em_rng_t g;
    em_rng_init(&g, mode, seed);
    int ni = m->n_initial;
    int32_t S[128]; double v[128];
    for (int64_t ii = 0; ii < n; ii++) {
        g.gidx = first_index + (uint64_t)ii;
        int good = 0;
        uint32_t attempt;
        for (attempt = 0; attempt < (uint32_t)o->max_attempts && !good; attempt++) {
            g.attempt = attempt;
            if (em_bn_sample(m, &g, m->r_initial, S) != 0) return -1;
            for (int kk = 1; kk <= ni; kk++) v[kk - 1] = dedisc(m, &g, kk, S[kk - 1], EM_SEC_GEOM_DEDISC, (uint32_t)(kk - 1));
            good = 1;
            if (o->bounds_sample)
                for (int kk = 0; kk < ni; kk++)
                    if (!(v[kk] >= o->bounds_sample[2 * kk] && v[kk] <= o->bounds_sample[2 * kk + 1])) good = 0;
            if (good && o->idx_own_speed > 0) {
                double s1 = v[o->idx_own_speed - 1], s2 = v[o->idx_int_speed - 1];
                if (!(s1 <= o->max1 && s1 >= o->min1 && s2 <= o->max2 && s2 >= o->min2)) good = 0;
            }
        }
        if (!good) return -3;
        if (attempts) attempts[ii] = (int32_t)attempt;
        for (int i = 0; i < ni; i++) { if (out_bin) out_bin[ii * ni + i] = S[i]; if (out_val) out_val[ii * ni + i] = v[i]; }
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Plain dbn_sample batch (no resample / dediscretize): raw events            */
/* ------------------------------------------------------------------------- */
int64_t em_dbn_sample_batch(const em_model_t *m, int mode, uint64_t seed, uint64_t first_index, int64_t n,
                            int t_max, int per_step, int32_t *init_bin,
                            double *ev_dt, int32_t *ev_var, int32_t *ev_bin, int32_t *ev_count, int ev_cap) {
    em_rng_t g;
    em_rng_init(&g, mode, seed);
    int ni = m->n_initial;
    int cap = (ni + 1) * t_max + 8;
    em_event_t *ev = (em_event_t *)malloc(sizeof(em_event_t) * (size_t)cap);
    int32_t ib[128];
    int64_t rc = 0;
    for (int64_t ii = 0; ii < n; ii++) {
        g.gidx = first_index + (uint64_t)ii;
        int k = em_dbn_sample(m, &g, t_max, per_step, ib, ev, cap);
        if (k < 0) { rc = k; break; }
        for (int i = 0; i < ni; i++) init_bin[ii * ni + i] = ib[i];
        ev_count[ii] = k;
        if (k > ev_cap) { rc = -2; break; }
        for (int e = 0; e < k; e++) { ev_dt[ii * ev_cap + e] = ev[e].dt; ev_var[ii * ev_cap + e] = ev[e].var; ev_bin[ii * ev_cap + e] = ev[e].bin; }
    }
    free(ev);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* Decision margins (test infrastructure for the .track parity tests): while em_margin_ptr points at a number, every        */
/* discrete decision of the track paths -- a discretize, a limit test, a rounding, an argmin -- records how close it came   */
/* to going the other way, |value - threshold| / scale, and the smallest such distance of the attempt is left there.        */
/* A GPU result that parts from this restatement at some attempt is only excusable when that number is at rounding level.   */
/* ------------------------------------------------------------------------- */
static _Thread_local double *em_margin_ptr = NULL;
static void em_note_s(double v, double thr, double scale) {
    if (!em_margin_ptr || isnan(v) || isnan(thr) || isinf(v) || isinf(thr)) return;
    if (v == thr) return;   /* an exact tie is structural (a speed clamped to its limit, an altitude of exactly 0): the same constant on both sides */
    const double s = fmax(fmax(fabs(v), fabs(thr)), scale);
    const double r = s > 0 ? fabs(v - thr) / s : 0.0;
    if (r < *em_margin_ptr) *em_margin_ptr = r;
}
static void em_note(double v, double thr) { em_note_s(v, thr, 0.0); }

/* ------------------------------------------------------------------------- */
/* a16 discretize_bayes.m:14-22                                               */
/* ------------------------------------------------------------------------- */
int em_discretize_bayes(double x, const double *thresholds, int n) {
    int b = n + 1;
    if (!(x >= thresholds[n - 1]))
        for (int i = 0; i < n; i++) if (x < thresholds[i]) { b = i + 1; break; }
    if (em_margin_ptr) {   /* the two cut points that bound the chosen bin */
        if (b >= 2) em_note(x, thresholds[b - 2]);
        if (b <= n) em_note(x, thresholds[b - 1]);
    }
    return b;
}

/* a17 bn_dirichlet_prior.m:18-37 ('dbe' => 1/(r*q); numeric => constant), one node */
void em_dirichlet_prior_node(int r, int64_t q, int is_dbe, double prior, double *alpha) {
    double p = is_dbe ? 1.0 / ((double)r * (double)q) : prior;
    for (int64_t i = 0; i < (int64_t)r * q; i++) alpha[i] = p;
}

/* a17 setTransitionPriors.m:20-27, one node: alpha (r_jj x q) column-major,
 * alpha(kk, n*(kk-1)+1 : n*kk) = prior with n = q / r_jj */
void em_transition_prior_node(int r_jj, int64_t q, double prior, double *alpha) {
    int64_t n = q / r_jj;
    for (int64_t i = 0; i < (int64_t)r_jj * q; i++) alpha[i] = 0;
    for (int kk = 1; kk <= r_jj; kk++)
        for (int64_t c = n * (kk - 1) + 1; c <= n * kk; c++)
            alpha[(c - 1) * r_jj + (kk - 1)] = prior;
}

/* ------------------------------------------------------------------------- */
/* a15 PropagateTrajectory  @CorTerminalModel/createEncounter.m:93-265 with   */
/* CreateStartDistribution :268-294 and CheckTrajectoryConditions :296-329.   */
/* The trajectory model m has the 6 initial variables {intent, distance,       */
/* bearing, heading, altitude, speed} (:107-109,:293) and A_transition must    */
/* hold setTransitionPriors(G, r, temporal_map, 1) (:129).  em-core's          */
/* local_smooth (:88-89) is NOT applied (un-vendored dependency).             */
/* out: rows [t_s x_nm y_nm z_ft heading_deg v_ft_s]; returns the row count.  */
/* ------------------------------------------------------------------------- */
typedef struct { double minVel_ft_s, maxVel_ft_s, maxTurnRate_deg_s, maxAltitude_ft, maxVertRate_ft_s; } em_dynlims_t;

static double em_wrapTo360(double lon) {   /* Mapping Toolbox wrapTo360 */
    const int positive = lon > 0;
    lon = lon - floor(lon / 360.0) * 360.0; /* mod(lon, 360) */
    if (lon == 0 && positive) lon = 360.0;
    return lon;
}
static double em_atan2d(double y, double x) { return atan2(y, x) * (180.0 / 3.14159265358979323846); }
static void em_sincosd(double deg, double *s, double *c) {
    /* sind / cosd the way MATLAB's own sind.m / cosd.m reduce the argument: IN DEGREES, to [-45, 45] around the nearest
     * multiple of 90 (n = round(x/90); x = x - n*90; m = mod(n, 4)), then sin / cos of pi/180*x with the quadrant's sign
     * and swap.  Exact at every multiple of 90; the reduced argument is what the radian functions see. */
    const double n = round(deg / 90.0);
    const double x = (3.14159265358979323846 / 180.0) * (deg - n * 90.0);
    double m = fmod(n, 4.0); if (m < 0) m += 4.0;
    const double sx = sin(x), cx = cos(x);
    if (m == 0) { *s = sx; *c = cx; }
    else if (m == 1) { *s = cx; *c = -sx; }
    else if (m == 2) { *s = -sx; *c = -cx; }
    else { *s = -cx; *c = sx; }
}
static double em_round2(double x) { return round(x * 100.0) / 100.0; } /* round(x, 2) */
static double em_sign(double x) { return (x > 0) - (x < 0); }

int em_propagate_trajectory(const em_model_t *m, em_rng_t *g, int role, int is_ownship, double dt_s,
                            double x0_nm, double y0_nm, double z0_ft, double v0_ft_s, double heading0_deg, int intent,
                            double tmax_s, const em_dynlims_t *dl, int max_resample, double *out, int cap) {
    const int ni = m->n_initial, nt = m->n_transition;
    const int IDX_DIST = 2, IDX_BEAR = 3, IDX_HEAD = 4, IDX_ALT = 5, IDX_SPD = 6;
    if (ni != 6 || m->n_dyn != 3) return -1;
    /* "the attempt's first dediscretize draw" (slot map, round 5) is defined by the order heading, altitude, speed: the rows of the temporal map
     * are walked below, so they must be those three variables in that order (em_read.m:158-177 builds the map in variable order) */
    if (m->temporal_map[0] != IDX_HEAD || m->temporal_map[2] != IDX_ALT || m->temporal_map[4] != IDX_SPD) return -1;
    const double *bnd[7]; int nb[7];
    for (int v = 1; v <= 6; v++) { bnd[v] = m->boundaries + m->bnd_off[v - 1]; nb[v] = m->bnd_len[v - 1]; }
    /* discreteValidAlt / discreteValidV  (:121-127) */
    int alt_last = 0, spd_first = 0, spd_last = 0;
    for (int q = 0; q < nb[IDX_ALT]; q++) if (bnd[IDX_ALT][q] <= dl->maxAltitude_ft) alt_last = q + 1;
    for (int q = 0; q < nb[IDX_SPD]; q++) { if (!(bnd[IDX_SPD][q] >= dl->minVel_ft_s)) spd_first = q + 1; if (bnd[IDX_SPD][q] <= dl->maxVel_ft_s) spd_last = q + 1; }
    const double bounds_dist_hi = bnd[IDX_DIST][nb[IDX_DIST] - 1]; /* mdl.bounds_initial(idx.dist, 2) = max(boundaries) */

    double xy[2] = {x0_nm, y0_nm}, sh, chh;
    em_sincosd(heading0_deg, &sh, &chh);
    double v[2] = {chh * v0_ft_s - sh * 0.0, sh * v0_ft_s + chh * 0.0}; /* rotationmatrix(heading0)*[v0;0]  :145 */
    double z_ft = z0_ft, heading_deg = heading0_deg, t_s = 0;
    double prev_z_rec = 0;
    int ii = 1, rows = 0;
    int is_resample = 1;
    double w[EM_MAX_R];
    while (is_resample) {                                                           /* :160 */
        if (rows >= cap) return -2;
        double *row = out + (size_t)rows * 6;
        row[0] = t_s; row[1] = xy[0]; row[2] = xy[1]; row[3] = z_ft; row[4] = heading_deg; row[5] = sqrt(v[0] * v[0] + v[1] * v[1]);
        xy[0] += v[0] * dt_s / 6076.1154855643;                                     /* :171-173 */
        xy[1] += v[1] * dt_s / 6076.1154855643;
        const double curr_hdg_deg = em_wrapTo360(em_atan2d(v[1], v[0]));            /* :176-177 */
        row[4] = curr_hdg_deg;
        if (ii > 1) {                                                               /* :180-184 */
            const double alt_diff_ft = z_ft - prev_z_rec;
            row[3] = prev_z_rec + em_sign(alt_diff_ft) * fmin(dl->maxVertRate_ft_s, fabs(alt_diff_ft));
        }
        prev_z_rec = row[3];
        rows++;
        /* CreateStartDistribution :268-294 */
        int32_t start[6];
        start[0] = intent;
        start[1] = em_discretize_bayes(sqrt(xy[0] * xy[0] + xy[1] * xy[1]), bnd[IDX_DIST] + 1, nb[IDX_DIST] - 2);
        start[2] = em_discretize_bayes(em_wrapTo360(em_atan2d(xy[1], xy[0])), bnd[IDX_BEAR] + 1, nb[IDX_BEAR] - 2);
        const int heading_discrete = em_discretize_bayes(heading_deg, bnd[IDX_HEAD] + 1, nb[IDX_HEAD] - 2);
        start[3] = heading_discrete;
        start[4] = em_discretize_bayes(z_ft, bnd[IDX_ALT] + 1, nb[IDX_ALT] - 2);
        start[5] = em_discretize_bayes(sqrt(v[0] * v[0] + v[1] * v[1]), bnd[IDX_SPD] + 1, nb[IDX_SPD] - 2);
        int att = 0;
        is_resample = 1;
        while (is_resample) {                                                       /* :192 */
            if (att >= max_resample) return -3;
            g->attempt = (uint32_t)role + 4u * (uint32_t)att;
            att++;
            /* dbn_sample(mdl, prior_initial, prior_transition, 2, start) (:193): every initial variable is
             * preset, so bn_sample draws nothing; fast branch: rand(2,1) per dynamic variable, row 2 used */
            int32_t x[16], newbin[3];
            for (int q = 0; q < 6; q++) x[q] = start[q];
            for (int q = 6; q < nt; q++) x[q] = 0;
            for (int k = 0; k < 3; k++) {
                const int tv = m->temporal_map[2 * k + 1];
                const int r = m->r_transition[tv - 1];
                const int64_t j = parent_config(m->G_transition, nt, m->r_transition, x, tv);
                column_weights(m->N_transition, m->A_transition, m->off_transition[tv - 1], r, j, w);
                if (g->mode == EM_RNG_MT19937) (void)em_rand(g, EM_SEC_TERM_TRANS, 0u, 0);
                newbin[k] = em_select_random_r(w, r, em_rand(g, EM_SEC_TERM_TRANS, 0u, 4u * (uint32_t)ii + (uint32_t)k));
            }
            is_resample = 0;
            /* The dediscretize draws of one attempt (Philox slot map, round 5): the FIRST one made takes the fourth word of the attempt's
             * TERM_TRANS block (the block's words 0-2 are the three transition draws; word 3 was unused), any further one -- two events in one
             * step: 0.5 % of the steps -- its own word of the TERM_DEDISC block as before.  One Philox call per attempt for almost every step
             * instead of two.  (MT19937 mode draws sequentially and does not notice.) */
            int n_dedisc = 0;
#define EM_TERM_DEDISC_RAND(k_) ((n_dedisc++ == 0) ? em_rand(g, EM_SEC_TERM_TRANS, 0u, 4u * (uint32_t)ii + 3u) : em_rand(g, EM_SEC_TERM_DEDISC, 0u, 4u * (uint32_t)ii + (uint32_t)(k_)))
            /* events rows in ascending variable id: 4 heading, 5 altitude, 6 speed (:198-238) */
            for (int k = 0; k < 3 && !is_resample; k++) {
                const int var = m->temporal_map[2 * k];
                if (newbin[k] == start[var - 1]) continue;   /* no event (dbn_sample.m:151-161) */
                const int d = newbin[k];
                if (var == IDX_HEAD) {
                    if (d != heading_discrete) {
                        const double u = EM_TERM_DEDISC_RAND(k);
                        heading_deg = bnd[var][d - 1] + (bnd[var][d] - bnd[var][d - 1]) * u;
                    }
                } else if (var == IDX_ALT) {
                    if (alt_last >= 1 && d >= 1 && d <= alt_last) {      /* 1:[] is empty in MATLAB */
                        const double u = EM_TERM_DEDISC_RAND(k);
                        z_ft = bnd[var][d - 1] + (bnd[var][d] - bnd[var][d - 1]) * u;
                    } else is_resample = 1;
                } else if (var == IDX_SPD) {
                    if (spd_first >= 1 && d >= spd_first && d <= spd_last) { /* []:1:e is empty: no speed event is ever valid */
                        const double u = EM_TERM_DEDISC_RAND(k);
                        double s1 = bnd[var][d - 1] + (bnd[var][d] - bnd[var][d - 1]) * u;
                        if (s1 < dl->minVel_ft_s) s1 = dl->minVel_ft_s;
                        if (s1 > dl->maxVel_ft_s) s1 = dl->maxVel_ft_s;
                        em_sincosd(heading_deg, &sh, &chh);
                        v[0] = chh * s1; v[1] = sh * s1;                              /* rotationmatrix(heading_deg)*[v;0] */
                    } else is_resample = 1;
                }
            }
        }
#undef EM_TERM_DEDISC_RAND
        /* turn toward the desired heading at no more than maxTurnRate (:241-256) */
        const double turn1 = em_round2(heading_deg - curr_hdg_deg);
        const double delta = fmin(fabs(turn1), dl->maxTurnRate_deg_s) * em_sign(turn1);
        em_sincosd(delta, &sh, &chh);
        const double vx = chh * v[0] - sh * v[1], vy = sh * v[0] + chh * v[1];
        v[0] = vx; v[1] = vy;
        t_s += dt_s; ii++;
        /* CheckTrajectoryConditions :296-329 */
        const double d_nm = sqrt(xy[0] * xy[0] + xy[1] * xy[1]);
        const int violate_time = fabs(t_s) > tmax_s;
        const int violate_far = d_nm > bounds_dist_hi;
        const int violate_intent = (intent == 1 || intent == 2) ? (d_nm <= 0.25) : 0;
        const int violate_ownship = is_ownship && xy[1] > 0.25;
        is_resample = !(violate_time || violate_far || violate_intent || violate_ownship);
    }
    return rows;
}

/* Batch driver: encounter e, role = 2*aircraft + (backward), gidx = first_index + e.
 * geo: n x 12 doubles [x0 y0 z0 v0 heading0 intent] for aircraft 1 then 2 (what createEncounter.m:41-49
 * derives from the geometry sample).  models: 4 per-role model pointers are chosen by the caller
 * through model_of[e*4 + role] (index into the models array).  out: [n][4][cap][6], rows[n][4]. */
int64_t em_propagate_batch(const em_model_t *const *models, const int32_t *model_of, int mode, uint64_t seed, uint64_t first_index,
                           int64_t n, const double *geo, const em_dynlims_t *dl /* [2] */, double tmax_s, int max_resample,
                           double *out, int32_t *rows, int cap) {
    em_rng_t g;
    em_rng_init(&g, mode, seed);
    for (int64_t e = 0; e < n; e++) {
        g.gidx = first_index + (uint64_t)e;
        for (int role = 0; role < 4; role++) {
            const int ac = role >> 1, bck = role & 1;
            const double *q = geo + e * 12 + ac * 6;
            int r = em_propagate_trajectory(models[model_of[e * 4 + role]], &g, role, ac == 0, bck ? -1.0 : 1.0,
                                            q[0], q[1], q[2], q[3], q[4], (int)q[5], tmax_s, &dl[ac], max_resample,
                                            out + ((size_t)(e * 4 + role) * cap) * 6, cap);
            if (r < 0) return r;
            rows[e * 4 + role] = r;
        }
    }
    return 0;
}

/* bench.py's CPU leg for config 5: the work of em_propagate_batch on `threads` threads.  Philox mode only (an encounter's draws depend
 * on its global index alone, so encounters can be dealt to threads); every thread overwrites ONE private encounter's worth of tracks
 * (4 x cap x 6 doubles) -- the leg measures the rate, the results are the ones em_propagate_batch returns.  Returns the track rows
 * produced (sum over the batch), or a negative error. */
int64_t em_propagate_throughput_mt(const em_model_t *const *models, const int32_t *model_of, uint64_t seed, uint64_t first_index,
                                   int64_t n, const double *geo, const em_dynlims_t *dl /* [2] */, double tmax_s, int max_resample,
                                   int threads, int cap) {
    int64_t total = 0;
    int err = 0;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads) reduction(+ : total)
    {
        em_rng_t g;
        em_rng_init(&g, EM_RNG_PHILOX, seed);
        double *buf = (double *)malloc((size_t)4 * cap * 6 * sizeof(double));
#pragma omp for schedule(dynamic, 16)
        for (int64_t e = 0; e < n; e++) {
            if (!buf || err) continue;
            g.gidx = first_index + (uint64_t)e;
            for (int role = 0; role < 4; role++) {
                const int ac = role >> 1, bck = role & 1;
                const double *q = geo + e * 12 + ac * 6;
                const int r = em_propagate_trajectory(models[model_of[e * 4 + role]], &g, role, ac == 0, bck ? -1.0 : 1.0,
                                                      q[0], q[1], q[2], q[3], q[4], (int)q[5], tmax_s, &dl[ac], max_resample,
                                                      buf + (size_t)role * cap * 6, cap);
                if (r < 0) {
#pragma omp atomic write
                    err = r;
                } else total += r;
            }
        }
        if (!buf) {
#pragma omp atomic write
            err = -1;
        }
        free(buf);
    }
    return err ? err : total;
}

/* sample2track.m:183-243 -- the 1 Hz dead-reckoning track and its rejection tests.
 * alt0/speed0 [n] and updates [n][T][3] (vertical rate, acceleration, turn rate) in model units, as
 * read from initial.txt / transition.txt; unit ratios as :113-123; min/max speed = boundaries{v}([1 end]).
 * xyz [n][T+1][3], flags[n] (bit 0 CFIT :234-237, bit 1 speed :240), vmm [n][2] = min, max speed. */
void em_sample2track_batch(int64_t n, int T, double ur_speed, double ur_vertrate, double ur_heading, double min_speed, double max_speed,
                           const double *alt0, const double *speed0, const double *updates, double *xyz, uint8_t *flags, double *vmm) {
    const double vmin = min_speed * ur_speed, vmax = max_speed * ur_speed;         /* :138-139 */
    for (int64_t i = 0; i < n; i++) {
        double x = 0, y = 0, z = alt0[i], sp = speed0[i] * ur_speed, hd = 0;      /* :184-189, :126 */
        double lo = sp, hi = sp;
        unsigned fl = 0;
        if (z < 0) fl |= 1u;
        if (sp <= vmin || sp >= vmax) fl |= 2u;
        double *o = xyz ? xyz + (size_t)i * (T + 1) * 3 : NULL;
        if (o) { o[0] = x; o[1] = y; o[2] = z; }
        for (int t = 0; t < T; t++) {
            const double *u = updates + ((size_t)i * T + t) * 3;
            const double dz = u[0] * ur_vertrate, dsp = u[1] * ur_speed, dhd = u[2] * ur_heading;  /* :131-133 */
            double sh, ch;
            em_sincosd(hd, &sh, &ch);
            const double xn = x + sp * ch, yn = y + sp * sh;                        /* :211-212 use the previous speed and heading */
            z = z + dz; sp = sp + dsp; hd = hd + dhd;                               /* :207-209 */
            x = xn; y = yn;
            if (z < 0) fl |= 1u;
            if (sp <= vmin || sp >= vmax) fl |= 2u;
            if (sp < lo) lo = sp;
            if (sp > hi) hi = sp;
            if (o) { o[(t + 1) * 3] = x; o[(t + 1) * 3 + 1] = y; o[(t + 1) * 3 + 2] = z; }
        }
        if (flags) flags[i] = (uint8_t)fl;
        if (vmm) { vmm[2 * i] = lo; vmm[2 * i + 1] = hi; }
    }
}

/* MT19937 helpers for tests */
void em_mt_doubles(uint32_t seed, int n, double *out) {
    em_mt_t s; mt_seed(&s, seed);
    for (int i = 0; i < n; i++) out[i] = mt_res53(&s);
}
int em_sizeof_model(void) { return (int)sizeof(em_model_t); }

/* test hook: the cosd / sind restatement on a table of angles (tests/golden/make_matlab_goldens.py) */
void em_sincosd_table(const double *deg, int n, double *s, double *c) {
    for (int i = 0; i < n; i++) em_sincosd(deg[i], &s[i], &c[i]);
}

/* ------------------------------------------------------------------------- */
/* f1  UncorEncounterModel.track  (@UncorEncounterModel/UncorEncounterModel.m:318-471, coordSys 'NEU')       */
/*     + @UncorEncounterModel/getDynamicLimits.m:1-130.                                                       */
/*                                                                                                            */
/* The reference integrates the sampled controls with em-core's run_dynamics_fast mex and differentiates the */
/* altitude with em-core's computeVerticalRate.  em-core is NOT vendored by the reference and is absent here: */
/* "dynamics unpinned".  What replaces them is a point-mass model built from the quantities the reference     */
/* hands to run_dynamics_fast (ic = [v n e h psi theta phi a], :443; dyn = [v_low v_high dh_min dh_max qmax    */
/* rmax], :414; controls = [t hdot psidot a], :291-297), stated once here and mirrored line by line by the    */
/* HIP kernel (csrc/emgpu_kernels_utrack.hip):                                                                */
/*   dt = 0.1 s, g = 32.2 ft/s^2 (the constant of :440).  Per step, with the control row active at t:         */
/*     a      = 0 when it would drive v beyond [v_low, v_high]                                               */
/*     hd     = min(max(hdot, dh_min), dh_max)                                                                */
/*     theta += clamp((asin(clamp(hd / v, -1, 1)) - theta) / dt, -qmax, qmax) * dt      pitch follows the      */
/*     phi   += clamp((atan(v * psidot / g) - phi) / dt, -rmax, rmax) * dt              commanded climb / turn */
/*     n += v cos(theta) cos(psi) dt;  e += v cos(theta) sin(psi) dt;  h += v sin(theta) dt                    */
/*     psi += g tan(phi) / v * dt;  v = min(max(v + a dt, v_low), v_high)                                     */
/*   results.time = 0 : 0.1 : T.  computeVerticalRate(up(is_sec), time(is_sec)) := forward difference of the   */
/*   1 Hz altitudes, the last value repeated.                                                                 */
/* Everything around the dynamics follows the cited lines.                                                    */
/* ------------------------------------------------------------------------- */
typedef struct {
    int32_t idxG, idxA, idxL, idxV, idxDV, idxDH, idxDPsi; /* 1-based, 0 = absent (:385-391)      */
    int32_t is_rotorcraft;                                  /* :181-185                            */
} em_track_vars_t;

/* getDynamicLimits.m:1-130.  initial: dediscretised values (bins for categorical variables); up_min..speed_max */
/* in feet and ft/s over the WHOLE result (10 Hz).  out: minVel_ft_s, maxVel_ft_s, maxVertRate_ft_s.            */
static const double *em_N_init(const em_model_t *m, int var1) { return m->N_initial + m->off_initial[var1 - 1]; }
static int64_t em_q_init(const em_model_t *m, int var1) {
    int64_t q = 1;
    for (int p = 0; p < m->n_initial; p++) if (m->G_initial[(size_t)p * m->n_initial + (var1 - 1)]) q *= m->r_initial[p];
    return q;
}
static int em_cutpoints(const em_model_t *m, int var1, double *cut) { /* em_read.m:130-136: boundaries(2:end-1), or 2:n */
    int len = m->bnd_len[var1 - 1];
    if (len == 0) { int n = m->r_initial[var1 - 1]; for (int i = 0; i < n - 1; i++) cut[i] = 2 + i; return n - 1; }
    for (int i = 1; i < len - 1; i++) cut[i - 1] = m->boundaries[m->bnd_off[var1 - 1] + i];
    return len - 2;
}
void em_uncor_dynamic_limits(const em_model_t *m, const em_track_vars_t *tv, const double *initial,
                             double up_min, double up_max, double speed_min, double speed_max, double out[3]) {
    const int idx_G = tv->idxG, idx_A = tv->idxA, idx_L = tv->idxL, idx_V = tv->idxV, idx_DH = tv->idxDH;
    const int rV = m->r_initial[idx_V - 1], rDH = m->r_initial[idx_DH - 1];
    double v_initial[EM_MAX_R], dh_initial[EM_MAX_R];
    for (int i = 0; i < rV; i++) v_initial[i] = 0;
    for (int i = 0; i < rDH; i++) dh_initial[i] = 0;
    const double *NV = em_N_init(m, idx_V), *NDH = em_N_init(m, idx_DH);
    const int64_t qV = em_q_init(m, idx_V), qDH = em_q_init(m, idx_DH);
    const int is_idx = idx_G > 0 && idx_A > 0 && idx_L > 0 && idx_V > 0 && idx_DH > 0;                   /* :14 */
    if (is_idx && idx_G == 1 && idx_A == 2 && idx_L == 3 && idx_V == 4 && idx_DH == 6) {                  /* :17 */
        double cut[EM_MAX_R];
        const int rG = m->r_initial[0], rA = m->r_initial[1], rL = m->r_initial[2];
        int nc, dG, dA, dL0, dL1, dV0, dV1;
        if (m->bnd_len[idx_G - 1] == 0) dG = (int)initial[idx_G - 1];                                     /* :20-24 */
        else { nc = em_cutpoints(m, idx_G, cut); dG = em_discretize_bayes(initial[idx_G - 1], cut, nc); }
        if (m->bnd_len[idx_A - 1] == 0) dA = (int)initial[idx_A - 1];                                     /* :27-31 */
        else { nc = em_cutpoints(m, idx_A, cut); dA = em_discretize_bayes(initial[idx_A - 1], cut, nc); }
        if (m->bnd_len[idx_L - 1] == 0) dL0 = dL1 = (int)initial[idx_L - 1];                              /* :34-39 */
        else { nc = em_cutpoints(m, idx_L, cut); dL0 = em_discretize_bayes(up_min, cut, nc); dL1 = em_discretize_bayes(up_max, cut, nc); }
        if (m->bnd_len[idx_V - 1] == 0) dV0 = dV1 = (int)initial[idx_V - 1];                              /* :45-51 */
        else {
            nc = em_cutpoints(m, idx_V, cut);
            dV0 = em_discretize_bayes(speed_min * 0.592484, cut, nc); dV1 = em_discretize_bayes(speed_max * 0.592484, cut, nc);
        }
        /* column j (1-based) of N{V} survives (:, dG:rG:end), then (:, dA:rA:end), then (:, unique(dL)) (:57-66) */
        const int64_t nGA_V = qV / rG / rA;          /* columns left after the two slices */
        for (int dl = dL0; dl <= dL1; dl++) {
            if (dl < 1 || dl > nGA_V) continue;      /* MATLAB would raise an index error here: none of the shipped shapes does */
            const int64_t col = (dG - 1) + (int64_t)rG * ((dA - 1) + (int64_t)rA * (dl - 1));
            for (int i = 0; i < rV; i++) v_initial[i] += NV[col * rV + i];
        }
        /* N{DH}: G, A sliced, then summed over di = unique(dL) of (:, di:rL:end), then over unique(dV) of (:, di:rV:end) (:69-79) */
        const int64_t nGA_DH = qDH / rG / rA, nL = nGA_DH / rL, nLV = nL / rV;
        for (int dl = dL0; dl <= dL1; dl++)
            for (int dv = dV0; dv <= dV1; dv++)
                for (int64_t rest = 0; rest < nLV; rest++) {
                    const int64_t c3 = (dl - 1) + (int64_t)rL * ((dv - 1) + (int64_t)rV * rest); /* column within the G,A slice */
                    const int64_t col = (dG - 1) + (int64_t)rG * ((dA - 1) + (int64_t)rA * c3);
                    for (int i = 0; i < rDH; i++) dh_initial[i] += NDH[col * rDH + i];
                }
    } else {                                                                                              /* :85-88 */
        for (int64_t c = 0; c < qV; c++) for (int i = 0; i < rV; i++) v_initial[i] += NV[c * rV + i];
        for (int64_t c = 0; c < qDH; c++) for (int i = 0; i < rDH; i++) dh_initial[i] += NDH[c * rDH + i];
    }
    /* :94-103 */
    double tot = 0, cs = 0;
    for (int i = 0; i < rV; i++) tot += v_initial[i];
    int k_min = 0, k_max = 0;   /* find(cs >= prct, 1, 'first'); 0 = empty */
    for (int i = 0; i < rV; i++) {
        cs += 100.0 * v_initial[i] / tot;
        if (!k_min && cs >= 1.0) k_min = i + 1;
        if (!k_max && cs >= 99.0) k_max = i + 1;
    }
    const double *bV = m->boundaries + m->bnd_off[idx_V - 1], *bDH = m->boundaries + m->bnd_off[idx_DH - 1];
    double min_speed = bV[k_min] * 1.68780972222222, max_speed = bV[k_max] * 1.68780972222222;          /* boundaries(k + 1), 1-based */
    if (tv->is_rotorcraft && max_speed > 304) max_speed = 304;                                            /* :107-112 */
    if (!tv->is_rotorcraft && min_speed < 30) min_speed = 30;
    /* :118-127 */
    tot = 0; cs = 0;
    for (int i = 0; i < rDH; i++) tot += dh_initial[i];
    k_min = 0; k_max = 0;
    for (int i = 0; i < rDH; i++) {
        cs += 100.0 * dh_initial[i] / tot;
        if (!k_min && cs >= 1.0) k_min = i + 1;
        if (!k_max && cs >= 99.0) k_max = i + 1;
    }
    double a = fabs(bDH[k_min] / 60.0), b = fabs(bDH[k_max] / 60.0);
    double max_vr = a > b ? a : b;
    if (isnan(max_vr) || !(tot > 0)) max_vr = 0;          /* :124-127 (all-zero slice: NaN in MATLAB) */
    out[0] = min_speed; out[1] = max_speed; out[2] = max_vr;
}

/* The point-mass dynamics stated in the header comment.  ic: v n e h psi theta phi a; ctrl: per whole second c,  */
/* [hdot_ft_s psidot_rad_s a_ft_ss] (the control row active during [c, c+1), events2controls.m:16-27); dyn[6].    */
/* out: (10 T + 1) rows [time north east up speed phi theta psi]; mm: up_min up_max speed_min speed_max max|vr|. */
void em_point_mass_dynamics(const double ic[8], const double *ctrl, int T, const double dyn[6], double *out, double mm[5]) {
    const double dt = 0.1, g = 32.2;
    double v = ic[0], n = ic[1], e = ic[2], h = ic[3], psi = ic[4], theta = ic[5], phi = ic[6];
    double up_min = h, up_max = h, v_min = v, v_max = v, vr_max = 0, h_sec = h;
    size_t row = 0;
    if (out) { double *o = out; o[0] = 0; o[1] = n; o[2] = e; o[3] = h; o[4] = v; o[5] = phi; o[6] = theta; o[7] = psi; }
    for (int c = 0; c < T; c++) {
        const double hdot = ctrl[3 * c], psidot = ctrl[3 * c + 1], acmd = ctrl[3 * c + 2];
        for (int s = 0; s < 10; s++) {
            double a = acmd;
            if ((v >= dyn[1] && a > 0) || (v <= dyn[0] && a < 0)) a = 0;
            double hd = hdot < dyn[2] ? dyn[2] : (hdot > dyn[3] ? dyn[3] : hdot);
            double sn = hd / v; sn = sn < -1 ? -1 : (sn > 1 ? 1 : sn);
            double q = (asin(sn) - theta) / dt; q = q < -dyn[4] ? -dyn[4] : (q > dyn[4] ? dyn[4] : q);
            theta = theta + q * dt;
            double r = (atan(v * psidot / g) - phi) / dt; r = r < -dyn[5] ? -dyn[5] : (r > dyn[5] ? dyn[5] : r);
            phi = phi + r * dt;
            const double ct = cos(theta), st = sin(theta);
            n = n + v * ct * cos(psi) * dt;
            e = e + v * ct * sin(psi) * dt;
            h = h + v * st * dt;
            psi = psi + g * tan(phi) / v * dt;
            v = v + a * dt; v = v < dyn[0] ? dyn[0] : (v > dyn[1] ? dyn[1] : v);
            row++;
            if (out) { double *o = out + row * 8; o[0] = (double)(10 * c + s + 1) / 10.0; o[1] = n; o[2] = e; o[3] = h; o[4] = v; o[5] = phi; o[6] = theta; o[7] = psi; }
            up_min = h < up_min ? h : up_min; up_max = h > up_max ? h : up_max;
            v_min = v < v_min ? v : v_min; v_max = v > v_max ? v : v_max;
        }
        const double vr = fabs(h - h_sec);   /* computeVerticalRate on the 1 Hz samples: (up(c+1) - up(c)) / 1 s */
        vr_max = vr > vr_max ? vr : vr_max;
        h_sec = h;
    }
    mm[0] = up_min; mm[1] = up_max; mm[2] = v_min; mm[3] = v_max; mm[4] = vr_max;
}

/* One trajectory of UncorEncounterModel.track (:419-471).  Philox mode: attempt j uses the key seed + j and the   */
/* trajectory's global index (the reference restarts a fresh stream per attempt with seed + 1, :424-428).         */
/* MT19937 mode: *seed_io is the running seed, advanced once per attempt like the reference's `seed = seed + 1`.  */
/* f32_inputs: round the sampled values to f32 first (what the GPU path's boundary hands its dynamics kernel).    */
int em_uncor_track_one(const em_model_t *m, int mode, uint64_t *seed_io, uint64_t gidx, int T, const em_uncor_opts_t *o,
                       const em_track_vars_t *tv, int max_track_attempts, int f32_inputs,
                       double *out /* (10T+1) x 8 or NULL */, double limits[3], int32_t *attempts_out, int32_t *sample_attempts_out,
                       double *margins /* mcap per-attempt decision margins, or NULL */, int mcap) {
    int ni = m->n_initial;
    int cap = (ni + m->n_dyn + 1) * T + 8; if (cap < 64) cap = 64;
    em_event_t *ev = (em_event_t *)malloc(sizeof(em_event_t) * (size_t)cap);
    double *dv = (double *)malloc(sizeof(double) * (size_t)ni * (size_t)(T + 1));
    int32_t *db = (int32_t *)malloc(sizeof(int32_t) * (size_t)ni * (size_t)(T + 1));
    double *ctrl = (double *)malloc(sizeof(double) * 3 * (size_t)T);
    int32_t ib[128]; double iv[128];
    int rc = -3;
    const int nbL = m->bnd_len[tv->idxL - 1], nbV = m->bnd_len[tv->idxV - 1], nbDH = m->bnd_len[tv->idxDH - 1];
    const double *bL = m->boundaries + m->bnd_off[tv->idxL - 1], *bV = m->boundaries + m->bnd_off[tv->idxV - 1], *bDH = m->boundaries + m->bnd_off[tv->idxDH - 1];
    double min_alt = 0, max_alt = INFINITY;                                                               /* :397-405 */
    if (nbL > 0) { min_alt = max_alt = bL[0]; for (int i = 1; i < nbL; i++) { min_alt = bL[i] < min_alt ? bL[i] : min_alt; max_alt = bL[i] > max_alt ? bL[i] : max_alt; } }
    double lo = bDH[0], hi = bDH[0], vmaxb = bV[0];
    for (int i = 1; i < nbDH; i++) { lo = bDH[i] < lo ? bDH[i] : lo; hi = bDH[i] > hi ? bDH[i] : hi; }
    for (int i = 1; i < nbV; i++) vmaxb = bV[i] > vmaxb ? bV[i] : vmaxb;
    const double dyn[6] = {1.7, vmaxb * 1.68780972222222, lo / 60.0, hi / 60.0, 3.0 * (3.14159265358979323846 / 180.0), 1000000.0}; /* :414 */
    for (int j = 0; j < max_track_attempts; j++) {
        em_rng_t g;
        em_rng_init(&g, mode, *seed_io);
        g.gidx = gidx;
        *seed_io += 1;                                                                                   /* :428 */
        int32_t att = 0;
        int k = em_uncor_sample_one(m, &g, T, o, ib, iv, ev, cap, &att);                                 /* :424 */
        if (k < 0) { rc = k; break; }
        if (sample_attempts_out) *sample_attempts_out = att;
        if (em_events2samples(ni, iv, ib, ev, k, dv, db, T) != T) { rc = -4; break; }
        if (f32_inputs) {
            for (int i = 0; i < ni; i++) iv[i] = (double)(float)iv[i];
            for (size_t q = 0; q < (size_t)ni * (size_t)T; q++) dv[q] = (double)(float)dv[q];
        }
        for (int c = 0; c < T; c++) {                                                                    /* :291-297 */
            ctrl[3 * c] = dv[(size_t)c * ni + tv->idxDH - 1] / 60.0;
            ctrl[3 * c + 1] = dv[(size_t)c * ni + tv->idxDPsi - 1] * (3.14159265358979323846 / 180.0);
            ctrl[3 * c + 2] = dv[(size_t)c * ni + tv->idxDV - 1] * 1.68780972222222;
        }
        const double h_ft = iv[tv->idxL - 1], v_ft_s = iv[tv->idxV - 1] * 1.68780972222222;              /* :434-438 */
        const double dv_ft_ss = iv[tv->idxDV - 1] * 1.68780972222222, dh_ft_s = iv[tv->idxDH - 1] / 60.0;
        const double dpsi = iv[tv->idxDPsi - 1] * (3.14159265358979323846 / 180.0);
        const double ic[8] = {v_ft_s, 0, 0, h_ft, 0, asin(dh_ft_s / v_ft_s), atan(v_ft_s * dpsi / 32.2), dv_ft_ss}; /* :441-446 */
        double mm[5];
        em_point_mass_dynamics(ic, ctrl, T, dyn, out, mm);
        if (margins && j < mcap) { margins[j] = 1e300; em_margin_ptr = &margins[j]; }
        em_uncor_dynamic_limits(m, tv, iv, mm[0], mm[1], mm[2], mm[3], limits);                           /* :459 (its discretize calls note their margins) */
        em_note(mm[0], min_alt); em_note(mm[1], max_alt); em_note(mm[2], limits[0]); em_note(mm[3], limits[1]);
        em_note_s(mm[4], limits[2], fabs(h_ft));   /* a difference of altitudes: rounding scales with the altitude */
        em_margin_ptr = NULL;
        const int viol_L = mm[0] < min_alt || mm[1] > max_alt;                                            /* :462-464 */
        const int viol_V = mm[2] < limits[0] || mm[3] > limits[1];
        const int viol_DH = mm[4] > limits[2];
        if (!(viol_L || viol_V || viol_DH)) { rc = 0; *attempts_out = j + 1; break; }
    }
    free(ev); free(dv); free(db); free(ctrl);
    return rc;
}

int64_t em_uncor_track_batch(const em_model_t *m, int mode, uint64_t seed, uint64_t first_index, int64_t n, int T,
                             const em_uncor_opts_t *o, const em_track_vars_t *tv, int max_track_attempts, int f32_inputs,
                             double *out /* n x (10T+1) x 8 or NULL */, double *limits /* n x 3 */, int32_t *attempts,
                             double *margins /* n x mcap or NULL */, int mcap) {
    uint64_t running = seed;   /* MT19937: one seed counter across samples AND attempts, like the reference's loop */
    for (int64_t i = 0; i < n; i++) {
        uint64_t s = mode == EM_RNG_MT19937 ? running : seed;
        int rc = em_uncor_track_one(m, mode, &s, first_index + (uint64_t)i, T, o, tv, max_track_attempts, f32_inputs,
                                    out ? out + (size_t)i * (size_t)(10 * T + 1) * 8 : NULL, limits + 3 * i, attempts + i, NULL,
                                    margins ? margins + (size_t)i * (size_t)mcap : NULL, mcap);
        if (rc != 0) { if (rc == -3) { attempts[i] = -1; continue; } return rc; }
        if (mode == EM_RNG_MT19937) running = s;
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* f2  CorTerminalModel.track  (@CorTerminalModel/track.m:45-150) and the static checks it calls                */
/*     (CorTerminalModel.m:117-333: getGeneratedMissDistance, CheckCumTurn, CheckRunwayProximity,               */
/*     CheckIntentVertical, CheckDynamicLimits).  em-core's computeVerticalRate / computeHeadingRate are not     */
/*     vendored ("dynamics unpinned"): taken as forward differences over the 1 s samples, the last value         */
/*     repeated, the heading difference wrapped to (-pi, pi].  computeAcceleration only feeds a counter (:67-76) */
/*     and has no effect on acceptance.  Reference defects kept or decided: `isClimb` (:122,134) is undefined in  */
/*     the reference (take-off intents error out there): read as is_climb; the turn-rate test compares rad/s with  */
/*     a deg/s limit (CorTerminalModel.m:296) and the runway distance uses 1.68781 as nm->ft (:215): both kept.   */
/* ------------------------------------------------------------------------- */
typedef struct {
    int32_t idx[12];            /* 1-based variable ids: own {distance bearing alt speed heading intent}, then int */
    double min_enc_time_s, thres_dist_ft, thres_alt_low_ft, thres_vertrate_ft_s;   /* track.m:14-17 */
    double max_cum_turn_deg[2], pitch_deg[2];                                       /* getDynamicLimits.m */
    int32_t local_smooth, pad;  /* createEncounter.m:88-89 through the stand-in below */
} em_ttrack_opts_t;

typedef struct { int n; double t[260], x[260], y[260], z[260], hdg[260], v[260]; } em_traj_t;

static double em_wrapTo180(double x) { return (x < -180 || 180 < x) ? em_wrapTo360(x + 180) - 180 : x; }
static double em_wrapToPi(double x) {
    const double pi = 3.14159265358979323846;
    if (x < -pi || pi < x) { const int pos = (x + pi) > 0; double y = (x + pi) - floor((x + pi) / (2 * pi)) * (2 * pi); if (y == 0 && pos) y = 2 * pi; return y - pi; }
    return x;
}
static double em_round1(double x) { return round(x * 10.0) / 10.0; }

/* CorTerminalModel.m:135-185.  heading: degrees, n values.  Returns is_reject. */
int em_check_cum_turn(const double *heading_in, int n, double limit) {
    if (n < 2 || !(limit < INFINITY)) return 0;   /* |cumsum| > inf never holds: nothing below decides anything */
    double hd[260];
    const int m = n - 1;
    for (int i = 0; i < m; i++) {
        const double a = em_wrapTo180(heading_in[i + 1]), b = em_wrapTo180(heading_in[i]);
        hd[i] = em_round1(a - b);
        if (em_margin_ptr) {   /* the rounding to one decimal: distance of 10 (a - b) to the nearest half-integer, at the headings' scale */
            const double x = (a - b) * 10.0;
            em_note_s(x, floor(x) + 0.5, 10.0 * fmax(fabs(a), fabs(b)));
        }
    }
    int ts[260], te[260], nts = 0, nte = 0;
    for (int i = 0; i + 1 < m; i++) {
        if (hd[i] == 0 && hd[i + 1] != 0) ts[nts++] = i + 2;   /* find(...) + 1, 1-based */
        if (hd[i] != 0 && hd[i + 1] == 0) te[nte++] = i + 1;
    }
    if (nts == 0) { ts[0] = 1; nts = 1; }
    if (nte == 0) { te[0] = m; nte = 1; }
    if (nts > nte) te[nte++] = m;
    for (int i = 0; i < nts; i++) {
        double h[260];
        int k = 0;
        for (int q = ts[i]; q <= te[i]; q++) h[k++] = em_wrapTo180(hd[q - 1]);
        /* segments of constant sign: startIdx = find([0; diff(sign(h))] ~= 0), with 1 and k+1 added when non-empty */
        int seg0 = 0;
        double cum = 0;
        int any_break = 0;
        for (int q = 1; q < k; q++) if (em_sign(h[q]) != em_sign(h[q - 1])) any_break = 1;
        for (int q = 0; q < k; q++) {
            if (any_break && q > seg0 && em_sign(h[q]) != em_sign(h[q - 1])) { seg0 = q; cum = 0; }
            cum += h[q];
            em_note(fabs(cum), limit);
            if (fabs(cum) > limit) return 1;
        }
    }
    return 0;
}

/* forward difference per 1 s sample, last repeated (stand-in for em-core's computeVerticalRate) */
static void em_forward_rate(const double *z, int n, double *out) {
    if (n == 1) { out[0] = 0; return; }
    for (int i = 0; i + 1 < n; i++) out[i] = z[i + 1] - z[i];
    out[n - 1] = out[n - 2];
}

/* CorTerminalModel.m:268-316 */
int em_check_dynamic_limits(const em_traj_t *tr, const em_dynlims_t *dl, double max_cum_turn_deg, double pitch_deg) {
    const int n = tr->n;
    if (n <= 1) return 0;
    double dh[260];
    em_forward_rate(tr->z, n, dh);
    for (int i = 0; i < n; i++) {
        const int is_alt = tr->z[i] > 0 && tr->z[i] <= dl->maxAltitude_ft;
        const int is_spd = tr->v[i] >= dl->minVel_ft_s && tr->v[i] <= dl->maxVel_ft_s;
        const int is_vr = fabs(dh[i]) <= dl->maxVertRate_ft_s;
        if (em_margin_ptr) {
            em_note_s(tr->z[i], 0.0, 1.0); em_note(tr->z[i], dl->maxAltitude_ft);
            em_note(tr->v[i], dl->minVel_ft_s); em_note(tr->v[i], dl->maxVel_ft_s);
            em_note_s(fabs(dh[i]), dl->maxVertRate_ft_s, fabs(tr->z[i]));
        }
        double rate;
        {   /* computeHeadingRate(deg2rad(heading), 1:n) */
            const int k = i + 1 < n ? i : n - 2;
            rate = em_wrapToPi(tr->hdg[k + 1] * (3.14159265358979323846 / 180.0) - tr->hdg[k] * (3.14159265358979323846 / 180.0));
        }
        const int is_turn = fabs(rate) <= dl->maxTurnRate_deg_s;
        em_note_s(fabs(rate), dl->maxTurnRate_deg_s, 6.3);   /* a difference of headings in radians */
        int is_pitch = 1;
        if (i > 0) {
            const double ratio = fabs(tr->z[i] - tr->z[i - 1]) / tr->v[i];
            if (em_margin_ptr) {
                const double zs = fabs(tr->z[i]) / fmax(tr->v[i], 1e-300);   /* the ratio's rounding scales with z / v */
                em_note_s(ratio, 1.0, zs);
                if (ratio <= 1) em_note_s(ratio, sin(pitch_deg * (3.14159265358979323846 / 180.0)), zs);
            }
            /* abs(asind(r)): for r > 1 MATLAB returns a complex number whose magnitude exceeds 90 */
            is_pitch = ratio <= 1 ? fabs(asin(ratio) * (180.0 / 3.14159265358979323846)) <= pitch_deg : (pitch_deg == INFINITY);
        }
        if (!(is_alt && is_spd && is_vr && is_turn && is_pitch)) return 0;
    }
    return !em_check_cum_turn(tr->hdg, n, max_cum_turn_deg);
}

/* One encounter through the filters of track.m:62-145.  tr[0] ownship, tr[1] intruder, time-sorted 1 s samples.  */
/* meta: tcpa_s hmd_ft vmd_ft enc_time_s.  Returns is_good.                                                       */
int em_terminal_filters(const em_traj_t *tr, int own_intent, int int_intent, const em_dynlims_t *dl, const em_ttrack_opts_t *o, double meta[4]) {
    /* getGeneratedMissDistance (CorTerminalModel.m:117-133): the times are consecutive integers */
    const double t0 = fmax(tr[0].t[0], tr[1].t[0]), t1 = fmin(tr[0].t[tr[0].n - 1], tr[1].t[tr[1].n - 1]);
    const int nc = (int)(t1 - t0) + 1;
    meta[0] = meta[1] = meta[2] = 0; meta[3] = nc > 0 ? nc : 0;
    if (nc <= 0) return 0;    /* no common time: min() of an empty set; the reference would error, here: rejected */
    const int ia = (int)(t0 - tr[0].t[0]), ib = (int)(t0 - tr[1].t[0]);
    double hmd = INFINITY; int best = 0;
    for (int k = 0; k < nc; k++) {
        const double dx = tr[0].x[ia + k] - tr[1].x[ib + k], dy = tr[0].y[ia + k] - tr[1].y[ib + k];
        const double d = sqrt(dx * dx + dy * dy) * 6076.1154855643;
        if (d < hmd) { hmd = d; best = k; }
    }
    if (em_margin_ptr)   /* the argmin: how close the runner-up came, at the scale of the positions the distances are differences of */
        for (int k = 0; k < nc; k++) {
            if (k == best) continue;
            const double dx = tr[0].x[ia + k] - tr[1].x[ib + k], dy = tr[0].y[ia + k] - tr[1].y[ib + k];
            const double ps = fmax(fmax(fabs(tr[0].x[ia + k]), fabs(tr[1].x[ib + k])), fmax(fabs(tr[0].y[ia + k]), fabs(tr[1].y[ib + k])));
            em_note_s(sqrt(dx * dx + dy * dy) * 6076.1154855643, hmd, ps * 6076.1154855643);
        }
    meta[0] = tr[0].t[ia + best]; meta[1] = hmd; meta[2] = tr[1].z[ib + best] - tr[0].z[ia + best];
    if (!(fabs(meta[0]) <= 10)) return 0;                                                    /* :84-87 */
    const int is_long = nc >= o->min_enc_time_s;                                             /* :90-91 */
    int close[2], low[2], climb[2], descend[2];
    for (int a = 0; a < 2; a++) {                                                            /* CorTerminalModel.m:203-226 */
        close[a] = low[a] = 0;
        for (int i = 0; i < tr[a].n; i++) {
            const double d_ft = hypot(tr[a].x[i], tr[a].y[i]) * 1.68781;
            em_note(d_ft, o->thres_dist_ft);
            if (d_ft <= o->thres_dist_ft) { close[a] = 1; em_note(tr[a].z[i], o->thres_alt_low_ft); if (tr[a].z[i] <= o->thres_alt_low_ft) low[a] = 1; }
        }
        double dh[260], zmax = tr[a].z[0], zmin = tr[a].z[0];                                /* CheckIntentVertical :228-266 */
        em_forward_rate(tr[a].z, tr[a].n, dh);
        for (int i = 1; i < tr[a].n; i++) { zmax = fmax(zmax, tr[a].z[i]); zmin = fmin(zmin, tr[a].z[i]); }
        const double thr_time = (zmax - zmin) / o->thres_vertrate_ft_s, pth = fmin(0.2, thr_time / (double)tr[a].n);
        int nclimb = 0, ndesc = 0;
        for (int i = 0; i < tr[a].n; i++) {
            nclimb += dh[i] >= o->thres_vertrate_ft_s; ndesc += dh[i] <= -o->thres_vertrate_ft_s;
            em_note_s(dh[i], o->thres_vertrate_ft_s, fabs(tr[a].z[i])); em_note_s(dh[i], -o->thres_vertrate_ft_s, fabs(tr[a].z[i]));
        }
        climb[a] = (double)nclimb / tr[a].n >= pth; descend[a] = (double)ndesc / tr[a].n >= pth;
        em_note((double)nclimb / tr[a].n, pth); em_note((double)ndesc / tr[a].n, pth);
        em_note(0.2, thr_time / (double)tr[a].n);
    }
    const int prox1 = (close[0] && low[0]) || !close[0];                                     /* :98-112 */
    const int prox2 = int_intent == 3 ? !(close[1] && low[1]) : ((close[1] && low[1]) || !close[1]);
    const int int_ok = int_intent == 1 ? descend[1] : (int_intent == 2 ? climb[1] : 1);      /* :119-127 */
    int own_ok = 0;                                                                          /* :130-137 (other intents: the variable stays unset in the reference) */
    if (own_intent == 1 || own_intent == 2) {
        const double c = own_intent == 1 ? 90.0 : 270.0;
        int ok = 0;
        for (int i = 0; i < tr[0].n; i++) { ok += tr[0].hdg[i] >= c - 30 && tr[0].hdg[i] <= c + 30; em_note(tr[0].hdg[i], c - 30); em_note(tr[0].hdg[i], c + 30); }
        own_ok = (own_intent == 1 ? descend[0] : climb[0]) && ((double)ok / tr[0].n >= .95);
    }
    const int dyn1 = em_check_dynamic_limits(&tr[0], &dl[0], o->max_cum_turn_deg[0], o->pitch_deg[0]);   /* :140-141 */
    const int dyn2 = em_check_dynamic_limits(&tr[1], &dl[1], o->max_cum_turn_deg[1], o->pitch_deg[1]);
    return is_long && prox1 && prox2 && own_ok && int_ok && dyn1 && dyn2;                    /* :144 */
}

/* test hook: the filters on two tracks given as rows [t_s x_nm y_nm z_ft heading_deg v_ft_s] (tests/test_oracle.py pins them against pyref.py) */
int em_terminal_filters_rows(const double *own, int n_own, const double *intr, int n_int, int own_intent, int int_intent,
                             const em_dynlims_t *dl, const em_ttrack_opts_t *o, double meta[4]) {
    em_traj_t tr[2];
    const double *src[2] = {own, intr};
    const int n[2] = {n_own, n_int};
    for (int a = 0; a < 2; a++) {
        if (n[a] < 1 || n[a] > 260) return -1;
        tr[a].n = n[a];
        for (int r = 0; r < n[a]; r++) {
            const double *q = src[a] + (size_t)r * 6;
            tr[a].t[r] = q[0]; tr[a].x[r] = q[1]; tr[a].y[r] = q[2]; tr[a].z[r] = q[3]; tr[a].hdg[r] = q[4]; tr[a].v[r] = q[5];
        }
    }
    return em_terminal_filters(tr, own_intent, int_intent, dl, o, meta);
}

/* createEncounter.m:88-89: traj.v_ft_s = local_smooth(t_s, v_ft_s, 5); traj.z_ft = local_smooth(t_s, z_ft, 15).  local_smooth lives in  */
/* em-core, which the reference does not vendor: UNPINNED.  The stand-in (the same in csrc/emgpu_kernels_tfilter.hip): a centred moving  */
/* average over w samples of the 1 s track, the window shrunk symmetrically at the ends -- row i becomes the mean of rows i-k .. i+k,    */
/* k = min((w-1)/2, i, n-1-i) (MATLAB smooth()'s rule for a moving average) --, summed in ascending row order in f64.                    */
void em_local_smooth(const double *x, int n, int w, double *out) {
    const int h = (w - 1) / 2;
    for (int i = 0; i < n; i++) {
        int k = h; if (i < k) k = i; if (n - 1 - i < k) k = n - 1 - i;
        double s = 0;
        for (int q = i - k; q <= i + k; q++) s += x[q];
        out[i] = s / (double)(2 * k + 1);
    }
}

/* forward + backward tracks of one aircraft, as the device stores them (f32), merged and ordered in time (createEncounter.m:74-84) */
static void em_merge_tracks(const double *fwd, int rf, const double *bck, int rb, int f32, em_traj_t *out) {
    int n = 0;
    for (int r = rb - 1; r >= 1; r--, n++) {
        const double *q = bck + (size_t)r * 6;
        out->t[n] = q[0]; out->x[n] = q[1]; out->y[n] = q[2]; out->z[n] = q[3]; out->hdg[n] = q[4]; out->v[n] = q[5];
    }
    for (int r = 0; r < rf; r++, n++) {
        const double *q = fwd + (size_t)r * 6;
        out->t[n] = q[0]; out->x[n] = q[1]; out->y[n] = q[2]; out->z[n] = q[3]; out->hdg[n] = q[4]; out->v[n] = q[5];
    }
    out->n = n;
    if (f32) for (int i = 0; i < n; i++) {
        out->t[i] = (float)out->t[i]; out->x[i] = (float)out->x[i]; out->y[i] = (float)out->y[i];
        out->z[i] = (float)out->z[i]; out->hdg[i] = (float)out->hdg[i]; out->v[i] = (float)out->v[i];
    }
}

/* CorTerminalModel.track for n encounters (Philox mode): attempt j of encounter i draws the geometry and propagates with */
/* the key seed + j and the global index first_index + i (the reference continues one MT19937 stream, track.m:36-58).    */
/* models[0] = geometry model, models[1..10] = the trajectory models in CorTerminalModel.m:84-100 order.                  */
/* sample [n][n_i], traj [n][2][cap2][6] (cap2 >= 2*(tmax+2)), len [n][2], meta [n][4], attempts [n] (-1: cap).           */
int64_t em_terminal_track_batch(const em_model_t *const *models, uint64_t seed, uint64_t first_index, int64_t n, const em_geom_opts_t *go,
                                const em_dynlims_t *dl, const em_ttrack_opts_t *o, double tmax_s, int max_resample, int max_track_attempts,
                                int f32, double *sample, double *traj, int32_t *len, double *meta, int32_t *attempts, int cap2,
                                double *margins /* n x mcap per-attempt decision margins, or NULL */, int mcap) {
    const em_model_t *gm = models[0];
    const int ni = gm->n_initial, cap = (int)tmax_s + 3;
    int64_t rc_all = 0;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t i = 0; i < n; i++) {
        double *out4 = (double *)malloc(sizeof(double) * 4 * (size_t)cap * 6);
        attempts[i] = -1;
        for (int j = 0; j < max_track_attempts; j++) {
            double v[128]; int32_t S[128], att;
            double *mg = (margins && j < mcap) ? &margins[(size_t)i * (size_t)mcap + j] : NULL;
            if (mg) *mg = 1e300;
            if (em_geom_sample_batch(gm, EM_RNG_PHILOX, seed + (uint64_t)j, first_index + (uint64_t)i, 1, go, S, v, &att) != 0) { rc_all = -3; break; }
            if (f32) for (int k = 0; k < ni; k++) v[k] = (float)v[k];
            double geo[12]; int32_t mo[4]; int32_t rows[4];
            for (int a = 0; a < 2; a++) {                                                   /* createEncounter.m:21-49 */
                const int32_t *ix = o->idx + 6 * a;
                double s, c;
                em_sincosd(v[ix[1] - 1], &s, &c);
                geo[6 * a] = v[ix[0] - 1] * c; geo[6 * a + 1] = v[ix[0] - 1] * s; geo[6 * a + 2] = v[ix[2] - 1];
                geo[6 * a + 3] = v[ix[3] - 1]; geo[6 * a + 4] = v[ix[4] - 1]; geo[6 * a + 5] = v[ix[5] - 1];
            }
            const int oi = (int)geo[5], ii_ = (int)geo[11];
            mo[0] = 2 * (oi - 1); mo[1] = mo[0] + 1; mo[2] = 4 + 2 * (ii_ - 1); mo[3] = mo[2] + 1;
            em_margin_ptr = mg;   /* the discretize calls of the propagation and every test of the filters note their margins */
            const int prc = em_propagate_batch(models + 1, mo, EM_RNG_PHILOX, seed + (uint64_t)j, first_index + (uint64_t)i, 1, geo, dl, tmax_s, max_resample, out4, rows, cap);
            if (prc != 0) { em_margin_ptr = NULL; continue; }
            em_traj_t tr[2];
            for (int a = 0; a < 2; a++) em_merge_tracks(out4 + (size_t)(2 * a) * cap * 6, rows[2 * a], out4 + (size_t)(2 * a + 1) * cap * 6, rows[2 * a + 1], f32, &tr[a]);
            if (o->local_smooth)
                for (int a = 0; a < 2; a++) {
                    double tmp[260];
                    em_local_smooth(tr[a].v, tr[a].n, 5, tmp);
                    for (int r = 0; r < tr[a].n; r++) tr[a].v[r] = f32 ? (double)(float)tmp[r] : tmp[r];
                    em_local_smooth(tr[a].z, tr[a].n, 15, tmp);
                    for (int r = 0; r < tr[a].n; r++) tr[a].z[r] = f32 ? (double)(float)tmp[r] : tmp[r];
                }
            double mt[4];
            const int good = em_terminal_filters(tr, oi, ii_, dl, o, mt);
            em_margin_ptr = NULL;
            if (!good) continue;
            attempts[i] = j + 1;
            for (int k = 0; k < ni; k++) sample[i * ni + k] = v[k];
            for (int a = 0; a < 2; a++) {
                len[2 * i + a] = tr[a].n;
                for (int r = 0; r < tr[a].n && r < cap2; r++) {
                    double *q = traj + (((size_t)i * 2 + a) * cap2 + r) * 6;
                    q[0] = tr[a].t[r]; q[1] = tr[a].x[r]; q[2] = tr[a].y[r]; q[3] = tr[a].z[r]; q[4] = tr[a].hdg[r]; q[5] = tr[a].v[r];
                }
            }
            for (int k = 0; k < 4; k++) meta[4 * i + k] = mt[k];
            break;
        }
        free(out4);
    }
    return rc_all;
}
