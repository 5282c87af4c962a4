/*
 * emgpu.h -- C ABI of libemgpu.so: the MI355X-native replacement of the sampling hot path of
 * Airspace-Encounter-Models/em-model-manned-bayes.
 *
 * The reference has no FFI boundary of its own (plain MATLAB functions and handle classes); this
 * header is the boundary a maintainer binds instead.  Every entry point cites the reference
 * interface it replaces (paths relative to the reference's code/matlab/).  The MATLAB-side mex
 * binding and the Python ctypes binding are shown in INTEGRATION.md.
 *
 * Conventions
 *   - C linkage, plain pointers and sizes, no C++/torch types.
 *   - Every function returns an int status (EMGPU_OK or a negative EMGPU_ERR_*); the text of the
 *     last error of the calling thread is returned by emgpu_last_error().
 *   - The caller owns every output buffer.  The library owns only emgpu_model / emgpu_ctx handles.
 *   - Variable ids, bins and the temporal map are 1-based at this boundary, like the reference.
 *   - Matrices that mirror MATLAB arrays are column-major (N{i} is r_i x q_i, em_read.m:191-198).
 *   - Re-entrant per ctx; a ctx is bound to one device and launches on one HIP stream.  Calls on
 *     the same ctx from several threads are serialised by a lock inside the ctx; models are only
 *     read by the sampling calls (setters must not race with sampling, as in any MATLAB handle class).
 */
#ifndef EMGPU_H
#define EMGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMGPU_OK 0
#define EMGPU_ERR_ARG (-1)          /* bad argument / shape                                        */
#define EMGPU_ERR_IO (-2)           /* file could not be opened                                    */
#define EMGPU_ERR_PARSE (-3)        /* 'Unknown field' em_read.m:104, malformed sections           */
#define EMGPU_ERR_PRESET (-4)       /* 'Attempt to preset a dependent variable' bn_sample.m:47     */
#define EMGPU_ERR_HIP (-5)          /* HIP runtime error                                           */
#define EMGPU_ERR_REJECT_CAP (-6)   /* rejection loop hit max_attempts (reference loops forever)   */
#define EMGPU_ERR_EVENT_CAP (-7)    /* an event list did not fit event_cap rows                    */
#define EMGPU_ERR_NO_DEVICE (-8)    /* no HIP device: the product path has no CPU fallback         */
#define EMGPU_ERR_UNSUPPORTED (-9)  /* model shape beyond the compiled maxima                      */
#define EMGPU_ERR_PRIOR (-10)       /* 'prior:notdbe' / 'prior:unknown' bn_dirichlet_prior.m:28,37 */
#define EMGPU_ERR_SORT (-11)        /* 'Network could not be hierarchically sorted' bn_sort.m:23   */

typedef struct emgpu_model emgpu_model; /* parsed model + priors + start (EncounterModel.m:5-70)   */
typedef struct emgpu_ctx emgpu_ctx;     /* one device + one stream + uploaded tables               */

const char *emgpu_last_error(void);
const char *emgpu_version(void);   /* "emgpu <version> (gfx950) philox4x32-<rounds> src:<hash of the sources it was built from>" */
/* Rounds of the Philox4x32 generator this build draws with (7; builds before round 3 and -DEMGPU_PHILOX_ROUNDS=10 builds: 10).  The
 * round count is part of the sampler's identity: the same (seed, global index) gives other samples under another count, so data and
 * goldens of one count are not comparable with, or resumable by, a build of the other -- check it where that matters. */
int32_t emgpu_philox_rounds(void);
/* Revision of the RNG slot map (which Philox counter word every draw of the reference's algorithm takes).  Like the
 * round count it is part of the sampler's identity: a change gives other samples for the same (seed, global index).
 *   1  rounds 1-4 of this library
 *   2  round 5 on ("emgpu 0.4" is the first version string to say so): terminal propagation takes an attempt's FIRST dediscretize draw from word 3 of its TERM_TRANS block (createEncounter.m:203,
 *      208,216); every other section unchanged -- uncor / cor samples are identical under 1 and 2, terminal tracks are not. */
int32_t emgpu_slot_map_revision(void);

/* ------------------------------------------------------------------------------------------------
 * Model: replaces em_read.m:1-206 and the data half of @EncounterModel/EncounterModel.m
 * ---------------------------------------------------------------------------------------------- */

/* em_read(parameters_filename, 'idxZeroBoundaries', idx, 'isOverwriteZeroBoundaries', flag)
 * (em_read.m:1,41-42).  idx_zero_boundaries may be NULL (=> [1 2 3], em_read.m:42). */
int emgpu_model_load_txt(const char *path, const int32_t *idx_zero_boundaries, int32_t n_idx,
                         int32_t is_overwrite_zero_boundaries, emgpu_model **out);

/* Binary model cache: em_read parses 0.5-3 MB of text and the plan compiler then searches a quantile threshold for
 * every count -- emgpu_model_save_bin writes the parsed model (every field of em_read.m:47-107, the priors, `start`) AND its compiled plan
 * to one file; emgpu_model_load_bin reads it back (no parse, no search).  The file carries the hash of the library's sources
 * (emgpu_version()): a file written by other sources is refused with EMGPU_ERR_PARSE and the caller reads the .txt again.  A loaded model
 * is an ordinary model: setters invalidate its plan like anybody's. */
int emgpu_model_save_bin(const emgpu_model *m, const char *path);
int emgpu_model_load_bin(const char *path, emgpu_model **out);

/* Build a model from caller arrays (the MATLAB struct contract of dbn_sample.m:25-33), so that
 * MATLAB-side edits of N_initial / N_transition / boundaries propagate.
 *   G_*: n x n row-major uint8, [parent][child] (em_read.m:204, bn_sample.m:42).
 *   N_initial: concatenation over nodes 1..n_initial of r_i x q_i column-major counts.
 *   N_transition: concatenation over nodes n_initial+1..n_transition (em_read.m:92).
 *   temporal_map: n_dyn x 2 row-major (var at t, var at t+1), may be NULL when n_transition==0.
 *   boundaries: concatenated, bnd_len[i]==0 means '*' (em_read.m:97-99).  zero_bins (0 = none)
 *   may be NULL => derived like extract_zero_bins (em_read.m:143-156).
 *   labels_* : '\n'-separated, may be NULL. */
typedef struct {
    int32_t n_initial, n_transition, n_dyn, _pad;
    const uint8_t *G_initial, *G_transition;
    const int32_t *r_initial, *r_transition;
    const int32_t *temporal_map;
    const double *N_initial;
    int64_t n_N_initial;
    const double *N_transition;
    int64_t n_N_transition;
    const double *boundaries;
    const int32_t *bnd_len;
    const int32_t *zero_bins;
    const double *resample_rates;
    const char *labels_initial, *labels_transition;
} emgpu_model_desc;
int emgpu_model_from_arrays(const emgpu_model_desc *d, emgpu_model **out);
void emgpu_model_free(emgpu_model *m);

typedef struct {
    int32_t n_initial, n_transition, n_dyn;
    int32_t is_dynvar_depend;  /* any(G_transition(dyn,dyn),'all')  dbn_sample.m:55 */
    int64_t n_N_initial, n_N_transition;
    int32_t max_r, n_resample_active;
} emgpu_model_info_t;
int emgpu_model_info(const emgpu_model *m, emgpu_model_info_t *out);

/* Field access.  get_* copy into out (cap elements) and return the element count (>=0) or an
 * error (<0); passing out==NULL returns the count only. */
enum {
    EMGPU_F_R_INITIAL = 1, EMGPU_F_R_TRANSITION, EMGPU_F_ORDER_INITIAL, EMGPU_F_ORDER_TRANSITION,
    EMGPU_F_TEMPORAL_MAP,   /* n_dyn x 2 row-major                                         */
    EMGPU_F_ZERO_BINS, EMGPU_F_START, EMGPU_F_G_INITIAL, EMGPU_F_G_TRANSITION, /* n*n row-major */
    EMGPU_F_N_INITIAL = 32, EMGPU_F_N_TRANSITION, EMGPU_F_ALPHA_INITIAL, EMGPU_F_ALPHA_TRANSITION,
                            /* node = 1-based variable id; r x q column-major                */
    EMGPU_F_BOUNDARIES,     /* node = 1-based initial variable                               */
    EMGPU_F_RESAMPLE_RATES,
    EMGPU_F_LABELS_INITIAL = 64, EMGPU_F_LABELS_TRANSITION /* '\n'-separated, via get_text   */
};
int64_t emgpu_model_get_i32(const emgpu_model *m, int32_t field, int32_t *out, int64_t cap);
int64_t emgpu_model_get_f64(const emgpu_model *m, int32_t field, int32_t node, double *out, int64_t cap);
int64_t emgpu_model_get_text(const emgpu_model *m, int32_t field, char *out, int64_t cap);
int emgpu_model_set_f64(emgpu_model *m, int32_t field, int32_t node, const double *v, int64_t n);

/* EncounterModel.prior / set.prior (EncounterModel.m:45,194-203) -> bn_dirichlet_prior.m:18-37.
 * kind: 0 = numeric constant `value`; 1 = 'dbe' (1/(r*q)).  Rebuilds both alpha sets. */
int emgpu_model_set_prior(emgpu_model *m, int32_t kind, double value);
/* setTransitionPriors.m:12-33: alpha_transition{ii}(kk, n(kk-1)+1:n*kk) = prior for every dynamic
 * variable that has parents (used by createEncounter.m:129). */
int emgpu_model_set_transition_stay_prior(emgpu_model *m, double prior);
/* EncounterModel.start (EncounterModel.m:52,205-207): n_initial entries, 0 = unset ([] or NaN). */
int emgpu_model_set_start(emgpu_model *m, const int32_t *start, int32_t n);
/* Importance-sampling hook next to `start` and `layers` (UncorEncounterModel.m:204,259-272; InitStartTerminal.m:1-92):
 * log of the model probability of the preset values, sum over preset nodes of log P(x_i = start_i | parents) with
 * P = (N + alpha) column-normalised.  Preset nodes have only preset parents (bn_sample.m:45-47), so this is ONE number per
 * call -- the log-weight of every sample drawn with this `start` (0 when nothing is preset, -inf for an impossible preset).
 * A start GRID (one row of presets per sample) with per-sample weights: emgpu_sample_params.start / emgpu_sample_out.log_weight and
 * emgpu_bn_params.start / .log_weight. */
int emgpu_model_start_log_weight(const emgpu_model *m, double *out);
/* EncounterModel.zero_bins (EncounterModel.m:40; derived once by em_read.m:110-114,143-156 and, like in
 * the reference, NOT re-derived when boundaries are replaced): n_initial entries, 0 = none. */
int emgpu_model_set_zero_bins(emgpu_model *m, const int32_t *zero_bins, int32_t n);

/* ------------------------------------------------------------------------------------------------
 * Context
 * ---------------------------------------------------------------------------------------------- */
/* A ctx owns: a stream, the uploaded tables of the models it has sampled (an LRU cache of ~48) and the device scratch of the
 * host-pointer and .track entry points (kept between calls and grown on demand -- a fresh hipMalloc / hipFree of gigabytes per call
 * costs up to 100 ms; emgpu_ctx_free releases everything). */
int emgpu_ctx_create(int32_t device, emgpu_ctx **out);
/* Launch on a caller stream (a hipStream_t passed as void*; NULL = the HIP default stream).
 * A new ctx launches on its own non-blocking stream until this is called. */
int emgpu_ctx_set_stream(emgpu_ctx *ctx, void *hip_stream);
/* Wait for the ctx stream and report deferred per-trajectory errors of *_device calls
 * (EMGPU_ERR_REJECT_CAP / EMGPU_ERR_EVENT_CAP). */
int emgpu_ctx_sync(emgpu_ctx *ctx);
/* Give the device scratch of the host-pointer and .track entry points back (it is kept between calls and only ever grows: one
 * 1 M x 240 s dense host call leaves 3-4 GB with the ctx); the uploaded model tables stay.  Synchronises the ctx stream. */
int emgpu_ctx_trim(emgpu_ctx *ctx);
void emgpu_ctx_free(emgpu_ctx *ctx);

/* ------------------------------------------------------------------------------------------------
 * Sampling: replaces UncorEncounterModel.sample (UncorEncounterModel.m:192-313) and what it calls:
 * dbn_hierarchical_sample.m:1, dbn_sample.m:1, bn_sample.m:1, select_random.m:1, asub2ind.m:1,
 * resample_events.m:1, dediscretize.m:1.
 * ---------------------------------------------------------------------------------------------- */
enum { EMGPU_TRANSITION_REFERENCE_AUTO = 0, EMGPU_TRANSITION_PER_STEP = 1 };
#define EMGPU_FLAG_QUANTIZE500 1u /* 'isQuantize500' UncorEncounterModel.m:202,266-268            */
#define EMGPU_FLAG_NO_RESAMPLE 2u /* skip resample_events (plain dbn_sample.m semantics)           */
#define EMGPU_FLAG_NO_DEDISC 4u   /* skip dediscretize: values are the bin indices                 */
#define EMGPU_FLAG_NO_TERMINATOR 8u /* event lists end without the [T-sum(dt) 0 0] row            */
#define EMGPU_FLAG_LOCAL_SMOOTH 16u /* terminal tracks: smooth speed (5 s) and altitude (15 s) like createEncounter.m:88-89.  local_smooth is
                                       em-core's (not vendored: UNPINNED); the stand-in is a centred moving average whose window shrinks
                                       symmetrically at a track's ends (row i = mean of rows i-k..i+k, k = min((w-1)/2, i, n-1-i)) */

typedef struct {
    uint64_t seed;        /* Philox key.  'seed' of .sample (UncorEncounterModel.m:201)            */
    uint64_t first_index; /* global index of trajectory 0 of this call (multi-GPU sharding)        */
    int64_t n;            /* n_samples                                                             */
    int32_t sample_time;  /* T, seconds (>=1)                                                      */
    int32_t transition_mode;
    uint32_t flags;
    int32_t max_attempts; /* rejection cap (reference: unbounded while, UncorEncounterModel.m:248) */
    int32_t idx_L, idx_v, idx_dh; /* 1-based ids of "L","v","\dot h" (UncorEncounterModel.m:225-229);
                                     idx_v==0 || idx_dh==0 disables the rejection test (:275)      */
    int32_t n_layers;     /* rows of `layers` (r_L) or 0                                           */
    const double *layers; /* n_layers x 2 row-major [lo hi] (UncorEncounterModel.m:204,259-260)    */
    int32_t event_cap;    /* rows per trajectory in `events`                                       */
    int32_t _pad;
    const uint64_t *indices; /* optional: n global indices replacing first_index + i (a device pointer for
                                *_device calls, a host pointer for *_host): arbitrary subsets of a batch, e.g.
                                the trajectories .track re-draws; NULL = the contiguous range               */
    const int32_t *start;    /* optional: a start GRID -- n rows of n_initial preset bins by variable id (row-major; 0 = unset: the
                                model's own `start` then applies), one row per trajectory: what a loop over EncounterModel.start values
                                (UncorEncounterModel.m:204, bn_sample.m:44-50) does in as many calls, in ONE.  A device pointer for
                                *_device calls, a host pointer for *_host.  A row that presets a node without presetting its parents,
                                or a bin outside 1..r, is reported as EMGPU_ERR_PRESET (by emgpu_ctx_sync for *_device calls).
                                Calls with a start grid or log-weights run on the general kernel.                                   */
} emgpu_sample_params;

/* Event row (8 bytes): what one row [dt var value] of out_events{i} carries. */
typedef struct {
    uint16_t dt;  /* seconds since the previous row                                               */
    uint8_t var;  /* 1-based initial-network variable, 0 = terminator row                         */
    uint8_t bin;  /* discrete value                                                               */
    float value;  /* dediscretised value                                                          */
} emgpu_event;

/* Device-native outputs (time-blocked SoA; any pointer may be NULL to skip that output).
 * G4 = ceil(T/4).  Column c of trajectory i (c = 0 is the initial state, events2samples.m:15-26):
 *   dyn_bin[((c/4)*n_dyn + k)*n + i]        byte (c%4) of the uint32 = bin (1-based)
 *   dyn_val[(((c/4)*n_dyn + k)*n + i)*4 + c%4]
 * with k = row of the temporal map (ascending variable id).  Padding columns >= T are 0.
 *
 * Writing a shard into a larger shared trace: `ld` is the trajectory dimension of the buffers (0 = the
 * call's own n) and `col_offset` the column of trajectory 0 of this call, i.e. trajectory i lands in
 * column col_offset + i of arrays dimensioned [..][ld] (events: list col_offset + i of [ld][event_cap]).
 * That is how the ranks / devices / model blocks of one batch fill one trace without temporaries
 * (SURVEY.md 8e "sort/block by model id"; RUN_1_emsample.m:24-47 is the reference's only sharding). */
typedef struct {
    uint8_t *init_bin;   /* [n_initial][n]                                                        */
    float *init_val;     /* [n_initial][n]                                                        */
    uint32_t *dyn_bin;   /* [G4][n_dyn][n]                                                        */
    float *dyn_val;      /* [G4][n_dyn][n][4]                                                     */
    uint32_t *ev_count;  /* [n] rows written (may exceed event_cap => EMGPU_ERR_EVENT_CAP)        */
    emgpu_event *events; /* [n][event_cap]                                                        */
    int32_t *attempts;   /* [n] attempts used by the rejection loop; <0 => cap hit                */
    int64_t ld;          /* trajectory dimension of every buffer above; 0 => params.n             */
    int64_t col_offset;  /* column of trajectory 0 of this call; col_offset + n <= ld             */
    double *log_weight;  /* [n] (NOT offset by col_offset) per-trajectory importance weight of its presets: the sum over the trajectory's
                            preset nodes of log P(preset | parents), P = (N + alpha) column-normalised -- emgpu_model_start_log_weight
                            per row of the start grid; NULL: not wanted                            */
} emgpu_sample_out;

/* Asynchronous: enqueue on the ctx stream with DEVICE pointers in `out`; returns after launch.
 * Deferred errors are reported by emgpu_ctx_sync. */
int emgpu_sample_dbn_device(emgpu_ctx *ctx, const emgpu_model *m, const emgpu_sample_params *p,
                            const emgpu_sample_out *out);
/* Synchronous, for the MATLAB / Python class layer: HOST pointers in `out` (UncorEncounterModel.m:283-300 hands host arrays back).
 * A pipeline (round 6): the batch is cut into chunks; chunk k's kernel runs on the ctx stream while chunk k-1 crosses PCIe on a copy
 * stream into pinned memory and chunk k-2 is copied into the caller's arrays by a few host threads (pageable arrays), or the copy engine
 * writes straight into the caller's arrays (arrays from emgpu_host_alloc / hipHostMalloc / hipHostRegister).  Event lists are packed on
 * the device first (a prefix sum over ev_count): sum(ev_count) rows cross PCIe, not n x event_cap.  PCIe-inclusive: reported by bench.py
 * as `host_path`, never as the headline.  Of `events` only rows [0, min(ev_count[i], event_cap)) of list i are defined on return. */
int emgpu_sample_dbn_host(emgpu_ctx *ctx, const emgpu_model *m, const emgpu_sample_params *p,
                          const emgpu_sample_out *out);

/* ------------------------------------------------------------------------------------------------
 * Trace placement (round 6).  WHERE a 36 GB trace lies in device memory decides how fast the sampler writes it: the same launch takes
 * 5.9, 6.6 or 7.0 ms depending on the allocation, launch after launch (profiles/r05_placement_probe.txt, profiles/r06_placement_probe.txt).
 * A consumer of emgpu_sample_dbn_device therefore asks the LIBRARY for its trace instead of calling hipMalloc.  Two things happen there:
 *   (i)  blocks of 1 GiB and more are ONE address range backed by separately created 1 GiB physical chunks (the HIP virtual-memory calls):
 *        four to six of six such blocks are of the fast kind where hipMalloc's are mostly of the middle one (why is not known: measured);
 *   (ii) emgpu_trace_alloc allocates `candidates` blocks, times the caller's own call (m, p) on each -- 0.5 s of launches to load the device,
 *        then two rounds of 2 untimed + 5 timed launches per candidate, the better round counts -- keeps the fastest and frees the others.
 *        Candidate 0 is a plain hipMalloc block: report.first_allocation_ms is what the caller's own allocation would have got.
 * What the loop over samples of UncorEncounterModel.m:244 writes into is, here, one such trace.
 *   want        EMGPU_TRACE_* : which outputs the trace holds (events: [ld][p->event_cap] rows)
 *   candidates  0 = automatic: one block (nothing timed) below 1 GiB, where the launch is too short for placement to matter; else 6, or as
 *               many as the device's free memory holds.  1 = one block of kind (i), nothing timed.  n > 1: that many.
 * The trace's trajectory dimension ld is p->n rounded up to 1 024 columns (every row of every array starts on a 1 KiB boundary).
 * emgpu_trace_free gives the block back to the ctx's POOL: the next emgpu_trace_alloc it fits (and is not more than 25 % too large for) takes
 * it without a new probe (report.reused = 1).  emgpu_ctx_trim / emgpu_ctx_free release the pool (the memory goes back to the device; the
 * ADDRESS RANGE of a chunked block is never re-used -- a HIP runtime crashes when a new range overlaps a released one that had been the source
 * of copies; address space is the only cost).  A trace must be freed before its ctx.  The probe launches overwrite the trace with the samples
 * of (p->seed, p->first_index ...): the same samples the caller's own call will write.
 * ---------------------------------------------------------------------------------------------- */
#define EMGPU_TRACE_INIT 1u     /* init_bin + init_val                  */
#define EMGPU_TRACE_DENSE 2u    /* dyn_bin + dyn_val                    */
#define EMGPU_TRACE_EVENTS 4u   /* ev_count + events[ld][p->event_cap]  */
#define EMGPU_TRACE_ATTEMPTS 8u /* attempts                             */
typedef struct emgpu_trace emgpu_trace;
typedef struct {
    int64_t bytes;             /* size of one candidate = one device allocation                                         */
    int64_t ld;                /* trajectory dimension of every array of the trace                                      */
    int32_t candidates;        /* blocks allocated and timed (1: nothing was timed)                                     */
    int32_t kept;              /* index of the kept one, in allocation order                                            */
    int32_t reused;            /* 1: a block of the ctx's pool, placed by an earlier call; nothing was timed now        */
    int32_t _pad;
    float ms[8];               /* per candidate: ms per launch, the better of its two rounds                            */
    float first_allocation_ms; /* = ms[0]: what a caller who keeps the first allocation gets (0: nothing was timed)     */
    float kept_ms;
} emgpu_trace_report_t;
int emgpu_trace_alloc(emgpu_ctx *ctx, const emgpu_model *m, const emgpu_sample_params *p, uint32_t want, int32_t candidates,
                      emgpu_trace **out);
/* The trace as the `out` argument of emgpu_sample_dbn_device / _blocks_device: device pointers, ld set, col_offset 0. */
int emgpu_trace_out(const emgpu_trace *t, emgpu_sample_out *out);
int emgpu_trace_report(const emgpu_trace *t, emgpu_trace_report_t *out);
int emgpu_trace_free(emgpu_ctx *ctx, emgpu_trace *t);

/* Device memory from the allocator the traces come from (nothing is timed): for outputs that are not a DBN trace -- the joined tracks of
 * emgpu_sample_terminal_device (createEncounter.m:74-84), a consumer's own buffers.  Freed by emgpu_device_free, or with the ctx. */
int emgpu_device_alloc(emgpu_ctx *ctx, uint64_t bytes, void **out);
int emgpu_device_free(emgpu_ctx *ctx, void *p);

/* Pinned host memory for the outputs of the *_host entry points (hipHostMalloc, kept in a per-ctx pool: pinning gigabytes costs about as
 * much as copying them).  emgpu_sample_dbn_host recognises pinned output arrays and lets the copy engine write straight into them;
 * pageable arrays go through the library's own pinned staging buffers and a few host threads (below). */
int emgpu_host_alloc(emgpu_ctx *ctx, uint64_t bytes, void **out);
int emgpu_host_free(emgpu_ctx *ctx, void *p);   /* back to the pool; emgpu_ctx_trim releases the pool's free blocks */

/* Phases of the last emgpu_sample_dbn_host call on this ctx (the call is a pipeline: chunk k's kernel runs while chunk k-1 crosses PCIe
 * and chunk k-2 is copied from staging into the caller's arrays, so the phases overlap and do not add up to total_ms). */
typedef struct {
    double total_ms;        /* wall time of the call                                                                    */
    double kernel_ms;       /* sum of the chunks' launch durations (HIP events on the launch stream)                    */
    double d2h_ms;          /* sum of the chunks' copy durations (HIP events on the copy stream)                        */
    double scatter_ms;      /* host wall time spent copying staging -> caller arrays (0 for pinned outputs)             */
    int64_t bytes_d2h;      /* bytes that crossed PCIe                                                                  */
    int64_t event_rows;     /* rows of the event lists copied (the lists are packed on the device first)               */
    int32_t chunks, chunk_n;
    int32_t threads;        /* host threads of the scatter                                                              */
    int32_t direct;         /* 1: every large output was pinned memory (no staging)                                    */
} emgpu_host_stats_t;
int emgpu_host_stats(const emgpu_ctx *ctx, emgpu_host_stats_t *out);

/* ------------------------------------------------------------------------------------------------
 * One batch, several models / several devices.  The reference shards only by running em_sample in
 * a parfor over model files (RUN_1_emsample.m:13,24-47) and loops over samples serially
 * (UncorEncounterModel.m:244); trajectories are independent and every RNG slot is keyed by the GLOBAL
 * sample index, so any split gives the same trajectories.
 * ---------------------------------------------------------------------------------------------- */

/* The contiguous block [*lo, *hi) of `rank` when n_total indices are split over `world` parts (the
 * first n_total % world parts get one extra): the split every sharded entry point below uses. */
int emgpu_shard_range(int64_t n_total, int32_t rank, int32_t world, int64_t *lo, int64_t *hi);
int emgpu_device_count(int32_t *count);

/* Mixed-model batch (BASELINE.json configs[3]): block b samples trajectories
 * [first_index, first_index + n) (global indices) from models[model] into columns
 * out->col_offset + (first_index - p->first_index) + [0, n) of ONE shared trace whose trajectory
 * dimension is out->ld (0 => p->n).  p->n / p->first_index describe the range the trace covers;
 * every block must lie inside it.  All models must agree in n_initial and n_dyn (the trace shape).
 * One launch per block on the ctx stream; device pointers; asynchronous. */
typedef struct {
    int32_t model, _pad;
    uint64_t first_index;
    int64_t n;
} emgpu_block;
int emgpu_sample_dbn_blocks_device(emgpu_ctx *ctx, const emgpu_model *const *models, int32_t n_models,
                                   const emgpu_sample_params *p, const emgpu_block *blocks, int32_t n_blocks,
                                   const emgpu_sample_out *out);
/* The blocks of the equal-contiguous-block assignment "model m owns emgpu_shard_range(n_total, m,
 * n_models)" that intersect [lo, hi) (one rank's share): writes at most n_models entries, returns the count. */
int32_t emgpu_mixed_blocks(int64_t n_total, int32_t n_models, int64_t lo, int64_t hi, emgpu_block *blocks);

/* One call, several devices (one host thread + one HIP stream per device inside a
 * call): [p->first_index, p->first_index + p->n) is split with emgpu_shard_range over the n_ctx
 * contexts (each bound to its own device -- or to the same one, which only adds streams).
 *   _multi_host:   HOST pointers in `out` dimensioned for all p->n trajectories; synchronous; deferred
 *                  errors (rejection / event cap) are returned.  What a single-threaded MATLAB / Python
 *                  caller uses to drive 8 GPUs.
 *   _multi_device: outs[d] holds DEVICE pointers on ctx d's device for ITS shard only (ld / col_offset
 *                  as above, relative to the shard); asynchronous: emgpu_ctx_sync each ctx afterwards. */
int emgpu_sample_dbn_multi_host(emgpu_ctx *const *ctxs, int32_t n_ctx, const emgpu_model *m,
                                const emgpu_sample_params *p, const emgpu_sample_out *out);
int emgpu_sample_dbn_multi_device(emgpu_ctx *const *ctxs, int32_t n_ctx, const emgpu_model *m,
                                  const emgpu_sample_params *p, const emgpu_sample_out *outs);

/* bn_sample(G,r,N,alpha,num_samples,start,order) (bn_sample.m:1) on the initial network with an
 * optional dediscretize + rejection stage = the geometry draw of @CorTerminalModel/sample.m:29-77.
 * out_bin [n_initial][n] uint8, out_val [n_initial][n] float (device or host per the suffix). */
typedef struct {
    uint64_t seed, first_index;
    int64_t n;
    uint32_t flags;          /* EMGPU_FLAG_NO_DEDISC => plain bn_sample                           */
    int32_t max_attempts;
    const double *bounds_sample; /* n_initial x 2 row-major or NULL (sample.m:45-53)              */
    int32_t idx_own_speed, idx_int_speed; /* 1-based; 0 = no speed test (sample.m:64-70)          */
    double min_vel1, max_vel1, min_vel2, max_vel2;
    const int32_t *start;    /* optional start grid [n][n_initial] and per-sample log-weights [n]: as in emgpu_sample_params /
                                emgpu_sample_out -- InitStartTerminal.m:57-90 builds such a grid and RUN_terminal.m:33-50 loops over its 18
                                rows with one .sample call each; here the grid is ONE launch (device pointers for _device, host for _host) */
    double *log_weight;
} emgpu_bn_params;
int emgpu_sample_bn_device(emgpu_ctx *ctx, const emgpu_model *m, const emgpu_bn_params *p,
                           uint8_t *out_bin, float *out_val, int32_t *attempts);
int emgpu_sample_bn_host(emgpu_ctx *ctx, const emgpu_model *m, const emgpu_bn_params *p,
                         uint8_t *out_bin, float *out_val, int32_t *attempts);

/* ------------------------------------------------------------------------------------------------
 * Terminal trajectory propagation: replaces PropagateTrajectory (@CorTerminalModel/createEncounter.m:
 * 93-265) for both aircraft and both directions of n encounters (createEncounter.m:52-72); em-core's
 * local_smooth (:88-89) only as the stand-in of EMGPU_FLAG_LOCAL_SMOOTH.  models[]: trajectory models with the 6 initial variables
 * {intent, distance, bearing, heading, altitude, speed} and 3 dynamic ones, all of the same shapes and
 * boundaries; the caller has applied setTransitionPriors(...,1) (createEncounter.m:129) through
 * emgpu_model_set_transition_stay_prior.
 *   geo      [n][12] f64: x0_nm y0_nm z0_ft v0_ft_s heading0_deg intent for aircraft 1, then 2
 *            (createEncounter.m:41-49)
 *   model_of [4n] i32: index into models[] for track 4e + role, role = 2*(aircraft-1) + (backward)
 *   traj     [2n][W][5] f32: TRACK-MAJOR -- the joined, time-ordered track of aircraft 2e + (aircraft-1), i.e. what
 *            createEncounter.m:74-84 builds ([fwd, bck(1, 2:end)] sorted by t_s): row C+t holds second t (t < 0: the backward
 *            track), fields x_nm y_nm z_ft heading_deg v_ft_s (createEncounter.m:162-167); t_s is the row number minus C and is
 *            not stored.  C = EMGPU_TERMINAL_T0_ROW(cap) (cap rounded up to a multiple of 8: eight rows are 160 bytes, so the pieces
 *            a wave writes start on 32-byte boundaries), W = EMGPU_TERMINAL_BLOCK_ROWS(cap) = 2 C.
 *            Only rows C-(rows_bck-1) .. C+(rows_fwd-1) are written; the rest of the block is left untouched.
 *            (Rounds 1-3 wrote six row-synchronous planes [6][cap][4n]; a lane now starts its next track the moment its own ends, so
 *            the lanes of a wave are at unrelated rows and the output is by track.)
 *   rows     [4n] i32: seconds of track 4e + role, its t = 0 row included (<= tmax_s + 2); negative => cap / resample cap exceeded
 * ---------------------------------------------------------------------------------------------- */
#define EMGPU_TERMINAL_T0_ROW(cap) (((cap) + 7) & ~7)
#define EMGPU_TERMINAL_BLOCK_ROWS(cap) (2 * EMGPU_TERMINAL_T0_ROW(cap))
typedef struct {
    uint64_t seed, first_index;
    int64_t n;                   /* < 2^29 encounters per call                                      */
    double tmax_s;               /* 120 in @CorTerminalModel/track.m:33                            */
    int32_t max_resample, cap;   /* inner re-draw cap (reference: unbounded); rows per direction   */
    double dyn_limits[2][5];     /* per aircraft: minVel_ft_s maxVel_ft_s maxTurnRate_deg_s
                                    maxAltitude_ft maxVertRate_ft_s (getDynamicLimits.m:15-62)     */
    uint32_t flags, _pad;        /* EMGPU_FLAG_LOCAL_SMOOTH: the joined tracks' v_ft_s and z_ft are smoothed in place (:88-89)  */
} emgpu_term_params;
int emgpu_propagate_terminal_device(emgpu_ctx *ctx, const emgpu_model *const *models, int32_t n_models,
                                    const emgpu_term_params *p, const double *geo, const int32_t *model_of,
                                    float *traj, int32_t *rows);
int emgpu_propagate_terminal_host(emgpu_ctx *ctx, const emgpu_model *const *models, int32_t n_models,
                                  const emgpu_term_params *p, const double *geo, const int32_t *model_of,
                                  float *traj, int32_t *rows);

/* CorTerminalModel.sample + createEncounter for n encounters in one call, everything device-resident (RUN_terminal.m:39-44 without the
 * filters of track.m): the geometry draw of @CorTerminalModel/sample.m:29-77 (bn_sample + dediscretize + box / speed rejection), the
 * inputs of createEncounter.m:21-49 (x0 y0 from distance and bearing, the trajectory model of each of the four tracks) and
 * PropagateTrajectory x 4 (createEncounter.m:52-72), three launches on the ctx stream, asynchronous (emgpu_ctx_sync reports a geometry
 * rejection cap or a re-draw cap).  geom_model: the 15-variable geometry network; traj_models[10] in CorTerminalModel.m:84-100 order with
 * the stay prior set.  DEVICE pointers:
 *   geom_bin [n_i][n] u8 (may be NULL), geom_val [n_i][n] f32: the accepted geometry sample;  attempts [n] i32 (may be NULL)
 *   geo [n][12] f64, model_of [4n] i32: the inputs of createEncounter (outputs here);  traj, rows: as emgpu_propagate_terminal_device */
typedef struct {
    uint64_t seed, first_index;
    int64_t n;
    double tmax_s;
    int32_t max_resample, cap;
    double dyn_limits[2][5];
    int32_t max_attempts;            /* cap of the geometry rejection loop (sample.m:32; unbounded in the reference) */
    uint32_t flags;                  /* EMGPU_FLAG_LOCAL_SMOOTH                                              */
    const double *bounds_sample;     /* HOST pointer: n_initial x 2 row-major or NULL (sample.m:45-53)       */
    int32_t idx[12];                 /* 1-based geometry variable ids: own {distance bearing alt speed heading intent}, then int */
} emgpu_tsample_params;
int emgpu_sample_terminal_device(emgpu_ctx *ctx, const emgpu_model *geom_model, const emgpu_model *const *traj_models, int32_t n_traj_models,
                                 const emgpu_tsample_params *p, uint8_t *geom_bin, float *geom_val, double *geo, int32_t *model_of,
                                 float *traj, int32_t *rows, int32_t *attempts);

/* sample2track.m:183-237 -- the 1 Hz dead-reckoning track `sample2track` builds from the em_sample
 * files, and its rejection tests, for n trajectories in one launch:
 *   z += vertrate*ur_vertrate; speed += acc*ur_speed; heading += turnrate*ur_heading;
 *   x += speed*cosd(heading); y += speed*sind(heading)            (:201-212, previous speed/heading)
 * flags[i] bit 0: some z < 0 (CFIT, :234-237); bit 1: some speed <= min or >= max (:240); the
 * reference keeps a track iff flags[i] == 0 (:243).  speed, min_speed, max_speed are in model
 * units (knots) and are converted with ur_speed like :126-139.
 * _device: consumes the sampler's device output in place -- alt0/speed0 = rows of init_val,
 *   dyn_val = the time-blocked dense trace with nd rows per block, slot_* = the rows holding
 *   \dot h, \dot v, \dot\psi; xyz f64 [T+1][3][n], speed_minmax f64 [2][n] (either may be NULL).
 * _host: values as parsed from initial.txt / transition.txt: updates [n][T][3] = vertical rate,
 *   acceleration, turn rate; xyz [n][T+1][3], speed_minmax [n][2] (either may be NULL). */
typedef struct {
    int64_t n;
    int32_t T;                 /* transition rows per id */
    int32_t nd;                /* _device: rows per dense block */
    int32_t slot_vertrate, slot_acc, slot_turnrate;
    int32_t reserved;
    double ur_speed, ur_vertrate, ur_heading;
    double min_speed, max_speed;
} emgpu_track_params;
int emgpu_sample2track_device(emgpu_ctx *ctx, const emgpu_track_params *p, const float *alt0, const float *speed0,
                              const float *dyn_val, double *xyz, uint8_t *flags, double *speed_minmax);
int emgpu_sample2track_host(emgpu_ctx *ctx, const emgpu_track_params *p, const double *alt0, const double *speed0,
                            const double *updates, double *xyz, uint8_t *flags, double *speed_minmax);

/* ------------------------------------------------------------------------------------------------
 * UncorEncounterModel.track (@UncorEncounterModel/UncorEncounterModel.m:318-471) with coordSys 'NEU': per trajectory
 * sample -> dynamics -> the rejection tests of :462-470 against @UncorEncounterModel/getDynamicLimits.m:1-130, retried
 * with the next seed (:424-428: attempt j of EVERY trajectory uses the Philox key seed + j; the trajectory is told apart by
 * its global index).  Rounds: round j samples the trajectories still rejected (round 0: the whole contiguous range on the
 * fast kernels, later rounds an index list), integrates and tests them on the device; only a counter crosses PCIe per round.
 * DYNAMICS: the reference calls em-core's run_dynamics_fast / computeVerticalRate, which it does not vendor.  The point-mass
 * model used instead (dt 0.1 s, first-order pitch / bank response limited by dyn(5:6); "dynamics unpinned") is stated in
 * HISTORY.md section 10 and at the top of csrc/emgpu_kernels_utrack.hip.  The 'geodetic' branch (:480-540: DEM, obstacles, placeTrack) is out of scope.
 *   tracks   [n][S][8] f64: time_s north_ft east_ft up_ft speed_ft_s phi_rad theta_rad psi_rad, S = 10*T/record_stride + 1
 *   limits   [n][3] f64: minVel_ft_s maxVel_ft_s maxVertRate_ft_s of the accepted attempt (getDynamicLimits.m:129-133)
 *   attempts [n] i32: attempts used (>= 1); -1 => max_track_attempts reached (the call then returns EMGPU_ERR_REJECT_CAP)
 * Any output may be NULL.  _host: host pointers; _device: device pointers.  Both synchronise the ctx stream.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t seed, first_index;  /* 'initialSeed' (:327)                                               */
    int64_t n;
    int32_t sample_time;         /* seconds                                                            */
    uint32_t flags;              /* EMGPU_FLAG_QUANTIZE500                                             */
    int32_t max_track_attempts;  /* cap of the outer while (:421; unbounded in the reference)          */
    int32_t max_attempts;        /* cap of .sample's own loop (:248)                                   */
    int32_t idx_G, idx_A, idx_L, idx_v, idx_dv, idx_dh, idx_dpsi; /* 1-based variable ids (:385-391), 0 = absent */
    int32_t is_rotorcraft;       /* self.isRotorcraft (:181-185)                                       */
    int32_t record_stride;       /* keep every record_stride-th 0.1 s step: 1 = the reference's timetable, 10 = 1 Hz */
    int32_t _pad;
} emgpu_utrack_params;
int emgpu_track_uncor_host(emgpu_ctx *ctx, const emgpu_model *m, const emgpu_utrack_params *p,
                           double *tracks, double *limits, int32_t *attempts);
int emgpu_track_uncor_device(emgpu_ctx *ctx, const emgpu_model *m, const emgpu_utrack_params *p,
                             double *tracks, double *limits, int32_t *attempts);
/* getDynamicLimits.m:1-130 for one trajectory on the host (what the table of the track kernel holds): initial = the
 * n_initial sampled values (bins for categorical variables), the extrema over the 10 Hz result in ft and ft/s. */
int emgpu_uncor_dynamic_limits(const emgpu_model *m, const emgpu_utrack_params *vars, const double *initial,
                               double up_min_ft, double up_max_ft, double speed_min_ft_s, double speed_max_ft_s, double out[3]);

/* ------------------------------------------------------------------------------------------------
 * CorTerminalModel.track (@CorTerminalModel/track.m:45-150): per encounter, loop { geometry sample (sample.m) ->
 * createEncounter -> the filters of :62-145 (CPA within +-10 s, overlap >= minEncTime_s, runway proximity, vertical intent,
 * CheckDynamicLimits with CheckCumTurn: CorTerminalModel.m:117-316) } until one passes.  Rounds on the device like
 * emgpu_track_uncor_*: attempt j of every encounter uses the Philox key seed + j (the reference continues one MT19937 stream,
 * track.m:36-58) and the encounter's global index; a track that hits the re-draw cap voids its attempt.
 * em-core's computeVerticalRate / computeHeadingRate (not vendored) are forward differences of the 1 s samples; `isClimb`
 * (track.m:122,134, undefined in the reference) is read as is_climb; local_smooth (createEncounter.m:88-89) is the flagged stand-in of
 * EMGPU_FLAG_LOCAL_SMOOTH (off: the filters read the unsmoothed tracks).
 *   geom_model   the 15-variable geometry network; traj_models[10] in CorTerminalModel.m:84-100 order with the stay prior set
 *   sample   [n][n_initial(geom)] f64   the accepted geometry sample (out_results(ii).sample, track.m:158)
 *   traj     [n][2][cap2][6] f64        ownship / intruder, time-ordered: t_s x_nm y_nm z_ft heading_deg v_ft_s (createEncounter.m:74-84)
 *   len      [n][2] i32 rows of each;  meta [n][4] f64: tcpa_s hmd_ft vmd_ft enc_time_s (:79, :90);  attempts [n] i32 (-1: cap)
 * Host pointers; any output may be NULL.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t seed, first_index;      /* 'initialSeed' (track.m:11)                                         */
    int64_t n;
    double tmax_s;                   /* 120 (track.m:33)                                                   */
    int32_t max_resample;            /* re-draw cap inside PropagateTrajectory                             */
    int32_t max_track_attempts;      /* cap of the while of track.m:55 (unbounded in the reference)        */
    int32_t max_attempts;            /* cap of the geometry rejection loop (sample.m:32)                   */
    uint32_t flags;                  /* EMGPU_FLAG_LOCAL_SMOOTH: the filters read the smoothed tracks, like the reference's (:88-89) */
    double dyn_limits[2][5];         /* per aircraft: minVel maxVel maxTurnRate_deg_s maxAltitude maxVertRate (getDynamicLimits.m:15-62) */
    double max_cum_turn_deg[2], pitch_deg[2];
    double min_enc_time_s, thres_dist_ft, thres_alt_low_ft, thres_vertrate_ft_s;  /* track.m:14-17 */
    const double *bounds_sample;     /* n_initial x 2 row-major or NULL (sample.m:45-53)                   */
    int32_t idx[12];                 /* 1-based geometry variable ids: own {distance bearing alt speed heading intent}, then int */
} emgpu_ttrack_params;
int emgpu_track_terminal_host(emgpu_ctx *ctx, const emgpu_model *geom_model, const emgpu_model *const *traj_models, int32_t n_traj_models,
                              const emgpu_ttrack_params *p, double *sample, double *traj, int32_t cap2, int32_t *len,
                              double *meta, int32_t *attempts);

/* Introspection for benchmarks/tests: name of the kernel variant the last *_device call used and
 * the algorithmic output bytes per trajectory of that call (5*n_i + 5*T*n_d for dense output). */
const char *emgpu_last_kernel_name(const emgpu_ctx *ctx);
/* Kernel launches the last emgpu_sample_dbn_*_device call on this ctx issued (a mixed batch whose models share a kernel
 * instance is ONE launch with the model id per workgroup; RUN_1_emsample.m:13,24-47 is the reference's per-file loop). */
int32_t emgpu_last_launch_count(const emgpu_ctx *ctx);

/* Test hooks into the plan compiler (host only): the u32 quantile thresholds of one CPT column
 * (r weights -> r-1 thresholds; bin = 1 + #{k : min(x, 2^32-2) >= out[k]} reproduces
 * select_random.m:17-20 for u = (min(x, 2^32-2) + 0.5) * 2^-32) and the resample hit threshold
 * (hit <=> min(x, 2^32-2) < R reproduces `u < rate`, resample_events.m:24). */
int emgpu_debug_column_thresholds(const double *weights, int32_t r, uint32_t *out);
uint32_t emgpu_debug_bernoulli_threshold(double rate);
/* Column `col` (0-based, asub2ind order) of the k-th dynamic variable's transition table as the
 * kernels read it: the r-1 quantile thresholds (thr, room for 15) and the compacted form -- *meff
 * distinct thresholds (cthr, room for 7) plus the nibble map, bin = (map >> 4n) & 15 with
 * n = #{t < meff : min(x, 2^32-2) >= cthr[t]}.  *meff = 0: this variable is not compacted.
 * k counts the temporal_map rows in plan order; *tvar receives the 1-based transition variable. */
int emgpu_debug_dynamic_column(const emgpu_model *m, int32_t k, int64_t col, int32_t *tvar, int32_t *r, int64_t *q,
                               uint32_t *thr, int32_t *meff, uint32_t *cthr, uint32_t *map);
/* The same column in the padded form the per-timestep kernel loads: *width = 4 words {t0, t1, t2, map} or
 * 8 words {t0..t5, map_lo, map_hi} (0: the variable has no padded table); unused thresholds repeat the last real one (2^32-1 if none);
 * the map is a byte table: entry b = 1-based bin when b of the 3 (6) thresholds did NOT fire. */
int emgpu_debug_padded_column(const emgpu_model *m, int32_t k, int64_t col, int32_t *width, uint32_t *words);
/* The same column in the packed-compare form of the per-timestep kernel (4 words): {T'0 | T'1 << 16, T'2 | T'3 << 16, T'4 | T'5 << 16,
 * nibble map}.  With x_h the draw's high halfword and d_t = min(sat16(x_h - T'_t), 2): the sum over t is 2 * (thresholds fired),
 * odd exactly when the draw's low halfword decides some compare; nibble (sum / 2) of the map is the 1-based bin
 * (select_random.m:17-20 on the high halfword alone; x_h = 0 is always referred to the full 32-bit compare).
 * A variable whose padded width is 4 (at most 3 thresholds) holds the PLAIN form instead: {H0, H1, H2, map}, H_t = the threshold's high
 * half (0x10000: no such threshold); with a_t = H_t - x_h: a_t < 0 <=> threshold t fired, a_t == 0 <=> the low halfword decides; the
 * 1-based bin is (map >> 7 * fired) & 15. */
int emgpu_debug_pk_column(const emgpu_model *m, int32_t k, int64_t col, uint32_t *words);

/* Test hook: the point-mass dynamics kernel of emgpu_track_uncor_* (k_uncor_track) on caller-given inputs, without sampling or limits.
 *   init [n][5] f32: L (ft), v (kt), \dot v (kt/s), \dot h (ft/min), \dot psi (deg/s) -- the sampled values of UncorEncounterModel.m:434-446
 *   controls [n][T][3] f32: \dot h (ft/min), \dot psi (deg/s), \dot v (kt/s) active during each second (events2controls.m:16-27)
 *   dyn[6]: v_low v_high dh_min dh_max q_max r_max (:414);  literal != 0: the literal step even where the reduced one applies
 *   tracks [n][10 T / record_stride + 1][8] f64 as emgpu_track_uncor_host.  Host pointers. */
int emgpu_debug_uncor_dynamics_host(emgpu_ctx *ctx, int64_t n, int32_t T, int32_t record_stride, int32_t literal, const double dyn[6],
                                    const float *init, const float *controls, double *tracks);

/* Measuring builds of k_terminal_propagate (-DEMGPU_TERM_COUNTERS; tools/term_counters.py): the lanes that took each path of the loop
 * since the last call (24 counters, cleared by the call).  Returns 1, or 0 in a normal build (out untouched). */
int emgpu_debug_terminal_counters(emgpu_ctx *ctx, uint64_t *out, int32_t n);

/* Which dynamic variables are parents of which (t+1) node, in plan order k = 0..n_dyn-1: bit 4k+q of cur_mask = the time-t
 * node of dynamic variable q is a parent of k's (t+1) node; of new_mask = its (t+1) node is (dbn_sample.m:65-93: the
 * dependent branch; q < k by the topological order).  The per-timestep kernel picks its instance by these masks. */
int emgpu_debug_parent_masks(const emgpu_model *m, uint32_t *cur_mask, uint32_t *new_mask);

/* Host helpers that mirror small reference functions (used by the class layer and tests). */
int32_t emgpu_discretize_bayes(double x, const double *thresholds, int32_t n); /* discretize_bayes.m:14-22 */
int64_t emgpu_asub2ind(const int32_t *siz, const int32_t *x, int32_t n);       /* asub2ind.m:13-14        */

#ifdef __cplusplus
}
#endif
#endif /* EMGPU_H */
