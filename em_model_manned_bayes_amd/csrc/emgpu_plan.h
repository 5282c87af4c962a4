// emgpu_plan.h -- POD "plan" handed to the HIP kernels (by value, in the kernarg segment) and the
// RNG slot map shared by every kernel.  Built on the host by emgpu_model.cpp (compile_plan).
//
// The plan restates a model in the form the device wants:
//   * the initial network relabelled into topological order (bn_sort.m / bn_sample.m:41), so the
//     kernel walks positions 0..ni-1 and never indexes registers dynamically;
//   * asub2ind (asub2ind.m:13-14) folded into per-parent column strides (0 for non-parents);
//   * select_random (select_random.m:14-20) folded into u32 "quantile thresholds": for CPT column
//     j of a node with r bins, X[j][k] (k < r-1) is the smallest 32-bit draw x for which
//     s_k < s_end * u(x); the sampled bin is 1 + #{k : x >= X[j][k]}.  Exactly equivalent to the
//     reference's f64 compare for the uniform u(x) defined below.
#pragma once
#include <stdint.h>

#define EMGPU_MAX_NI 16 // initial-network variables
#define EMGPU_MAX_ND 4  // dynamic variables (rows of the temporal map)
#define EMGPU_MAX_R 64  // bins per variable
#define EMGPU_MAX_MIXED 16 // model blocks of a mixed batch served by ONE launch (k_uncor_fast_mixed)

// ---- RNG slot map -------------------------------------------------------
// Philox4x32-R (R = EMGPU_PHILOX_ROUNDS below), key = {seed_lo, seed_hi},
// ctr = {gidx_lo, gidx_hi, attempt, section<<28 | a<<20 | block}.
// uniform: x' = min(x, 2^32-2); u = (x' + 0.5) * 2^-32.
// Rounds of the Philox4x32 bijection: 7, the fewest that Salmon et al. (SC'11, table 2) report as Crush-resistant (BigCrush passed
// with sequential counters -- exactly how the slot map uses it); Random123's default of 10 adds a safety margin that costs this
// VALU-bound path 7-9 % (same-box A/B in HISTORY.md section 5; builds before round 3 were specified at 10).  The test suite's
// CPU checker is built with the same value (its EM_PHILOX_ROUNDS): a different value is a different generator.
#ifndef EMGPU_PHILOX_ROUNDS
#define EMGPU_PHILOX_ROUNDS 7
#endif
#define EMGPU_SEC_INIT 1u
#define EMGPU_SEC_DEDISC_INIT 2u
#define EMGPU_SEC_TRANS 3u
#define EMGPU_SEC_RES 4u
#define EMGPU_SEC_DEDISC_RES 5u
#define EMGPU_SEC_DEDISC_TRANS 6u
#define EMGPU_SEC_LAYER 7u
#define EMGPU_SEC_GEOM_DEDISC 8u
// TRANS and RES are SPLIT slots: x = (H << 16) | L, H = halfword (idx & 7) of block
// (section, a, idx >> 3), L = halfword (idx & 7) of block (section + 6, a, idx >> 3); halfword h of
// a block = word h >> 1, upper 16 bits when h is odd, lower 16 bits when h is even.  Every other
// section is a WORD slot: word (idx & 3) of block (section, a, idx >> 2).
#define EMGPU_SEC_TRANS_LO 9u
#define EMGPU_SEC_RES_LO 10u

struct EmgpuPlan {
    int32_t ni, nd, nact, depend; // depend: is_dynvar_depend (dbn_sample.m:55)
    uint32_t thr_total;           // entries in thr[]; the dynamic variables' tables are thr[d_off[0] .. thr_total)
    uint32_t cthr_total;          // entries in cthr[]
    // ---- initial network, by topological position p
    uint8_t i_var[EMGPU_MAX_NI];   // 0-based variable id
    uint8_t i_r[EMGPU_MAX_NI];     // bins
    uint8_t i_start[EMGPU_MAX_NI]; // 0 = unset, else preset bin (1-based)
    uint32_t i_off[EMGPU_MAX_NI];  // offset of the node's thresholds in thr[]
    uint32_t i_stride[EMGPU_MAX_NI][EMGPU_MAX_NI]; // [p][q<p]: column stride of parent at position q
    // ---- dediscretize, by topological position p (dediscretize.m:7-40)
    uint8_t i_nb[EMGPU_MAX_NI];    // number of boundaries (0 => categorical: value = bin)
    uint8_t i_zero[EMGPU_MAX_NI];  // zero bin (0 = none)
    uint8_t i_skip[EMGPU_MAX_NI];  // dbn_hierarchical_sample.m:26 length(params)==r-2 branch
    uint16_t i_boff[EMGPU_MAX_NI]; // offset into bnd[]
    // ---- dynamic variables, k in sampling order (order_transition restricted to dynamic vars)
    uint8_t d_tvar[EMGPU_MAX_ND]; // 0-based id of the (t+1) node in the transition network (RNG slot)
    uint8_t d_ivar[EMGPU_MAX_ND]; // 0-based id of the mapped initial variable (RNG slot, event var)
    uint8_t d_ipos[EMGPU_MAX_ND]; // topological position of that initial variable
    uint8_t d_r[EMGPU_MAX_ND];
    uint8_t d_row[EMGPU_MAX_ND];  // row of the temporal map == output slot (ascending variable id)
    uint8_t d_emit[EMGPU_MAX_ND]; // d_emit[e] = k of the e-th dynamic variable in ascending ivar order
    uint8_t d_nb[EMGPU_MAX_ND], d_zero[EMGPU_MAX_ND];
    uint16_t d_boff[EMGPU_MAX_ND];
    uint32_t d_off[EMGPU_MAX_ND];
    // rows of 9..48 thresholds also have a pivot row of 8 words in thr[] (offset d_pivoff, past thr_total; 0 = none): every 6th
    // threshold (index 6q + 5) of the row's full groups but the last, 2^32-1 elsewhere -- k_terminal_propagate finds the group of a
    // draw from two 16-byte loads instead of seven strided ones
    uint32_t d_pivoff[EMGPU_MAX_ND];
    // the same rows also have a COMPACT form of 8 words in thr[] (offset d_c8off, after the pivot rows; 0 = none): a row's distinct
    // thresholds that can fire (0 < X < 2^32-1), ascending, padded with 2^32-1 to six, then the byte map "n of them fired -> 1-based
    // bin" (7 bytes); byte 7 = 0xFF when the row has more than six (the kernel then takes the pivot path for that lane).  Zero-count
    // bins make most thresholds of a 36-bin row coincide (a heading moves to a neighbouring bin or stays): one 32-byte gather
    // decides the draw where pivots + group take two dependent ones, from a table a quarter of the size.
    uint32_t d_c8off[EMGPU_MAX_ND];
    uint32_t d_stride_static[EMGPU_MAX_ND][EMGPU_MAX_NI]; // parents that never change, by position p
    uint32_t d_stride_cur[EMGPU_MAX_ND][EMGPU_MAX_ND];    // time-t node of dynamic var k' as parent
    uint32_t d_stride_new[EMGPU_MAX_ND][EMGPU_MAX_ND];    // (t+1) node of dynamic var k' (sampled earlier)
    // ---- resample_events.m:24: variables with rate > 0, ascending variable id
    uint8_t a_var[EMGPU_MAX_NI];  // 0-based variable id
    uint8_t a_pos[EMGPU_MAX_NI];  // topological position
    int8_t a_dyn[EMGPU_MAX_NI];   // k if dynamic else -1
    uint32_t a_R[EMGPU_MAX_NI];   // hit  <=>  x' < R
    // ---- compacted tables of the dynamic variables (see compile_plan): column j of variable k is
    // cthr[d_coff[k] + j*(d_meff[k]+1) ...]: d_meff[k] DISTINCT real thresholds (padded with copies of the last one)
    // followed by one word of nibbles, bin(n) = (map >> 4n) & 15 for n = #{t : x' >= threshold t}.
    // Zero-count bins make most of a column's r-1 thresholds coincide, so d_meff is often far below
    // r-1 (uncor_1200code_v2p1: 2, 4, 2 instead of 4, 6, 6).  d_meff[k] == 0 => not compacted.
    uint8_t d_meff[EMGPU_MAX_ND];
    uint32_t d_coff[EMGPU_MAX_ND];
    const uint32_t *cthr;
    // ---- the same columns padded to a power of two for the per-timestep kernel (one or two 16-byte
    // loads per draw): d_pw[k] = 4 words {t0, t1, t2, map} (d_meff <= 3) or 8 words {t0..t5, map_lo,
    // map_hi} (d_meff <= 6); unused thresholds repeat the last real one; the map is a BYTE table indexed by the
    // number of thresholds that did NOT fire (b = 3 or 6 minus the fired count) -> 1-based bin, ready
    // for v_perm_b32.  d_pw[k] == 0 => no padded table for this variable.
    uint8_t d_pw[EMGPU_MAX_ND];
    uint32_t d_poff[EMGPU_MAX_ND];
    // every padded variable also has the PACKED-COMPARE form (word offset d_poffpk, 4 words per column):
    // {T'0 | T'1 << 16, T'2 | T'3 << 16, T'4 | T'5 << 16, nibble map}.  T'_t is the 16-bit value for which d = sat(x_h - T'_t) reads
    // 0: threshold t not fired, 1: the low halfword decides (a tie with the threshold's high half), >= 2: fired -- H_t - 1, made
    // strictly increasing along the column, 0 for H_t = 0 (x_h = 0 is then a tie by a separate test), 0xFFFF beyond the column's
    // thresholds (exactly load_cthr_pk of the fast kernel, built on the host because the column changes every second here).
    // The sum of min(d, 2) over the thresholds is 2 * fired, odd exactly when some compare needs the low halfword; nibble n of the
    // map is the 1-based bin when n thresholds fired.  Three v_pk_sub_u16 / v_pk_min_u16 pairs decide a draw: no carries.
    // A 4-word variable (at most 3 thresholds) holds the PLAIN form in the same place: {H0, H1, H2, map}, H_t = the threshold's high half as a
    // 32-bit word (0x10000: none), bins 7 bits apart in the map: a_t = H_t - x_h is negative when threshold t fired and 0 on a tie;
    // sum of (a_t >>> 29) = 7 * fired = the bin's bit offset.  Plain VOP2 subtracts and shifts issue at twice the rate of the packed ops.
    uint32_t d_poffpk[EMGPU_MAX_ND];
    uint32_t pthr_total, _pad1;
    const uint32_t *pthr;
    // ---- device tables
    const uint32_t *thr; // quantile thresholds, node after node, column after column, r-1 each
    const double *bnd;   // boundaries
};

struct EmgpuRun {
    uint64_t seed, first_index;
    int64_t n;
    int64_t ld; // trajectory dimension of the output arrays (>= n: a shard may be written into a larger shared trace)
    int64_t col0; // column of the trace the output pointers below already point at: kernels that care line their workgroups up with
                  // the TRACE's columns (multiples of 256), not with the shard's first one -- a wave store that straddles 128-byte lines
                  // costs this write-bound path 20-35 % (measured: 6.25 M columns, shards at odd offsets)
    int32_t T, per_step;
    uint32_t flags;
    int32_t max_attempts;
    int32_t pos_L, pos_v, pos_dh; // topological positions, -1 = absent
    int32_t n_layers;
    const double *layers;
    int32_t event_cap, _pad;
    // outputs
    uint8_t *init_bin;
    float *init_val;
    uint32_t *dyn_bin;
    float *dyn_val;
    uint32_t *ev_count;
    uint64_t *events; // emgpu_event rows as packed 64-bit words
    int32_t *attempts;
    uint32_t *status; // device word: bit0 = rejection cap hit, bit1 = event cap hit
    const uint64_t *indices; // optional: global index of lane i (instead of first_index + i); every DBN kernel but the round-1 k_dbn_step
};
// per-sample presets (a start GRID in one launch: InitStartTerminal.m:57-90, UncorEncounterModel.m:204): a block of device memory handed to
// k_dbn_generic as an argument of its own.  NOT part of EmgpuRun: the benchmark kernel sits at the edge of its scalar registers, and a
// pointer it never reads cost it 45 more spill reloads in its loops (+3.7 % vector instructions, measured).
struct EmgpuPresets {
    const int32_t *start;    // [n][ni] by variable id, 0 = unset (then the model's own start applies); null: the model's start for every lane
    double *log_weight;      // [n] sum over the lane's preset nodes of log P(preset | parents); null: not wanted
    const double *logp;      // log of the column-normalised (N + alpha) of the initial network, node after node (by position), column after column
    uint32_t lp_off[EMGPU_MAX_NI];
};

struct EmgpuBnRun {
    uint64_t seed, first_index;
    int64_t n;
    int64_t ld; // trajectory dimension of out_bin / out_val
    uint32_t flags;
    int32_t max_attempts;
    int32_t has_bounds, pos_own_speed, pos_int_speed, _pad;
    double bounds[EMGPU_MAX_NI][2]; // by topological position
    double min1, max1, min2, max2;
    uint8_t *out_bin;
    float *out_val;
    int32_t *attempts;
    uint32_t *status;
    const uint64_t *indices; // optional: global index of lane i (instead of first_index + i)
    const int32_t *start;    // per-sample presets, log-weights and the table they read: as in EmgpuRun
    double *log_weight;
    const double *logp;
    uint32_t lp_off[EMGPU_MAX_NI];
};

struct EmgpuTermRun {
    uint64_t seed, first_index;
    int64_t n;                       // encounters; 4 lanes each
    const double *geo;               // [n][12]: x0 y0 z0 v0 heading0 intent for aircraft 1, then aircraft 2
    const int32_t *model_of;         // [4n]: index into thr_base for lane 4e + role
    const uint32_t *const *thr_base; // [n_models] first dynamic-variable table of each trajectory model (same shapes)
    double tmax_s;
    double dl[2][5];                 // minVel, maxVel, maxTurnRate, maxAltitude, maxVertRate per aircraft
    int32_t max_resample, cap;
    float *traj;                     // [2n][W][5]: the joined track of aircraft 2e + a (createEncounter.m:74-84), row C + t for second t (C = EMGPU_TERMINAL_T0_ROW(cap), W = 2 C):
                                     // x_nm y_nm z_ft heading_deg v_ft_s (t_s is the row number; rows outside a track's span are not written)
    int32_t *rows;                   // [4n] rows of track 4e + 2a + (backward), its t = 0 row included; < 0: failed (cap / resample cap)
    uint32_t *status;
    uint32_t *queue;                 // device word, zeroed by the launcher: the next track nobody has taken
    const uint64_t *indices;         // optional: global index of encounter e (instead of first_index + e)
    int32_t quiet;                   // a failed track only marks rows < 0 (CorTerminalModel.track re-draws it) instead of raising the status bit
};

// CorTerminalModel.track (track.m:45-150): geometry sample -> createEncounter inputs, and the filters on the propagated tracks.
struct EmgpuTGeoRun {
    int64_t n;
    const float *val;                // [n_i][n] geometry sample (k_bn)
    int32_t idx[12];                 // 0-based rows: own {distance bearing alt speed heading intent}, then int
    double *geo;                     // [n][12]
    int32_t *model_of;               // [4n]
};
struct EmgpuTFilterRun {
    int64_t n;                       // encounters of this round
    const float *tracks;             // [2n][EMGPU_TERMINAL_BLOCK_ROWS(cap)][5] from k_terminal_propagate
    const int32_t *rows;             // [4n]
    int32_t cap;
    const double *geo;               // [n][12] (intents)
    const float *val;                // [n_i][n] geometry sample of this round
    int32_t n_i;
    double dl[2][5];                 // minVel maxVel maxTurnRate maxAltitude maxVertRate per aircraft
    double max_cum_turn[2], pitch[2];
    double min_enc_time_s, thres_dist_ft, thres_alt_low_ft, thres_vertrate_ft_s;
    const int64_t *slot;             // [n] output position (null: i)
    uint8_t *accepted;               // [n]
    // outputs by slot (any may be null)
    double *sample;                  // [slots][n_i]
    double *traj;                    // [slots][2][cap2][6]
    int32_t cap2;
    int32_t *len;                    // [slots][2]
    double *meta;                    // [slots][4]: tcpa_s hmd_ft vmd_ft enc_time_s
    int32_t *attempts;               // [slots]
    int32_t attempt_no, last_round;
};

// UncorEncounterModel.track (UncorEncounterModel.m:419-471): point-mass dynamics over the sampler's dense trace and the
// three rejection tests, one lane per trajectory (k_uncor_track, emgpu_kernels_utrack.hip).
struct EmgpuUTrackRun {
    int64_t n;                 // lanes (trajectories of this round)
    int64_t ld;                // trajectory dimension of the sampler's buffers
    int32_t T, stride;         // seconds; keep every stride-th 0.1 s step
    const float *iG, *iA, *iL, *iV, *iDV, *iDH, *iDPsi; // rows of init_val (iG / iA may be null when the limits are global)
    const float *dyn_val;      // [ceil(T/4)][nd][ld][4]
    int32_t nd, sDV, sDH, sDPsi; // rows of the dense trace
    const int64_t *slot;       // [n] output position of lane i (null: i)
    double dyn[6];             // v_low v_high dh_min dh_max qmax rmax            (:414)
    double min_alt, max_alt;   //                                                  (:397-405)
    int32_t ordered, rG, rA, rL, rV, ncL, ncV, discL, discV; // getDynamicLimits.m:17 branch and its dimensions
    double cutL[16], cutV[16]; // cutpoints_initial{L}, {V}
    const double *lim;         // ordered: [rG][rA][rL][rL][rV][rV][3]; else [3]: minVel maxVel maxVertRate
    double *tracks;            // [slots][S][8] time north east up speed phi theta psi (may be null)
    int64_t S;
    double *limits;            // [slots][3] (may be null)
    uint8_t *accepted;         // [n] by lane
    int32_t *attempts;         // [slots]: written with attempt_no by accepted lanes, with -1 by rejected lanes of the last round
    int32_t attempt_no, last_round;
};

// sample2track.m:183-237 for n trajectories (k_sample2track).  Exactly one input form is set.
struct EmgpuTrackRun {
    int64_t n;
    int32_t T;                              // transition rows per id (= num_transition_samples of em_sample)
    double ur_speed, ur_vertrate, ur_heading; // sample2track.m:113-123
    double min_speed, max_speed;            // boundaries{speed}([1 end]) in model units (sample2track.m:104-105)
    // DENSE: the sampler's device output
    const float *alt0_f, *speed0_f;         // [n]: rows of init_val
    const float *dyn_val;                   // [ceil(T/4)][nd][n][4]
    int32_t nd, s_vr, s_acc, s_tr;          // rows of vertical rate, acceleration, turn rate
    // PLANAR: f64 columns parsed from the files
    const double *alt0_d, *speed0_d;        // [n]
    const double *upd;                      // [T][3][n]: vertical rate, acceleration, turn rate in model units
    // outputs (each may be null)
    double *xyz;                            // [T+1][3][n] feet
    uint8_t *flags;                         // [n]: bit 0 = below ground (CFIT), bit 1 = speed outside (min, max)
    double *vmm;                            // [2][n]: min and max speed (ft/s)
};
