// emgpu_kernels.hip -- HIP kernels of the sampling path (gfx950 / CDNA4), one lane per trajectory.
//
//   k_dbn_generic<NI,ND,NA> : every model shape, both transition semantics (dbn_sample.m FAST and
//                             DEPENDENT branches), dense and/or event-list output, rejection loop,
//                             layers / quantize500, presets.  Compile-time maxima, run-time counts.
//   k_bn<NI>                : bn_sample + dediscretize + rejection (CorTerminalModel geometry draw).
//   k_uncor_fast (emgpu_kernels_fast.hip) : the benchmarked specialisation.
//
// Reference lines restated here: bn_sample.m:39-57, dbn_sample.m:36-166, resample_events.m:16-37,
// dediscretize.m:7-40, dbn_hierarchical_sample.m:9-37, UncorEncounterModel.m:244-281,
// @CorTerminalModel/sample.m:29-77.  This is a gather + RNG + store path: no MFMA.
#include <hip/hip_runtime.h>

#include "../../include/emgpu.h"
#include "emgpu_device.h"
#include "emgpu_launch.h"

namespace emgpu {

template <int NI, int ND, int NA>
__global__ void __launch_bounds__(256) k_dbn_generic(const EmgpuPlan P, const EmgpuRun A, const EmgpuPresets *Q /* a start grid / log-weights, or null */) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n) return;
    const uint64_t gidx = A.indices ? A.indices[i] : A.first_index + (uint64_t)i;
    Rng rng{(uint32_t)gidx, (uint32_t)(gidx >> 32), 0u, (uint32_t)A.seed, (uint32_t)(A.seed >> 32)};
    const bool no_dedisc = (A.flags & EMGPU_FLAG_NO_DEDISC) != 0;
    const bool no_resample = (A.flags & EMGPU_FLAG_NO_RESAMPLE) != 0;
    const bool want_events = A.ev_count != nullptr;
    const int T = A.T;

    int bin[NI];     // 0-based, by topological position
    double val[NI];  // dediscretised (f64 like the reference; rounded to f32 on store)
#pragma unroll
    for (int p = 0; p < NI; p++) { bin[p] = 0; val[p] = 0.0; }

    const int32_t attempts_used = init_network_ps<NI>(P, A, Q, rng, bin, val, i);   // (the one DBN kernel that takes a start grid)
    if (attempts_used < 0) atomicOr(A.status, 1u);
    if (A.attempts) A.attempts[i] = attempts_used;
#pragma unroll
    for (int p = 0; p < NI; p++) {
        if (p >= P.ni) continue;
        if (A.init_bin) A.init_bin[(size_t)P.i_var[p] * A.ld + i] = (uint8_t)(bin[p] + 1);
        if (A.init_val) A.init_val[(size_t)P.i_var[p] * A.ld + i] = (float)val[p];
    }
    if (P.nd == 0 && !want_events) return;

    // ---------------- transition network ---------------------------------------------------------
    int cur[ND];
    float cval[ND];
    uint32_t basecol[ND];
    const uint32_t *tptr[ND];
#pragma unroll
    for (int k = 0; k < ND; k++) {
        cur[k] = 0; cval[k] = 0.f; basecol[k] = 0; tptr[k] = P.thr;
        if (k >= P.nd) continue;
        cur[k] = pick<NI>(bin, P.d_ipos[k]);
        cval[k] = (float)pick<NI>(val, P.d_ipos[k]);
        uint32_t b = 0;
#pragma unroll
        for (int p = 0; p < NI; p++) b += P.d_stride_static[k][p] * (uint32_t)bin[p];
        basecol[k] = b;
    }
    const bool per_step = A.per_step != 0 || P.depend != 0; // dbn_sample.m:55,65
    if (!per_step) {
        // FAST branch: parent configuration frozen at the initial state (dbn_sample.m:110-135)
#pragma unroll
        for (int k = 0; k < ND; k++) {
            if (k >= P.nd) continue;
            uint32_t col = basecol[k];
#pragma unroll
            for (int q = 0; q < ND; q++) col += P.d_stride_cur[k][q] * (uint32_t)cur[q];
            tptr[k] = P.thr + P.d_off[k] + (size_t)col * (uint32_t)(P.d_r[k] - 1);
        }
    }

    uint32_t ecount = 0;
    int last_t = 0;
    uint64_t *ev = want_events ? A.events + (size_t)i * (size_t)A.event_cap : nullptr;
    auto emit = [&](int at, int var1, int bin1, float v) {
        const uint32_t dt = (uint32_t)(at - last_t);
        last_t = at;
        if (ecount < (uint32_t)A.event_cap)
            ev[ecount] = (uint64_t)(dt & 0xFFFFu) | ((uint64_t)(uint32_t)var1 << 16) | ((uint64_t)(uint32_t)bin1 << 24) | ((uint64_t)__float_as_uint(v) << 32);
        ecount++;
    };

    const int G4 = (T + 3) >> 2;
    const int G8 = (want_events && !no_resample) ? (T >> 3) + 1 : (T + 7) >> 3;
    for (int g8 = 0; g8 < G8; g8++) {
        // split slots (TRANS, RES): primary and secondary block of the 8 seconds 8*g8 .. 8*g8+7
        uint4 th[ND], tl[ND], rh[NA], rl[NA];
#pragma unroll
        for (int k = 0; k < ND; k++) {
            th[k] = tl[k] = make_uint4(0, 0, 0, 0);
            if (k < P.nd) {
                th[k] = rng.block(EMGPU_SEC_TRANS, P.d_tvar[k], (uint32_t)g8);
                tl[k] = rng.block(EMGPU_SEC_TRANS_LO, P.d_tvar[k], (uint32_t)g8);
            }
        }
#pragma unroll
        for (int a = 0; a < NA; a++) {
            rh[a] = rl[a] = make_uint4(0, 0, 0, 0);
            if (a < P.nact && !no_resample) {
                rh[a] = rng.block(EMGPU_SEC_RES, P.a_var[a], (uint32_t)g8);
                rl[a] = rng.block(EMGPU_SEC_RES_LO, P.a_var[a], (uint32_t)g8);
            }
        }
        uint32_t pb[ND][2];
        float pv[ND][2][4];
#pragma unroll
        for (int k = 0; k < ND; k++)
#pragma unroll
            for (int h = 0; h < 2; h++) { pb[k][h] = 0; pv[k][h][0] = pv[k][h][1] = pv[k][h][2] = pv[k][h][3] = 0.f; }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int c = 8 * g8 + j; // absolute event time == column produced
            const int w = j & 3, hb = j >> 2;
            const uint32_t g4 = (uint32_t)(2 * g8 + hb); // word-slot block of time c
            if (c >= 1 && c <= T) {
                // ---- resample_events.m:23-29: hits of the c-th second, ascending variable id
                if (!no_resample) {
#pragma unroll
                    for (int a = 0; a < NA; a++) {
                        if (a >= P.nact) continue;
                        const bool hit = clamp32(split_draw(rh[a], rl[a], j)) < P.a_R[a];
                        const int k = P.a_dyn[a];
                        const int b0 = (k >= 0) ? pick<ND>(cur, k) : pick<NI>(bin, P.a_pos[a]);
                        if (hit) {
                            const int nb = P.i_nb[P.a_pos[a]], zero = P.i_zero[P.a_pos[a]];
                            float v = (float)(b0 + 1);
                            if (!no_dedisc && nb != 0) {
                                if (zero == b0 + 1) v = 0.f;
                                else v = (float)dedisc_f64(P.bnd, P.i_boff[P.a_pos[a]], b0,
                                                           word_of(rng.block(EMGPU_SEC_DEDISC_RES, P.a_var[a], g4), w));
                            }
                            if (k >= 0) put<ND>(cval, k, v);
                            if (want_events) emit(c, (int)P.a_var[a] + 1, b0 + 1, v);
                        }
                    }
                }
                // ---- dbn_sample.m:66-93 / :138-162: one transition step producing column c
                if (c < T) {
                    int nbin[ND];
#pragma unroll
                    for (int k = 0; k < ND; k++) {
                        nbin[k] = 0;
                        if (k >= P.nd) continue;
                        const uint32_t *t = tptr[k];
                        if (per_step) {
                            uint32_t col = basecol[k];
#pragma unroll
                            for (int q = 0; q < ND; q++) col += P.d_stride_cur[k][q] * (uint32_t)cur[q];
#pragma unroll
                            for (int q = 0; q < k; q++) col += P.d_stride_new[k][q] * (uint32_t)nbin[q];
                            t = P.thr + P.d_off[k] + (size_t)col * (uint32_t)(P.d_r[k] - 1);
                        }
                        nbin[k] = draw_bin(t, P.d_r[k], split_draw(th[k], tl[k], j));
                    }
                    // map back (:82/:149) and event rows in ascending variable id (:84-91/:151-161)
#pragma unroll
                    for (int e = 0; e < ND; e++) {
                        if (e >= P.nd) continue;
                        const int k = P.d_emit[e];
                        const int nbk = pick<ND>(nbin, k);
                        if (nbk != pick<ND>(cur, k)) {
                            const int nb = P.d_nb[k], zero = P.d_zero[k];
                            float v = (float)(nbk + 1);
                            if (!no_dedisc && nb != 0) {
                                if (zero == nbk + 1) v = 0.f;
                                else v = (float)dedisc_f64(P.bnd, P.d_boff[k], nbk,
                                                           word_of(rng.block(EMGPU_SEC_DEDISC_TRANS, P.d_ivar[k], g4), w));
                            }
                            put<ND>(cur, k, nbk);
                            put<ND>(cval, k, v);
                            if (want_events) emit(c, (int)P.d_ivar[k] + 1, nbk + 1, v);
                        }
                    }
                }
            }
            if (c < T) { // events2samples.m:15-26 column c
#pragma unroll
                for (int k = 0; k < ND; k++) {
                    pb[k][hb] |= (uint32_t)(cur[k] + 1) << (8 * w);
                    pv[k][hb][w] = cval[k];
                }
            }
        }
#pragma unroll
        for (int hb = 0; hb < 2; hb++) {
            const int g = 2 * g8 + hb;
            if (g >= G4) continue;
#pragma unroll
            for (int k = 0; k < ND; k++) {
                if (k >= P.nd) continue;
                const size_t o = ((size_t)g * P.nd + P.d_row[k]) * (size_t)A.ld + (size_t)i;
                if (A.dyn_bin) A.dyn_bin[o] = pb[k][hb];
                if (A.dyn_val) reinterpret_cast<float4 *>(A.dyn_val)[o] = make_float4(pv[k][hb][0], pv[k][hb][1], pv[k][hb][2], pv[k][hb][3]);
            }
        }
    }
    if (want_events) {
        if (!(A.flags & EMGPU_FLAG_NO_TERMINATOR)) emit(T, 0, 0, 0.f); // dbn_hierarchical_sample.m:15-19
        A.ev_count[i] = ecount;
        if (ecount > (uint32_t)A.event_cap) atomicOr(A.status, 2u);
    }
}

// ---------------------------------------------------------------------------------------------
// bn_sample.m:39-57 + @CorTerminalModel/sample.m:32-72
// ---------------------------------------------------------------------------------------------
// PS: per-sample presets and / or log-weights (a start grid in one launch, InitStartTerminal.m:57-90)
template <int NI, bool PS>
__global__ void __launch_bounds__(256) k_bn(const EmgpuPlan P, const EmgpuBnRun A) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n) return;
    const uint64_t gidx = A.indices ? A.indices[i] : A.first_index + (uint64_t)i;
    Rng rng{(uint32_t)gidx, (uint32_t)(gidx >> 32), 0u, (uint32_t)A.seed, (uint32_t)(A.seed >> 32)};
    const bool no_dedisc = (A.flags & EMGPU_FLAG_NO_DEDISC) != 0;
    int bin[NI];
    double val[NI];
#pragma unroll
    for (int p = 0; p < NI; p++) { bin[p] = 0; val[p] = 0.0; }
    int32_t attempts_used = -1;
    int sp[NI];
    if constexpr (PS) {
        const double lw = lane_presets<NI>(P, A.start ? A.start + (size_t)i * (size_t)P.ni : nullptr, A.log_weight ? A.logp : nullptr, A.lp_off, A.status, sp);
        if (A.log_weight) A.log_weight[i] = lw;
    }
    for (uint32_t attempt = 0; attempt < (uint32_t)A.max_attempts; attempt++) {
        rng.attempt = attempt;
        uint4 wc = make_uint4(0, 0, 0, 0);
        int wblk = -1;
#pragma unroll
        for (int p = 0; p < NI; p++) {
            if (p >= P.ni) continue;
            const int preset = PS ? sp[p] : (int)P.i_start[p];
            if (preset != 0) {
                bin[p] = preset - 1;
            } else {
                uint32_t col = 0;
#pragma unroll
                for (int q = 0; q < p; q++) col += P.i_stride[p][q] * (uint32_t)bin[q];
                const int r = P.i_r[p];
                const int var = P.i_var[p];
                if ((var >> 2) != wblk) { wblk = var >> 2; wc = rng.block(EMGPU_SEC_INIT, 0u, (uint32_t)wblk); }
                bin[p] = draw_bin(P.thr + P.i_off[p] + (size_t)col * (uint32_t)(r - 1), r, word_of(wc, var & 3));
            }
        }
        wblk = -1;
        bool good = true;
#pragma unroll
        for (int p = 0; p < NI; p++) {
            if (p >= P.ni) continue;
            double v = (double)(bin[p] + 1);
            if (!no_dedisc && P.i_nb[p] != 0) { // sample.m:37-42
                const int var = P.i_var[p];
                if ((var >> 2) != wblk) { wblk = var >> 2; wc = rng.block(EMGPU_SEC_GEOM_DEDISC, 0u, (uint32_t)wblk); }
                v = (P.i_zero[p] == bin[p] + 1) ? 0.0 : dedisc_f64(P.bnd, P.i_boff[p], bin[p], word_of(wc, var & 3));
            }
            val[p] = v;
            if (A.has_bounds) good = good && (v >= A.bounds[p][0]) && (v <= A.bounds[p][1]); // sample.m:45-53
        }
        if (good && A.pos_own_speed >= 0) { // sample.m:64-70
            const double s1 = pick<NI>(val, A.pos_own_speed), s2 = pick<NI>(val, A.pos_int_speed);
            good = (s1 <= A.max1) && (s1 >= A.min1) && (s2 <= A.max2) && (s2 >= A.min2);
        }
        if (good) { attempts_used = (int32_t)attempt + 1; break; }
    }
    if (attempts_used < 0) atomicOr(A.status, 1u);
    if (A.attempts) A.attempts[i] = attempts_used;
#pragma unroll
    for (int p = 0; p < NI; p++) {
        if (p >= P.ni) continue;
        if (A.out_bin) A.out_bin[(size_t)P.i_var[p] * A.ld + i] = (uint8_t)(bin[p] + 1);
        if (A.out_val) A.out_val[(size_t)P.i_var[p] * A.ld + i] = (float)val[p];
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
template <int NI, int ND, int NA>
static hipError_t launch_generic_t(const EmgpuPlan &P, const EmgpuRun &A, const EmgpuPresets *Q, hipStream_t s) {
    const int64_t blocks = (A.n + 255) / 256;
    hipLaunchKernelGGL((k_dbn_generic<NI, ND, NA>), dim3((unsigned)blocks), dim3(256), 0, s, P, A, Q);
    return hipGetLastError();
}

hipError_t launch_dbn_generic(const EmgpuPlan &P, const EmgpuRun &A, hipStream_t s, const char **name, const EmgpuPresets *Q) {
    if (A.n <= 0) return hipSuccess;
    if (P.ni <= 7 && P.nd <= 3 && P.nact <= 4) { *name = "k_dbn_generic<7,3,4>"; return launch_generic_t<7, 3, 4>(P, A, Q, s); }
    if (P.ni <= 7 && P.nd <= 3 && P.nact <= 7) { *name = "k_dbn_generic<7,3,7>"; return launch_generic_t<7, 3, 7>(P, A, Q, s); }
    if (P.ni <= 9 && P.nd <= 3 && P.nact <= 9) { *name = "k_dbn_generic<9,3,9>"; return launch_generic_t<9, 3, 9>(P, A, Q, s); }
    if (P.nact <= 4) { *name = "k_dbn_generic<16,4,4>"; return launch_generic_t<16, 4, 4>(P, A, Q, s); }
    *name = "k_dbn_generic<16,4,16>";
    return launch_generic_t<16, 4, 16>(P, A, Q, s);
}

hipError_t launch_bn(const EmgpuPlan &P, const EmgpuBnRun &A, hipStream_t s, const char **name) {
    if (A.n <= 0) return hipSuccess;
    const int64_t blocks = (A.n + 255) / 256;
    const bool ps = A.start != nullptr || A.log_weight != nullptr;
    if (P.ni <= 8) {
        *name = ps ? "k_bn<8>+start" : "k_bn<8>";
        if (ps) hipLaunchKernelGGL((k_bn<8, true>), dim3((unsigned)blocks), dim3(256), 0, s, P, A);
        else hipLaunchKernelGGL((k_bn<8, false>), dim3((unsigned)blocks), dim3(256), 0, s, P, A);
    } else {
        *name = ps ? "k_bn<16>+start" : "k_bn<16>";
        if (ps) hipLaunchKernelGGL((k_bn<16, true>), dim3((unsigned)blocks), dim3(256), 0, s, P, A);
        else hipLaunchKernelGGL((k_bn<16, false>), dim3((unsigned)blocks), dim3(256), 0, s, P, A);
    }
    return hipGetLastError();
}

} // namespace emgpu
