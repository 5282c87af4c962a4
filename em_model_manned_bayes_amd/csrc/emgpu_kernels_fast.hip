// emgpu_kernels_fast.hip -- the benchmarked kernel: uncorrelated DBN, REFERENCE_AUTO semantics on a
// "fast-branch" model (dbn_sample.m:95-166: parent configuration frozen at the initial state),
// compact dense trace output.  One lane = one trajectory, 3 dynamic variables.
//
// Per trajectory and 4-second block: 3 Philox calls for the transition draws (dbn_sample.m:133,144),
// 3 for the resample Bernoullis of the dynamic variables (resample_events.m:24; variables that are
// not dynamic cannot change the dense trace, SURVEY.md section 8d scope note), u32 threshold compares on
// register-resident quantile thresholds (select_random.m:17-20), rare dediscretize draws
// (dediscretize.m:39) and one 4-byte + one 16-byte store per variable (time-blocked SoA).
// Bound: HBM writes (3635 B / trajectory) co-limited by the integer multiplies of Philox4x32-10
// (DESIGN.md section 5); no MFMA: there is no contraction on this path.
#include <hip/hip_runtime.h>

#include "emgpu_device.h"
#include "emgpu_launch.h"

namespace emgpu {

struct FastArgs {
    uint32_t Rk[3];   // resample hit threshold of dynamic variable k (0 = rate 0)
    uint32_t slot[3]; // output row of dynamic variable k
};

template <int R>
__device__ __forceinline__ int draw_reg(const uint32_t (&th)[R - 1], uint32_t x) {
    const uint32_t xp = clamp32(x);
    int b = 0;
#pragma unroll
    for (int j = 0; j < R - 1; j++) b += (xp >= th[j]) ? 1 : 0;
    return b;
}

template <int R>
__device__ __forceinline__ void load_thr(uint32_t (&th)[R - 1], const uint32_t *__restrict__ p) {
#pragma unroll
    for (int j = 0; j < R - 1; j++) th[j] = p[j];
}

template <int NI, int R0, int R1, int R2>
__global__ void __launch_bounds__(256) k_uncor_fast(const EmgpuPlan P, const EmgpuRun A, const FastArgs F) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n) return;
    const uint64_t gidx = A.first_index + (uint64_t)i;
    Rng rng{(uint32_t)gidx, (uint32_t)(gidx >> 32), 0u, (uint32_t)A.seed, (uint32_t)(A.seed >> 32)};
    const int T = A.T;

    int bin[NI];
    double val[NI];
#pragma unroll
    for (int p = 0; p < NI; p++) { bin[p] = 0; val[p] = 0.0; }
    const int32_t attempts_used = init_network<NI>(P, A, rng, bin, val);
    if (attempts_used < 0) atomicOr(A.status, 1u);
    if (A.attempts) A.attempts[i] = attempts_used;
#pragma unroll
    for (int p = 0; p < NI; p++) {
        if (p >= P.ni) continue;
        if (A.init_bin) A.init_bin[(size_t)P.i_var[p] * A.n + i] = (uint8_t)(bin[p] + 1);
        if (A.init_val) A.init_val[(size_t)P.i_var[p] * A.n + i] = (float)val[p];
    }

    // frozen parent configuration -> one CPT column per dynamic variable (dbn_sample.m:110-135)
    int cur[3];
    float cval[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        cur[k] = pick<NI>(bin, P.d_ipos[k]);
        cval[k] = (float)pick<NI>(val, P.d_ipos[k]);
    }
    uint32_t th0[R0 - 1], th1[R1 - 1], th2[R2 - 1];
    {
        uint32_t col[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            uint32_t c = 0;
#pragma unroll
            for (int p = 0; p < NI; p++) c += P.d_stride_static[k][p] * (uint32_t)bin[p];
#pragma unroll
            for (int q = 0; q < 3; q++) c += P.d_stride_cur[k][q] * (uint32_t)cur[q];
            col[k] = c;
        }
        load_thr<R0>(th0, P.thr + P.d_off[0] + (size_t)col[0] * (R0 - 1));
        load_thr<R1>(th1, P.thr + P.d_off[1] + (size_t)col[1] * (R1 - 1));
        load_thr<R2>(th2, P.thr + P.d_off[2] + (size_t)col[2] * (R2 - 1));
    }

    const int G4 = (T + 3) >> 2;
    for (int g = 0; g < G4; g++) {
        uint4 tw[3], rw[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            tw[k] = rng.block(EMGPU_SEC_TRANS, P.d_tvar[k], (uint32_t)g);
            rw[k] = rng.block(EMGPU_SEC_RES, P.d_ivar[k], (uint32_t)g);
        }
        uint32_t pb[3] = {0u, 0u, 0u};
        float pv[3][4];
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const int c = 4 * g + w; // absolute event time == column produced
            if (c >= 1 && c < T) {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const bool hit = clamp32(word_of(rw[k], w)) < F.Rk[k];                 // resample_events.m:24
                    const int nb = k == 0 ? draw_reg<R0>(th0, word_of(tw[0], w))
                                 : k == 1 ? draw_reg<R1>(th1, word_of(tw[1], w))
                                          : draw_reg<R2>(th2, word_of(tw[2], w));         // dbn_sample.m:144
                    const bool changed = nb != cur[k];
                    cur[k] = nb;                                                           // map back, dbn_sample.m:149
                    const bool zero = (int)P.d_zero[k] == nb + 1;                          // dediscretize.m:24-25
                    if (changed && zero) cval[k] = 0.f;
                    if ((changed || hit) && !zero) {
                        // a transition event hides a resample event of the same second in the dense trace
                        const uint32_t sec = changed ? EMGPU_SEC_DEDISC_TRANS : EMGPU_SEC_DEDISC_RES;
                        const uint4 dw = rng.block(sec, P.d_ivar[k], (uint32_t)g);
                        cval[k] = (float)dedisc_f64(P.bnd, P.d_boff[k], nb, word_of(dw, w));  // dediscretize.m:39
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const bool live = c < T;
                pb[k] |= live ? ((uint32_t)(cur[k] + 1) << (8 * w)) : 0u;
                pv[k][w] = live ? cval[k] : 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const size_t o = ((size_t)g * 3 + F.slot[k]) * (size_t)A.n + (size_t)i;
            if (A.dyn_bin) A.dyn_bin[o] = pb[k];
            if (A.dyn_val) reinterpret_cast<float4 *>(A.dyn_val)[o] = make_float4(pv[k][0], pv[k][1], pv[k][2], pv[k][3]);
        }
    }
}

bool fast_uncor_eligible(const EmgpuPlan &P, const EmgpuRun &A) {
    if (P.nd != 3 || P.depend || A.per_step) return false;
    if (A.ev_count != nullptr || A.events != nullptr) return false;
    if (A.flags & (EMGPU_FLAG_NO_RESAMPLE | EMGPU_FLAG_NO_DEDISC)) return false;
    for (int k = 0; k < 3; k++)
        if (P.d_nb[k] == 0) return false;
    const int r0 = P.d_r[0], r1 = P.d_r[1], r2 = P.d_r[2];
    if (P.ni <= 7 && r0 == 5 && r1 == 7 && r2 == 7) return true;
    if (P.ni <= 7 && r0 == 5 && r1 == 9 && r2 == 7) return true;
    if (P.ni <= 9 && r0 == 7 && r1 == 7 && r2 == 5) return true;
    return false;
}

template <int NI, int R0, int R1, int R2>
static hipError_t launch_t(const EmgpuPlan &P, const EmgpuRun &A, const FastArgs &F, hipStream_t s) {
    const int64_t blocks = (A.n + 255) / 256;
    hipLaunchKernelGGL((k_uncor_fast<NI, R0, R1, R2>), dim3((unsigned)blocks), dim3(256), 0, s, P, A, F);
    return hipGetLastError();
}

hipError_t launch_uncor_fast(const EmgpuPlan &P, const EmgpuRun &A, hipStream_t s, const char **name) {
    if (A.n <= 0) return hipSuccess;
    FastArgs F{};
    for (int k = 0; k < 3; k++) {
        F.slot[k] = P.d_row[k];
        for (int a = 0; a < P.nact; a++)
            if (P.a_dyn[a] == k) F.Rk[k] = P.a_R[a];
    }
    const int r0 = P.d_r[0], r1 = P.d_r[1];
    if (r0 == 5 && r1 == 7) { *name = "k_uncor_fast<7,5,7,7>"; return launch_t<7, 5, 7, 7>(P, A, F, s); }
    if (r0 == 5 && r1 == 9) { *name = "k_uncor_fast<7,5,9,7>"; return launch_t<7, 5, 9, 7>(P, A, F, s); }
    *name = "k_uncor_fast<9,7,7,5>";
    return launch_t<9, 7, 7, 5>(P, A, F, s);
}

} // namespace emgpu
