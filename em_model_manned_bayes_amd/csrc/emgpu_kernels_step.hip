// emgpu_kernels_step.hip -- k_dbn_step: the true per-timestep DBN (dbn_sample.m:65-93, the
// "dependent" branch, also EMGPU_TRANSITION_PER_STEP on any model) with dense trace output.
// Same 8-second-block structure as k_uncor_fast (emgpu_kernels_fast.hip): Philox blocks per
// variable, flag streams, wave-cooperative dediscretize, time-blocked SoA stores.  Differences:
// the parent configuration of every dynamic variable is rebuilt each second from the current
// state (asub2ind.m:13-14 as strides), so its quantile thresholds are gathered from the table
// (L1/L2-resident: 29 KB for cor_v1, 246 KB for uncor_1200code_v2p1) instead of living in
// registers, and the draws are always full 32-bit (primary + secondary halfword blocks): with
// coupled variables a tie would force the whole block to be redone far too often.
// Bound: VALU + L1/L2 gather latency; HBM writes are 4 880 B (cor) / 3 635 B (uncor) per trajectory.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "emgpu_coop.h"
#include "emgpu_device.h"
#include "emgpu_launch.h"

namespace emgpu {

struct StepArgs {
    uint32_t Rk[EMGPU_MAX_ND];   // resample hit threshold of dynamic variable k (0 = rate 0)
    uint32_t slot[EMGPU_MAX_ND]; // output row of dynamic variable k
};

// LDS_T: the dynamic variables' threshold tables are staged in (dynamic) LDS by the workgroup
// (cor_v1: 18 KB compacted); otherwise they are gathered from global memory (L1/L2).
// CT: the tables are the compacted ones (EmgpuPlan::cthr: distinct thresholds + bin map per column).
template <int NI, int ND, int RM1, bool LDS_T, bool CT>
__global__ void __launch_bounds__(256, 2) k_dbn_step(const EmgpuPlan P, const EmgpuRun A, const StepArgs F) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_thr[];
    __shared__ CoopLds<ND> s_wave[4];
    __shared__ double s_bnd[ND][16];
    const int tid = threadIdx.x, lane = tid & 63;
    CoopLds<ND> &W = s_wave[tid >> 6];
    const int64_t i = (int64_t)blockIdx.x * 256 + tid;
    const bool valid = i < A.n; // lanes past the end stay alive: they serve as workers for their wave
    const uint64_t gidx = A.first_index + (uint64_t)i;
    Rng rng{(uint32_t)gidx, (uint32_t)(gidx >> 32), 0u, (uint32_t)A.seed, (uint32_t)(A.seed >> 32)};
    const int T = A.T;
    if (tid < 16 * ND) {
        const int k = tid >> 4, q = tid & 15;
        s_bnd[k][q] = (k < P.nd && q < (int)P.d_nb[k]) ? P.bnd[P.d_boff[k] + q] : 0.0;
    }

    int bin[NI];
    double val[NI];
#pragma unroll
    for (int p = 0; p < NI; p++) { bin[p] = 0; val[p] = 0.0; }
    const int32_t attempts_used = init_network<NI>(P, A, rng, bin, val);
    if (valid) {
        if (attempts_used < 0) atomicOr(A.status, 1u);
        if (A.attempts) A.attempts[i] = attempts_used;
#pragma unroll
        for (int p = 0; p < NI; p++) {
            if (p < P.ni) {
                if (A.init_bin) A.init_bin[(size_t)P.i_var[p] * A.ld + i] = (uint8_t)(bin[p] + 1);
                if (A.init_val) A.init_val[(size_t)P.i_var[p] * A.ld + i] = (float)val[p];
            }
        }
    }
    W.attempt[lane] = rng.attempt;
    const uint32_t *__restrict__ gtab = CT ? P.cthr : P.thr + P.d_off[0];
    const uint32_t ntab = CT ? P.cthr_total : P.thr_total - P.d_off[0];
    if (LDS_T)
        for (uint32_t q = (uint32_t)tid; q < ntab; q += 256u) s_thr[q] = gtab[q];
    __syncthreads();
    const uint32_t *__restrict__ tbase = LDS_T ? (const uint32_t *)s_thr : gtab;

    uint32_t cur1[ND], basecol[ND], ivs[ND];
    float cval[ND];
#pragma unroll
    for (int k = 0; k < ND; k++) {
        cur1[k] = 1u; cval[k] = 0.f; basecol[k] = 0u; ivs[k] = P.d_ivar[k];
        if (k < P.nd) {
            cur1[k] = (uint32_t)pick<NI>(bin, P.d_ipos[k]) + 1u;
            cval[k] = (float)pick<NI>(val, P.d_ipos[k]);
            uint32_t b = 0;
#pragma unroll
            for (int p = 0; p < NI; p++) b += P.d_stride_static[k][p] * (uint32_t)bin[p];
            basecol[k] = b;
        }
    }

    const int G4 = (T + 3) >> 2, G8 = (T + 7) >> 3;
    for (int g8 = 0; g8 < G8; g8++) {
        uint4 th[ND], tl[ND], rh[ND], rl[ND];
#pragma unroll
        for (int k = 0; k < ND; k++) {
            th[k] = tl[k] = rh[k] = rl[k] = make_uint4(0, 0, 0, 0);
            if (k < P.nd) {
                th[k] = rng.block(EMGPU_SEC_TRANS, P.d_tvar[k], (uint32_t)g8);
                tl[k] = rng.block(EMGPU_SEC_TRANS_LO, P.d_tvar[k], (uint32_t)g8);
                if (F.Rk[k] != 0u) {
                    rh[k] = rng.block(EMGPU_SEC_RES, P.d_ivar[k], (uint32_t)g8);
                    rl[k] = rng.block(EMGPU_SEC_RES_LO, P.d_ivar[k], (uint32_t)g8);
                }
            }
        }
        uint32_t pbA[ND], pbB[ND], hit8[ND], chg8[ND], zer8[ND];
#pragma unroll
        for (int k = 0; k < ND; k++) pbA[k] = pbB[k] = hit8[k] = chg8[k] = zer8[k] = 0u;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int c = 8 * g8 + j; // absolute event time == column produced
            uint32_t h[ND], ch[ND], z[ND];
#pragma unroll
            for (int k = 0; k < ND; k++) h[k] = ch[k] = z[k] = 0u;
            if (c >= 1 && c < T) {
                uint32_t nb1[ND];
#pragma unroll
                for (int k = 0; k < ND; k++) {
                    nb1[k] = 1u;
                    if (k >= P.nd) continue;
                    h[k] = (clamp32(split_draw(rh[k], rl[k], j)) < F.Rk[k]) ? 1u : 0u;      // resample_events.m:24
                    uint32_t col = basecol[k];                                               // asub2ind.m:13-14
#pragma unroll
                    for (int q = 0; q < ND; q++) col += P.d_stride_cur[k][q] * (cur1[q] - 1u);
#pragma unroll
                    for (int q = 0; q < k; q++) col += P.d_stride_new[k][q] * (nb1[q] - 1u);
                    const int rm1 = CT ? (int)P.d_meff[k] : (int)P.d_r[k] - 1;               // thresholds per column
                    const uint32_t *__restrict__ t = tbase + (CT ? P.d_coff[k] + (size_t)col * (uint32_t)(rm1 + 1)
                                                                 : (P.d_off[k] - P.d_off[0]) + (size_t)col * (uint32_t)rm1);
                    const uint32_t x = clamp32(split_draw(th[k], tl[k], j));
                    uint32_t borrows = 0u;
#pragma unroll
                    for (int q = 0; q < RM1; q++)
                        if (q < rm1) borrows += (x < t[q]) ? 1u : 0u;                        // select_random.m:19-20
                    const uint32_t fired = (uint32_t)rm1 - borrows;
                    nb1[k] = CT ? ((t[rm1] >> (fired << 2)) & 15u) : fired + 1u;             // dbn_sample.m:77
                }
#pragma unroll
                for (int k = 0; k < ND; k++) {
                    if (k >= P.nd) continue;
                    ch[k] = (nb1[k] != cur1[k]) ? 1u : 0u;
                    cur1[k] = nb1[k];                                                        // map back, dbn_sample.m:82
                    z[k] = (nb1[k] == (uint32_t)P.d_zero[k]) ? 1u : 0u;                      // dediscretize.m:24-25
                }
            }
#pragma unroll
            for (int k = 0; k < ND; k++) {
                hit8[k] |= h[k] << j; chg8[k] |= ch[k] << j; zer8[k] |= z[k] << j;
                const uint32_t b = (c < T) ? (cur1[k] << (8 * (j & 3))) : 0u;
                if (j < 4) pbA[k] |= b; else pbB[k] |= b;
            }
        }
        uint32_t need = 0u, kind = 0u, need8[ND], zero8[ND];
#pragma unroll
        for (int k = 0; k < ND; k++) {
            need8[k] = (valid && k < P.nd) ? ((hit8[k] | chg8[k]) & ~zer8[k]) : 0u;   // a dediscretize draw is due
            zero8[k] = chg8[k] & zer8[k];                                              // changed into the zero bin
            need |= need8[k] << (8 * k);
            kind |= chg8[k] << (8 * k);
        }
        coop_dedisc<ND>(W, lane, gidx, rng, g8, need, kind, pbA, pbB, ivs, s_bnd);
#pragma unroll
        for (int k = 0; k < ND; k++)
            if (k < P.nd)
                coop_fill_store<ND>(W, lane, k, g8, T, G4, valid, need8[k], zero8[k], cval[k], pbA[k], pbB[k],
                                    (uint32_t)P.nd, F.slot[k], i, A.ld, A.dyn_bin, A.dyn_val);
        wave_sync();
    }
}

bool step_eligible(const EmgpuPlan &P, const EmgpuRun &A) {
    if (A.indices != nullptr) return false; // an index list goes through the generic kernel
    if (P.nd < 1 || P.nd > 4) return false;
    if (!(P.depend || A.per_step)) return false;
    if (A.ev_count != nullptr || A.events != nullptr) return false;
    if (A.flags & (EMGPU_FLAG_NO_RESAMPLE | EMGPU_FLAG_NO_DEDISC)) return false;
    for (int k = 0; k < P.nd; k++) {
        if (P.d_nb[k] == 0 || P.d_nb[k] > 16 || P.d_r[k] > 9) return false;
        for (int a = 0; a < P.nact; a++)
            if (P.a_dyn[a] == k && P.a_R[a] == 0xFFFFFFFFu) return false;
    }
    return true;
}

static bool step_compact(const EmgpuPlan &P) {
    static const bool off = getenv("EMGPU_DEBUG_STEP_NO_COMPACT") != nullptr;
    if (off) return false;
    for (int k = 0; k < P.nd; k++)
        if (P.d_meff[k] == 0) return false;
    return true;
}
static size_t step_table_bytes(const EmgpuPlan &P) {
    return (size_t)(step_compact(P) ? P.cthr_total : P.thr_total - P.d_off[0]) * sizeof(uint32_t);
}

template <int NI, int ND, int RM1>
static hipError_t launch_t(const EmgpuPlan &P, const EmgpuRun &A, const StepArgs &F, hipStream_t s, bool lds) {
    const int64_t blocks = (A.n + 255) / 256;
    const bool ct = step_compact(P);
    const size_t bytes = lds ? step_table_bytes(P) : 0;
    if (lds && ct) hipLaunchKernelGGL((k_dbn_step<NI, ND, RM1, true, true>), dim3((unsigned)blocks), dim3(256), bytes, s, P, A, F);
    else if (lds) hipLaunchKernelGGL((k_dbn_step<NI, ND, RM1, true, false>), dim3((unsigned)blocks), dim3(256), bytes, s, P, A, F);
    else if (ct) hipLaunchKernelGGL((k_dbn_step<NI, ND, RM1, false, true>), dim3((unsigned)blocks), dim3(256), 0, s, P, A, F);
    else hipLaunchKernelGGL((k_dbn_step<NI, ND, RM1, false, false>), dim3((unsigned)blocks), dim3(256), 0, s, P, A, F);
    return hipGetLastError();
}

hipError_t launch_dbn_step(const EmgpuPlan &P, const EmgpuRun &A, hipStream_t s, const char **name) {
    if (A.n <= 0) return hipSuccess;
    StepArgs F{};
    for (int k = 0; k < P.nd; k++) {
        F.slot[k] = P.d_row[k];
        for (int a = 0; a < P.nact; a++)
            if (P.a_dyn[a] == k) F.Rk[k] = P.a_R[a];
    }
    // stage the dynamic tables in LDS when two workgroups per CU still fit beside the cooperative area
    static const bool no_lds = getenv("EMGPU_DEBUG_STEP_NO_LDS") != nullptr;
    const bool lds = !no_lds && step_table_bytes(P) <= 32768;
    if (P.ni <= 7 && P.nd <= 3) { *name = lds ? "k_dbn_step<7,3,8,lds>" : "k_dbn_step<7,3,8>"; return launch_t<7, 3, 8>(P, A, F, s, lds); }
    if (P.ni <= 9 && P.nd <= 3) { *name = lds ? "k_dbn_step<9,3,8,lds>" : "k_dbn_step<9,3,8>"; return launch_t<9, 3, 8>(P, A, F, s, lds); }
    *name = lds ? "k_dbn_step<16,4,8,lds>" : "k_dbn_step<16,4,8>";
    return launch_t<16, 4, 8>(P, A, F, s, lds);
}

} // namespace emgpu
