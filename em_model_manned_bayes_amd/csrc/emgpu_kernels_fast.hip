// emgpu_kernels_fast.hip -- launchers of the dense forms of the fast kernel (emgpu_kernels_fast.h): k_uncor_fast (the benchmark),
// k_uncor_fast_idx, k_uncor_fast_mixed; a call that wants an event list goes on to emgpu_kernels_fast_ev.hip.
#include "emgpu_kernels_fast.h"

namespace emgpu {

hipError_t launch_uncor_fast_events(const EmgpuPlan &P, const EmgpuRun &A, const FastArgs &F, hipStream_t s, const char **name);   // emgpu_kernels_fast_ev.hip

bool fast_uncor_eligible(const EmgpuPlan &P, const EmgpuRun &A) {
    if (P.nd != 3 || P.depend || A.per_step) return false;
    if ((A.ev_count != nullptr || A.events != nullptr) && !ev_plan_ok(P, A) && !(ev_plan_wide_ok(P, A) && P.ni <= 9)) return false;
    // plain dbn_sample.m (no resample rows, values = bins) returns a list and nothing else: k_uncor_fast_evu serves exactly that
    if ((A.flags & (EMGPU_FLAG_NO_RESAMPLE | EMGPU_FLAG_NO_DEDISC)) && !(A.ev_count != nullptr && A.dyn_bin == nullptr && A.dyn_val == nullptr && ev_plan_wide_ok(P, A)))
        return false;
    for (int k = 0; k < 3; k++) {
        if (P.d_nb[k] == 0 || P.d_nb[k] > 16 || P.d_meff[k] == 0) return false;
        for (int a = 0; a < P.nact; a++)
            if (P.a_dyn[a] == k && P.a_R[a] >= 0xFFFF0000u) return false; // rate ~ 1 (R_h + 1 must fit 16 bits): generic path
    }
    return fast_shape_of(P) >= 0;
}

template <int NI, int M0, int M1, int M2>
static hipError_t launch_t(const EmgpuPlan &P, const EmgpuRun &A, const FastArgs &F, hipStream_t s) {
    const int64_t blocks = (A.n + (A.col0 & 255) + 255) / 256;
    // EMGPU_DEBUG_EXTRA_LDS: bytes of unused dynamic LDS per workgroup, to study occupancy sensitivity
    static const int extra_lds = getenv("EMGPU_DEBUG_EXTRA_LDS") ? atoi(getenv("EMGPU_DEBUG_EXTRA_LDS")) : 0;
    hipLaunchKernelGGL((k_uncor_fast<NI, M0, M1, M2>), dim3((unsigned)blocks), dim3(256), (size_t)extra_lds, s, P, A, F);
    return hipGetLastError();
}

int uncor_fast_shape(const EmgpuPlan &P) { return fast_shape_of(P); }

size_t plan_f_bytes() { return sizeof(PlanF); }
void plan_f_fill(const EmgpuPlan &P, void *host_buf) {
    PlanF *pf = static_cast<PlanF *>(host_buf);
    pf->P = P; pf->F = fast_args_of(P);
}

template <int NI, int M0, int M1, int M2>
static hipError_t launch_mixed_t(const MixedHead &H, unsigned blocks, hipStream_t s) {
    hipLaunchKernelGGL((k_uncor_fast_mixed<NI, M0, M1, M2>), dim3(blocks), dim3(256), 0, s, H);
    return hipGetLastError();
}

// A: the call's run with the outputs bound at column 0; block b = n[b] trajectories from global index first[b] on, written from
// column col[b] on, with the device-resident plan d_planf[b] (plan_f_fill); all of instance `shape`.
hipError_t launch_uncor_fast_mixed(const EmgpuRun &A, int nb, const void *const *d_planf, const uint64_t *first, const int64_t *n, const int64_t *col,
                                   int shape, hipStream_t s, const char **name) {
    if (nb < 1 || nb > EMGPU_MAX_MIXED) return hipErrorInvalidValue;
    MixedHead H{};
    H.A = A;
    H.nb = nb;
    uint64_t wg = 0;
    for (int b = 0; b < nb; b++) {
        H.blk[b] = MixedBlock{static_cast<const PlanF *>(d_planf[b]), first[b], n[b], col[b]};
        H.wg_begin[b] = (uint32_t)wg;
        wg += (uint64_t)((n[b] + ((A.col0 + col[b]) & 255) + 255) / 256);   // lined up with the trace's columns (k_uncor_fast_mixed)
    }
    for (int b = nb; b <= EMGPU_MAX_MIXED; b++) H.wg_begin[b] = (uint32_t)wg;
    for (int b = nb; b < EMGPU_MAX_MIXED; b++) H.blk[b] = H.blk[nb - 1];
    if (wg == 0) return hipSuccess;
    if (wg > 0x7FFFFFFFull) return hipErrorInvalidValue;
    switch (shape) {
    case 0: *name = "k_uncor_fast_mixed<7,2,2,2>"; return launch_mixed_t<7, 2, 2, 2>(H, (unsigned)wg, s);
    case 1: *name = "k_uncor_fast_mixed<7,2,4,2>"; return launch_mixed_t<7, 2, 4, 2>(H, (unsigned)wg, s);
    case 2: *name = "k_uncor_fast_mixed<7,2,4,4>"; return launch_mixed_t<7, 2, 4, 4>(H, (unsigned)wg, s);
    case 3: *name = "k_uncor_fast_mixed<7,4,2,4>"; return launch_mixed_t<7, 4, 2, 4>(H, (unsigned)wg, s);
    case 4: *name = "k_uncor_fast_mixed<7,4,6,4>"; return launch_mixed_t<7, 4, 6, 4>(H, (unsigned)wg, s);
    case 5: *name = "k_uncor_fast_mixed<7,4,6,6>"; return launch_mixed_t<7, 4, 6, 6>(H, (unsigned)wg, s);
    case 6: *name = "k_uncor_fast_mixed<7,6,6,6>"; return launch_mixed_t<7, 6, 6, 6>(H, (unsigned)wg, s);
    case 7: *name = "k_uncor_fast_mixed<9,6,6,6>"; return launch_mixed_t<9, 6, 6, 6>(H, (unsigned)wg, s);
    default: *name = "none"; return hipErrorNotSupported;
    }
}

template <int NI, int M0, int M1, int M2>
static hipError_t launch_idx_t(const EmgpuPlan &P, const EmgpuRun &A, const FastArgs &F, hipStream_t s) {
    const int64_t blocks = (A.n + (A.col0 & 255) + 255) / 256;
    hipLaunchKernelGGL((k_uncor_fast_idx<NI, M0, M1, M2>), dim3((unsigned)blocks), dim3(256), 0, s, P, A, F);
    return hipGetLastError();
}

hipError_t launch_uncor_fast(const EmgpuPlan &P, const EmgpuRun &A, hipStream_t s, const char **name) {
    if (A.n <= 0) return hipSuccess;
    const FastArgs F = fast_args_of(P);
    if ((A.indices != nullptr || A.dyn_bin == nullptr || A.dyn_val == nullptr) && A.ev_count == nullptr) {   // an index list, or only one of the dense outputs
        switch (fast_shape_of(P)) {
        case 0: *name = "k_uncor_fast_idx<7,2,2,2>"; return launch_idx_t<7, 2, 2, 2>(P, A, F, s);
        case 1: *name = "k_uncor_fast_idx<7,2,4,2>"; return launch_idx_t<7, 2, 4, 2>(P, A, F, s);
        case 2: *name = "k_uncor_fast_idx<7,2,4,4>"; return launch_idx_t<7, 2, 4, 4>(P, A, F, s);
        case 3: *name = "k_uncor_fast_idx<7,4,2,4>"; return launch_idx_t<7, 4, 2, 4>(P, A, F, s);
        case 4: *name = "k_uncor_fast_idx<7,4,6,4>"; return launch_idx_t<7, 4, 6, 4>(P, A, F, s);
        case 5: *name = "k_uncor_fast_idx<7,4,6,6>"; return launch_idx_t<7, 4, 6, 6>(P, A, F, s);
        case 6: *name = "k_uncor_fast_idx<7,6,6,6>"; return launch_idx_t<7, 6, 6, 6>(P, A, F, s);
        case 7: *name = "k_uncor_fast_idx<9,6,6,6>"; return launch_idx_t<9, 6, 6, 6>(P, A, F, s);
        default: *name = "none"; return hipErrorNotSupported;
        }
    }
    if (A.ev_count != nullptr) return launch_uncor_fast_events(P, A, F, s, name);
    switch (fast_shape_of(P)) {
    case 0: *name = "k_uncor_fast<7,2,2,2>"; return launch_t<7, 2, 2, 2>(P, A, F, s);
    case 1: *name = "k_uncor_fast<7,2,4,2>"; return launch_t<7, 2, 4, 2>(P, A, F, s);
    case 2: *name = "k_uncor_fast<7,2,4,4>"; return launch_t<7, 2, 4, 4>(P, A, F, s);
    case 3: *name = "k_uncor_fast<7,4,2,4>"; return launch_t<7, 4, 2, 4>(P, A, F, s);
    case 4: *name = "k_uncor_fast<7,4,6,4>"; return launch_t<7, 4, 6, 4>(P, A, F, s);
    case 5: *name = "k_uncor_fast<7,4,6,6>"; return launch_t<7, 4, 6, 6>(P, A, F, s);
    case 6: *name = "k_uncor_fast<7,6,6,6>"; return launch_t<7, 6, 6, 6>(P, A, F, s);
    case 7: *name = "k_uncor_fast<9,6,6,6>"; return launch_t<9, 6, 6, 6>(P, A, F, s);
    default: *name = "none"; return hipErrorNotSupported;
    }
}

#ifdef EMGPU_DEBUG_COUNTERS
extern "C" int emgpu_debug_counters(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(g_dbg)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), z, sizeof z); }
    return 0;
}
#endif

} // namespace emgpu
