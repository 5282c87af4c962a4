// emgpu_kernels_step2.hip -- eligibility, dispatch and the 4-variable instances of k_dbn_step2 (emgpu_kernels_step2.h)
#include "emgpu_kernels_step2.h"

namespace emgpu {

void step_parent_masks(const EmgpuPlan &P, uint32_t *cur_mask, uint32_t *new_mask) {
    uint32_t c = 0u, n = 0u;
    for (int k = 0; k < P.nd; k++)
        for (int q = 0; q < P.nd; q++) {
            if (P.d_stride_cur[k][q] != 0u) c |= 1u << (4 * k + q);
            if (P.d_stride_new[k][q] != 0u) n |= 1u << (4 * k + q);
        }
    *cur_mask = c; *new_mask = n;
}

bool step2_eligible(const EmgpuPlan &P, const EmgpuRun &A) {
    if (A.indices != nullptr) return false; // an index list: k_uncor_fast_idx / _ev for the fast-branch models, else the generic kernel
    static const bool off = getenv("EMGPU_DEBUG_NO_STEP2") != nullptr;
    if (off) return false;
    if (P.nd < 1 || P.nd > 4) return false;
    // (a fast-branch model -- frozen columns, FRZ -- that k_uncor_fast did not take: four dynamic variables, or one or two: balloon_v1)
    // plain dbn_sample.m (no resample rows, the value of a row is its bin) returns a list and nothing else: the event instances serve it
    // with the resample streams switched off (launch_dbn_step2) and the rows' values taken from their bins
    const bool plain = (A.flags & (EMGPU_FLAG_NO_RESAMPLE | EMGPU_FLAG_NO_DEDISC)) != 0;
    if (plain && !(A.ev_count != nullptr && A.dyn_bin == nullptr && A.dyn_val == nullptr)) return false;
    if (A.flags & EMGPU_FLAG_NO_RESAMPLE) {
        EmgpuPlan Q = P;
        Q.nact = 0;
        EmgpuRun B = A;
        B.flags &= ~EMGPU_FLAG_NO_RESAMPLE;
        return step2_eligible(Q, B);
    }
    if ((A.ev_count != nullptr || A.events != nullptr) && step2_rows_by_wave(P, A)) {
        // a list asked for alone: 16 - ND streams of the instance (12 or 13 rated variables)
        const bool inst4 = !(P.depend || A.per_step) || !(P.ni <= 9 && P.nd <= 3);
        if (P.nact > (inst4 ? 12 : 13)) return false;
    } else if (A.ev_count != nullptr || A.events != nullptr) {
        if (!ev_plan_ok(P, A)) return false;
        // the event streams of a block are 8 - ND resample + ND transition streams of the INSTANCE that runs the model (ND = 4 for
        // the frozen instances and the 16-variable shape, else 3), not of the model: a model with fewer dynamic variables than its
        // instance may carry more rates than the instance has streams for
        const bool inst4 = !(P.depend || A.per_step) || !(P.ni <= 9 && P.nd <= 3);
        if (P.nact > (inst4 ? 4 : 5)) return false;
    }
    for (int k = 0; k < P.nd; k++) {
        if (P.d_nb[k] == 0 || P.d_nb[k] > 16 || P.d_pw[k] == 0) return false;
        for (int q = 0; q < P.nd; q++)
            if ((uint64_t)P.d_stride_cur[k][q] * 16u >= (1u << 24) || (uint64_t)P.d_stride_new[k][q] * 16u >= (1u << 24))
                return false; // 24-bit multiplies of the strides in bytes
        for (int a = 0; a < P.nact; a++)
            if (P.a_dyn[a] == k && P.a_R[a] >= 0xFFFF0000u) return false; // rate ~ 1 (R_h + 1 must fit 16 bits): older kernels
    }
    return true;
}

template <int NI, int ND>
static hipError_t launch_t(const EmgpuPlan &P, const EmgpuRun &A, const Step2Args &F, hipStream_t s, int wmode, bool reg) {
    const dim3 g((unsigned)((A.n + 255) / 256)), b(256);
    constexpr uint32_t C = ND == 4 ? kCurAll4 : kCurAll3, N = ND == 4 ? kNewAll4 : kNewAll3;
    if (A.ev_count != nullptr) {   // event lists: the per-variable-width instances only (half the instances for the rarer output)
        if (step2_rows_by_wave(P, A)) {
            if (reg) hipLaunchKernelGGL((k_dbn_step2<NI, ND, 0, true, C, N, false, 2>), g, b, step2_extra_lds(), s, P, A, F);
            else hipLaunchKernelGGL((k_dbn_step2<NI, ND, 0, false, C, N, false, 2>), g, b, step2_extra_lds(), s, P, A, F);
        } else if (reg) hipLaunchKernelGGL((k_dbn_step2<NI, ND, 0, true, C, N, false, 1>), g, b, step2_extra_lds(), s, P, A, F);
        else hipLaunchKernelGGL((k_dbn_step2<NI, ND, 0, false, C, N, false, 1>), g, b, step2_extra_lds(), s, P, A, F);
        return hipGetLastError();
    }
    if (reg && wmode == 4) hipLaunchKernelGGL((k_dbn_step2<NI, ND, 4, true, C, N>), g, b, step2_extra_lds(), s, P, A, F);
    else if (reg && wmode == 8) hipLaunchKernelGGL((k_dbn_step2<NI, ND, 8, true, C, N>), g, b, step2_extra_lds(), s, P, A, F);
    else if (reg) hipLaunchKernelGGL((k_dbn_step2<NI, ND, 0, true, C, N>), g, b, step2_extra_lds(), s, P, A, F);
    else hipLaunchKernelGGL((k_dbn_step2<NI, ND, 0, false, C, N>), g, b, step2_extra_lds(), s, P, A, F);
    return hipGetLastError();
}

// Instances built for the parent masks of the shipped model families (regular models only).  Returns false when none fits.
static bool launch_masked(const EmgpuPlan &P, const EmgpuRun &A, const Step2Args &F, hipStream_t s, int wmode, uint32_t cur, uint32_t nw, const char **tag) {
    const dim3 g((unsigned)((A.n + 255) / 256)), b(256);
    if (A.ev_count == nullptr && (A.dyn_bin == nullptr || A.dyn_val == nullptr)) return false;   // these instances store both dense outputs unconditionally
#define EMGPU_S2_CASE(NI_, ND_, W_, C_, N_, TAG_)                                                                  \
    if (P.ni <= NI_ && P.nd == ND_ && (W_ == 0 || wmode == W_) && cur == C_ && nw == N_) {                         \
        EMGPU_S2_LAUNCH(NI_, ND_, W_, true, C_, N_, false);                                                        \
        *tag = TAG_;                                                                                               \
        return true;                                                                                               \
    }
    EMGPU_S2_CASE(16, 4, 4, 0x8421u, 0x2100u, "[cor]")         // cor_v1: two independent aircraft, turn rate after vertical rate
    EMGPU_S2_CASE(16, 4, 8, 0x8421u, 0x2100u, "[cor]")
    if (P.nd == 3) return launch_masked3(P, A, F, s, cur, nw, tag);
#undef EMGPU_S2_CASE
    return false;
}

static hipError_t launch_dbn_step2_inner(const EmgpuPlan &P, const EmgpuRun &A, hipStream_t s, const char **name);
hipError_t launch_dbn_step2(const EmgpuPlan &P0, const EmgpuRun &A, hipStream_t s, const char **name) {
    EmgpuPlan P = P0;
    if (A.flags & EMGPU_FLAG_NO_RESAMPLE) P.nact = 0;   // no variable has a rate: no resample stream, no resample pass (the instances without "reg")
    const hipError_t e = launch_dbn_step2_inner(P, A, s, name);
    if (A.ev_count != nullptr) {   // the same kernel with the event list written as well
        static thread_local char evname[96];
        snprintf(evname, sizeof evname, "%s%s+events", *name, step2_rows_by_wave(P, A) ? "+rows-by-wave" : "");
        *name = evname;
    }
    return e;
}
static hipError_t launch_dbn_step2_inner(const EmgpuPlan &P, const EmgpuRun &A, hipStream_t s, const char **name) {
    if (A.n <= 0) return hipSuccess;
    Step2Args F{};
    const bool frozen = !(P.depend || A.per_step);
    bool all_res = true;
    int wmode = P.d_pw[0];
    for (int k = 0; k < P.nd; k++) {
        F.slot[k] = P.d_row[k];
        for (int a = 0; a < P.nact; a++)
            if (P.a_dyn[a] == k) F.Rk[k] = P.a_R[a];
        F.RR1[k] = ((F.Rk[k] >> 16) + 1u) * 0x00010001u;
        all_res = all_res && F.Rk[k] != 0u;
        if (P.d_pw[k] != wmode) wmode = 0;
    }
    // (Staging the tables in LDS was measured and dropped: random 16-byte gathers from LDS pay bank
    // conflicts and the extra LDS costs a workgroup of occupancy -- cor_v1 36.9 ms staged, 28.9 ms through L1/L2.)
    static const char *names[3][4] = {
        {"k_dbn_step2<7,3,w4,reg>", "k_dbn_step2<7,3,w8,reg>", "k_dbn_step2<7,3,reg>", "k_dbn_step2<7,3>"},
        {"k_dbn_step2<9,3,w4,reg>", "k_dbn_step2<9,3,w8,reg>", "k_dbn_step2<9,3,reg>", "k_dbn_step2<9,3>"},
        {"k_dbn_step2<16,4,w4,reg>", "k_dbn_step2<16,4,w8,reg>", "k_dbn_step2<16,4,reg>", "k_dbn_step2<16,4>"}};
    const int shape = (P.ni <= 7 && P.nd <= 3) ? 0 : ((P.ni <= 9 && P.nd <= 3) ? 1 : 2);
    const bool reg = all_res && P.nd == (shape == 2 ? 4 : 3);
    if (frozen) {   // fast branch: four dynamic variables, or fewer than three
        const dim3 g((unsigned)((A.n + 255) / 256)), b(256);
        uint32_t cur, nw;
        step_parent_masks(P, &cur, &nw);
        if (nw != 0u) return hipErrorNotSupported;   // (cannot be: is_dynvar_depend would be set)
        if (reg && wmode == 4 && P.ni <= 16 && cur == 0x8421u && (A.ev_count != nullptr || (A.dyn_bin != nullptr && A.dyn_val != nullptr))) {
            *name = "k_dbn_step2<16,4,w4,reg>[frozen]";   // littoral_cor_v1: every variable's only dynamic parent is its own current bin
            EMGPU_S2_LAUNCH(16, 4, 4, true, 0x8421u, 0u, true);
        } else {
            *name = "k_dbn_step2<16,4>[frozen]";
            EMGPU_S2_LAUNCH(16, 4, 0, false, kCurAll4, 0u, true);
        }
        return hipGetLastError();
    }
    static const bool no_masks = getenv("EMGPU_DEBUG_NO_STEP2_MASKS") != nullptr;
    if (reg && !no_masks) {
        uint32_t cur, nw;
        step_parent_masks(P, &cur, &nw);
        static thread_local char buf[64];
        const char *tag = "";
        const int w = shape == 2 ? wmode : 0;   // the 3-variable families run the per-variable width instance
        if ((shape == 2 ? (wmode == 4 || wmode == 8) : true) && launch_masked(P, A, F, s, w, cur, nw, &tag)) {
            snprintf(buf, sizeof buf, "%s%s", names[shape][w == 4 ? 0 : (w == 8 ? 1 : 2)], tag);
            *name = buf;
            return hipGetLastError();
        }
    }
    *name = names[shape][reg ? (wmode == 4 ? 0 : (wmode == 8 ? 1 : 2)) : 3];
    if (shape == 0) return launch_t<7, 3>(P, A, F, s, wmode, reg);
    if (shape == 1) return launch_t<9, 3>(P, A, F, s, wmode, reg);
    return launch_t<16, 4>(P, A, F, s, wmode, reg);
}

#ifdef EMGPU_DEBUG_COUNTERS
extern "C" int emgpu_debug_counters_step2(unsigned long long *out, int reset) {   // this translation unit's copy of g_dbg
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(g_dbg)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), z, sizeof z); }
    return 0;
}
#endif

} // namespace emgpu
