// emgpu_kernels_step2b.hip -- the instances of k_dbn_step2 (emgpu_kernels_step2.h) built for the parent masks and column widths of the
// shipped 3-variable model families; a translation unit of its own so that they compile beside the 4-variable ones.
#include "emgpu_kernels_step2.h"

namespace emgpu {

bool launch_masked3(const EmgpuPlan &P, const EmgpuRun &A, const Step2Args &F, hipStream_t s, uint32_t cur, uint32_t nw, const char **tag) {
    const dim3 g((unsigned)((A.n + 255) / 256)), b(256);
    const int wmode = 0;   // the 3-variable families run the per-variable width instance when no width-specific one fits
#define EMGPU_S2_CASE(NI_, ND_, W_, C_, N_, TAG_)                                                                  \
    if (P.ni <= NI_ && P.nd == ND_ && (W_ == 0 || wmode == W_) && cur == C_ && nw == N_) {                         \
        EMGPU_S2_LAUNCH(NI_, ND_, W_, true, C_, N_, false);                                                        \
        *tag = TAG_;                                                                                               \
        return true;                                                                                               \
    }
    // the 3-variable families with the widths of the shipped files as compile-time facts (dense output only; WMODE 16 + mask of the
    // 4-word variables): a width decided at run time is a wave-uniform branch per draw with both forms of the draw behind it
    uint32_t wm = 0u;
    for (int k = 0; k < P.nd; k++) wm |= (P.d_pw[k] == 4 ? 1u : 0u) << k;
#define EMGPU_S2_CASE_W(NI_, ND_, WM_, C_, N_, TAG_)                                                               \
    if (A.ev_count == nullptr && P.ni <= NI_ && P.nd == ND_ && wm == WM_ && cur == C_ && nw == N_) {               \
        hipLaunchKernelGGL((k_dbn_step2<NI_, ND_, 16 + WM_, true, C_, N_, false, false>), g, b, step2_extra_lds(), s, P, A, F);    \
        *tag = TAG_;                                                                                               \
        return true;                                                                                               \
    }
    EMGPU_S2_CASE_W(7, 3, 4, 0x0421u, 0x0310u, "[chain,w884]")      // glider_v1
    EMGPU_S2_CASE_W(7, 3, 0, 0x0421u, 0x0310u, "[chain,w888]")      // paraglider_v1
    EMGPU_S2_CASE_W(7, 3, 7, 0x0421u, 0x0210u, "[2<-1,w444]")       // littoral_uncor_v1
    EMGPU_S2_CASE_W(7, 3, 4, 0x0421u, 0x0210u, "[2<-1,w884]")       // paramotor_v1
    EMGPU_S2_CASE_W(7, 3, 5, 0x0421u, 0x0210u, "[2<-1,w484]")       // skydiving_v1
    EMGPU_S2_CASE_W(7, 3, 0, 0x0421u, 0x0110u, "[1<-0,2<-0,w888]")  // fai1_v1
    EMGPU_S2_CASE_W(7, 3, 2, 0x0421u, 0x0110u, "[1<-0,2<-0,w848]")  // fai5_v1
    EMGPU_S2_CASE_W(7, 3, 6, 0x0421u, 0x0300u, "[2<-0,1,w844]")     // uncor_1200code_v1
    EMGPU_S2_CASE_W(7, 3, 5, 0x0577u, 0x0000u, "[per-step,w484]")   // uncor_1200code_v2p1 under EMGPU_TRANSITION_PER_STEP
    EMGPU_S2_CASE_W(7, 3, 0, 0x0577u, 0x0000u, "[per-step,w888]")   // the v1.2 and allcode families under EMGPU_TRANSITION_PER_STEP
#undef EMGPU_S2_CASE_W
    EMGPU_S2_CASE(7, 3, 0, 0x0421u, 0x0310u, "[chain]")        // glider_v1, paraglider_v1
    EMGPU_S2_CASE(7, 3, 0, 0x0421u, 0x0210u, "[2<-1]")         // littoral_uncor_v1, paramotor_v1, skydiving_v1
    EMGPU_S2_CASE(7, 3, 0, 0x0421u, 0x0110u, "[1<-0,2<-0]")    // fai1_v1, fai5_v1
    EMGPU_S2_CASE(7, 3, 0, 0x0421u, 0x0300u, "[2<-0,1]")       // uncor_1200code_v1
    EMGPU_S2_CASE(7, 3, 0, 0x0577u, 0x0000u, "[per-step]")     // EMGPU_TRANSITION_PER_STEP on the conventional uncorrelated models
#undef EMGPU_S2_CASE
    return false;
}

#ifdef EMGPU_DEBUG_COUNTERS
extern "C" int emgpu_debug_counters_step2b(unsigned long long *out, int reset) {   // this translation unit's copy of g_dbg
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(g_dbg)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), z, sizeof z); }
    return 0;
}
#endif

} // namespace emgpu
