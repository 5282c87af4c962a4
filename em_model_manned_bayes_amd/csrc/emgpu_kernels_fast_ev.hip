// emgpu_kernels_fast_ev.hip -- launchers of the event-list forms of the fast kernel (emgpu_kernels_fast.h, emgpu_events.h), a translation
// unit of their own so that they compile beside the dense forms:
//   k_uncor_fast_evu  the list alone (what UncorEncounterModel.sample / dbn_hierarchical_sample return): the rows of a block built 64 at
//                     a time by the wave ("ROWS BY THE WAVE"), any number of rated variables up to 16 - 3; _long: its form for lists of
//                     hundreds of rows per wave and block
//   k_uncor_fast_ev   the list AND the dense trace, at most five rated variables: result slots + a row loop per lane
//   k_uncor_fast_evw  the same for more rated variables (haa_v1: seven), on the widest instance
#include "emgpu_kernels_fast.h"

namespace emgpu {

template <int NI, int M0, int M1, int M2>
static hipError_t launch_ev_t(const EmgpuPlan &P, const EmgpuRun &A, const FastArgs &F, hipStream_t s) {
    const int64_t blocks = (A.n + (A.col0 & 255) + 255) / 256;
    hipLaunchKernelGGL((k_uncor_fast_ev<NI, M0, M1, M2>), dim3((unsigned)blocks), dim3(256), 0, s, P, A, F);
    return hipGetLastError();
}
template <int NI, int M0, int M1, int M2>
static hipError_t launch_evu_t(const EmgpuPlan &P, const EmgpuRun &A, const FastArgs &F, hipStream_t s) {
    const int64_t blocks = (A.n + (A.col0 & 255) + 255) / 256;
    hipLaunchKernelGGL((k_uncor_fast_evu<NI, M0, M1, M2>), dim3((unsigned)blocks), dim3(256), 0, s, P, A, F);
    return hipGetLastError();
}

hipError_t launch_uncor_fast_events(const EmgpuPlan &P, const EmgpuRun &A, const FastArgs &F, hipStream_t s, const char **name) {
    // EMGPU_DEBUG_EVENT_ROWS (tests, A/B runs): "lane" = the list alone also takes the per-lane row loops (k_uncor_fast_ev / _evw),
    // "wide" = every list takes k_uncor_fast_evw (its instance holds any fast-branch shape), "long" = every list alone takes k_uncor_fast_evu_long
    static const char *rows_env = getenv("EMGPU_DEBUG_EVENT_ROWS");
    const bool force_long = rows_env != nullptr && rows_env[0] == 'l' && rows_env[1] == 'o';
    const bool force_lane = rows_env != nullptr && rows_env[0] == 'l' && !force_long, force_wide = rows_env != nullptr && rows_env[0] == 'w' && rows_env[1] == 'i';
    const bool list_alone = A.dyn_bin == nullptr && A.dyn_val == nullptr;
    const bool plain = (A.flags & (EMGPU_FLAG_NO_RESAMPLE | EMGPU_FLAG_NO_DEDISC)) != 0;   // (only eligible as a list alone: fast_uncor_eligible)
    if (list_alone && ((!force_lane && !force_wide) || plain) && ev_plan_wide_ok(P, A)) {
        // rows expected per wave and 8-second block from the resample rates alone (transition rows come on top): several hundred of them
        // (haa_v1: 1.27 per second and lane -> 650) would take the short queue's 254 requests per round three or four rounds per block
        double rate = 0.0;
        if (!(A.flags & EMGPU_FLAG_NO_RESAMPLE))
            for (int a = 0; a < P.nact; a++) rate += (double)P.a_R[a] * (1.0 / 4294967296.0);
        if (rate * 512.0 > 300.0 || (force_long && P.ni <= 9)) {
            const int64_t blocks = (A.n + (A.col0 & 255) + 255) / 256;
            *name = "k_uncor_fast_evu_long<9,6,6,6>";
            hipLaunchKernelGGL((k_uncor_fast_evu_long<9, 6, 6, 6>), dim3((unsigned)blocks), dim3(256), 0, s, P, A, F);
            return hipGetLastError();
        }
        switch (fast_shape_of(P)) {
        case 0: *name = "k_uncor_fast_evu<7,2,2,2>"; return launch_evu_t<7, 2, 2, 2>(P, A, F, s);
        case 1: *name = "k_uncor_fast_evu<7,2,4,2>"; return launch_evu_t<7, 2, 4, 2>(P, A, F, s);
        case 2: *name = "k_uncor_fast_evu<7,2,4,4>"; return launch_evu_t<7, 2, 4, 4>(P, A, F, s);
        case 3: *name = "k_uncor_fast_evu<7,4,2,4>"; return launch_evu_t<7, 4, 2, 4>(P, A, F, s);
        case 4: *name = "k_uncor_fast_evu<7,4,6,4>"; return launch_evu_t<7, 4, 6, 4>(P, A, F, s);
        case 5: *name = "k_uncor_fast_evu<7,4,6,6>"; return launch_evu_t<7, 4, 6, 6>(P, A, F, s);
        case 6: *name = "k_uncor_fast_evu<7,6,6,6>"; return launch_evu_t<7, 6, 6, 6>(P, A, F, s);
        case 7: *name = "k_uncor_fast_evu<9,6,6,6>"; return launch_evu_t<9, 6, 6, 6>(P, A, F, s);
        default: *name = "none"; return hipErrorNotSupported;
        }
    }
    if (!ev_plan_ok(P, A) || (force_wide && ev_plan_wide_ok(P, A))) {   // more rated variables than eight streams hold: the wide list, on the widest instance
        const int64_t blocks = (A.n + (A.col0 & 255) + 255) / 256;
        *name = "k_uncor_fast_evw<9,6,6,6>";
        hipLaunchKernelGGL((k_uncor_fast_evw<9, 6, 6, 6>), dim3((unsigned)blocks), dim3(256), 0, s, P, A, F);
        return hipGetLastError();
    }
    switch (fast_shape_of(P)) {
    case 0: *name = "k_uncor_fast_ev<7,2,2,2>"; return launch_ev_t<7, 2, 2, 2>(P, A, F, s);
    case 1: *name = "k_uncor_fast_ev<7,2,4,2>"; return launch_ev_t<7, 2, 4, 2>(P, A, F, s);
    case 2: *name = "k_uncor_fast_ev<7,2,4,4>"; return launch_ev_t<7, 2, 4, 4>(P, A, F, s);
    case 3: *name = "k_uncor_fast_ev<7,4,2,4>"; return launch_ev_t<7, 4, 2, 4>(P, A, F, s);
    case 4: *name = "k_uncor_fast_ev<7,4,6,4>"; return launch_ev_t<7, 4, 6, 4>(P, A, F, s);
    case 5: *name = "k_uncor_fast_ev<7,4,6,6>"; return launch_ev_t<7, 4, 6, 6>(P, A, F, s);
    case 6: *name = "k_uncor_fast_ev<7,6,6,6>"; return launch_ev_t<7, 6, 6, 6>(P, A, F, s);
    case 7: *name = "k_uncor_fast_ev<9,6,6,6>"; return launch_ev_t<9, 6, 6, 6>(P, A, F, s);
    default: *name = "none"; return hipErrorNotSupported;
    }
}

} // namespace emgpu
