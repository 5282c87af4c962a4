#pragma once
// emgpu_init_karg.h -- init_network (emgpu_device.h) reading the plan through the kernel-argument segment: shared by k_dbn_step2's dense
// 16-variable instances (emgpu_kernels_step2.h) and by the -DEMGPU_FAST_INIT_KARG measuring variant of k_uncor_fast (emgpu_kernels_fast.h).
#include "emgpu_device.h"

namespace emgpu {

// init_network (emgpu_device.h) for the dense 16-variable instances, reading the plan through the kernel-argument segment with the pointer
// laundered once per attempt.  The initial network of 16 variables uses 120 parent strides + five small tables: as loop invariants of the
// attempt loop they are loaded ahead of it, outlive the scalar register file (657 v_writelane ahead of the loop, 708 v_readlane inside it) and
// keep 30 vector registers as spill space for the whole kernel: 155 registers, three waves per SIMD.  Loaded where they are used they are
// scalar loads and nothing else: <= 128 registers, which (with the 36-word LDS rows of coop_dedisc_sc) is a fourth wave.
typedef const __attribute__((address_space(4))) EmgpuPlan *KargPlan;
template <int NI>
__device__ __forceinline__ int32_t init_network_karg(KargPlan Pk, const EmgpuRun &A, Rng &rng, int (&bin)[NI], double (&val)[NI]) {
    int32_t attempts_used = -1;
    const bool no_dedisc = (A.flags & EMGPU_FLAG_NO_DEDISC) != 0;
    for (uint32_t attempt = 0; attempt < (uint32_t)A.max_attempts; attempt++) {
        KargPlan P = Pk;
        asm volatile("" : "+s"(P));   // (the plan's entries are loaded where this attempt uses them)
        rng.attempt = attempt;
        uint4 wc = make_uint4(0, 0, 0, 0);
        int wblk = -1;
#pragma unroll
        for (int p = 0; p < NI; p++) {
            if (p < P->ni && P->i_start[p] != 0) { // bn_sample.m:44-50
                bin[p] = (int)P->i_start[p] - 1;
            } else if (p < P->ni) {
                uint32_t col = 0; // asub2ind.m:13-14 as strides
#pragma unroll
                for (int q = 0; q < p; q++) col += P->i_stride[p][q] * (uint32_t)bin[q];
                const int r = P->i_r[p];
                const int var = P->i_var[p];
                if ((var >> 2) != wblk) { wblk = var >> 2; wc = rng.block(EMGPU_SEC_INIT, 0u, (uint32_t)wblk); }
                bin[p] = draw_bin(P->thr + P->i_off[p] + (size_t)col * (uint32_t)(r - 1), r, word_of(wc, var & 3)); // bn_sample.m:55
            }
        }
        // dbn_hierarchical_sample.m:25-31
        wblk = -1;
#pragma unroll
        for (int p = 0; p < NI; p++) {
            double v = (double)(bin[p] + 1);
            if (p < P->ni && !no_dedisc && P->i_nb[p] != 0 && !P->i_skip[p]) {
                const int var = P->i_var[p];
                if ((var >> 2) != wblk) { wblk = var >> 2; wc = rng.block(EMGPU_SEC_DEDISC_INIT, 0u, (uint32_t)wblk); }
                v = (P->i_zero[p] == bin[p] + 1) ? 0.0 : dedisc_f64(P->bnd, P->i_boff[p], bin[p], word_of(wc, var & 3));
            }
            val[p] = v;
        }
        // UncorEncounterModel.m:259-272
        if (A.pos_L >= 0 && (A.layers != nullptr || (A.flags & EMGPU_FLAG_QUANTIZE500))) {
            double h_ft = pick<NI>(val, A.pos_L);
            if (A.layers != nullptr) {
                int b = (int)h_ft;
                b = b < 1 ? 1 : (b > A.n_layers ? A.n_layers : b);
                const double lo = A.layers[2 * (b - 1)], hi = A.layers[2 * (b - 1) + 1];
                const uint4 wl = rng.block(EMGPU_SEC_LAYER, 0u, 0u);
                {
#pragma clang fp contract(off)
                    const double d = hi - lo;
                    const double m = uniform32(wl.x) * d;
                    h_ft = lo + m;
                }
            }
            if ((A.flags & EMGPU_FLAG_QUANTIZE500) && A.pos_dh >= 0 && pick<NI>(val, A.pos_dh) == 0.0) h_ft = round500(h_ft);
            put<NI>(val, A.pos_L, h_ft);
        }
        bool good = true;
        if (A.pos_v >= 0 && A.pos_dh >= 0) { // :275
#pragma clang fp contract(off)
            const double lhs = pick<NI>(val, A.pos_v) * 1.68781;
            const double rhs = fabs(pick<NI>(val, A.pos_dh)) / 60.0;
            good = lhs > rhs;
        }
        if (good) { attempts_used = (int32_t)attempt + 1; break; }
    }
    return attempts_used;
}

} // namespace emgpu
