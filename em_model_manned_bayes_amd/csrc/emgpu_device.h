// emgpu_device.h -- device-side primitives shared by every kernel: Philox4x32 (EMGPU_PHILOX_ROUNDS rounds), the uniform,
// the threshold draw and dediscretize.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/emgpu.h"
#include "emgpu_plan.h"

namespace emgpu {

// Philox4x32-R, R = EMGPU_PHILOX_ROUNDS (Salmon et al. SC'11; emgpu_plan.h).  The key is wave-uniform (the seed), so the key schedule
// lives in SGPRs / literals; each round is two v_mad_u64_u32 (32x32->64) and four XORs.
__device__ __forceinline__ uint4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < EMGPU_PHILOX_ROUNDS; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        // three-input XOR in one v_bitop3_b32 (truth table 0x96), gfx950
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0 + (uint32_t)r * 0x9E3779B9u, 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1 + (uint32_t)r * 0xBB67AE85u, 0x96);
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
    }
    return make_uint4(c0, c1, c2, c3);
}

// Per-trajectory RNG handle: counter words 0..2 are fixed per (trajectory, attempt).
struct Rng {
    uint32_t c0, c1, attempt, k0, k1;
    __device__ __forceinline__ uint4 block(uint32_t section, uint32_t a, uint32_t blk) const {
        return philox4x32(c0, c1, attempt, (section << 28) | (a << 20) | blk, k0, k1);
    }
};

// word w of a block; w is compile-time or wave-uniform at every call site
__device__ __forceinline__ uint32_t word_of(const uint4 &v, int w) { return w == 0 ? v.x : (w == 1 ? v.y : (w == 2 ? v.z : v.w)); }

// halfword j (0..7) of a block, already shifted into the HIGH half of a 32-bit word
__device__ __forceinline__ uint32_t half_hi(const uint4 &v, int j) {
    const uint32_t w = word_of(v, j >> 1);
    return (j & 1) ? (w & 0xFFFF0000u) : (w << 16);
}
// halfword j (0..7) of a block in the LOW half
__device__ __forceinline__ uint32_t half_lo(const uint4 &v, int j) {
    const uint32_t w = word_of(v, j >> 1);
    return (j & 1) ? (w >> 16) : (w & 0xFFFFu);
}
// split slot: the full 32-bit draw from its primary (hi) and secondary (lo) blocks
__device__ __forceinline__ uint32_t split_draw(const uint4 &hi, const uint4 &lo, int j) { return half_hi(hi, j) | half_lo(lo, j); }

// x' of uniform32
__device__ __forceinline__ uint32_t clamp32(uint32_t x) { return x < 0xFFFFFFFEu ? x : 0xFFFFFFFEu; }

// u = (x' + 0.5) * 2^-32, exact in f64
__device__ __forceinline__ double uniform32(uint32_t x) { return ((double)clamp32(x) + 0.5) * (1.0 / 4294967296.0); }

// select_random.m:17-20 on precompiled thresholds: 0-based bin = #{k < r-1 : x' >= X[k]}
__device__ __forceinline__ int draw_bin(const uint32_t *__restrict__ t, int r, uint32_t x) {
    const uint32_t xp = clamp32(x);
    int b = 0;
    for (int k = 0; k < r - 1; k++) b += (xp >= t[k]) ? 1 : 0;
    return b;
}

// dediscretize.m:33-39  a + (b-a)*rand, in f64 without FMA contraction (bit-exact with the oracle)
__device__ __forceinline__ double dedisc_f64(const double *__restrict__ bnd, int boff, int bin0, uint32_t x) {
#pragma clang fp contract(off)
    const double a = bnd[boff + bin0];
    const double b = bnd[boff + bin0 + 1];
    const double d = b - a;
    const double m = d * uniform32(x);
    return a + m;
}

// sin and cos of |x| <= pi/4 (a reduced angle): Taylor sums to x^19 / x^18 in Horner form on explicit fma -- truncation below
// 1e-19, about two ulp of rounding.  The library's sin() / cos() spend three times the instructions on a range reduction that
// such an argument never needs.  Used where outputs are compared at a tolerance (tracks), never on the sampling path.
// A constant held in a SCALAR register pair at its use.  The f64 coefficients of a loop body are loop invariants: left to the compiler they are
// materialised once, ahead of the loop, and then sit in vector registers for its whole length -- 60 of them in k_terminal_propagate, a third
// of its budget and a whole wave of occupancy.  As scalar operands (one per VALU instruction) they cost two s_mov each at the use.
__device__ __forceinline__ double sk64(double c) { asm volatile("" : "+s"(c)); return c; }
// SK: the coefficients as scalar operands (sk64)
template <bool SK = false>
__device__ __forceinline__ void sincos_small(double x, double &s, double &c) {
    auto K = [](double v) { return SK ? sk64(v) : v; };
    const double z = x * x;
    double ps = K(-1.0 / 121645100408832000.0);          // -1/19!
    ps = fma(ps, z, K(1.0 / 355687428096000.0));         //  1/17!
    ps = fma(ps, z, K(-1.0 / 1307674368000.0));          // -1/15!
    ps = fma(ps, z, K(1.0 / 6227020800.0));              //  1/13!
    ps = fma(ps, z, K(-1.0 / 39916800.0));               // -1/11!
    ps = fma(ps, z, K(1.0 / 362880.0));                  //  1/9!
    ps = fma(ps, z, K(-1.0 / 5040.0));                   // -1/7!
    ps = fma(ps, z, K(1.0 / 120.0));                     //  1/5!
    ps = fma(ps, z, K(-1.0 / 6.0));                      // -1/3!
    s = fma(x * z, ps, x);
    double pc = K(-1.0 / 6402373705728000.0);            // -1/18!
    pc = fma(pc, z, K(1.0 / 20922789888000.0));          //  1/16!
    pc = fma(pc, z, K(-1.0 / 87178291200.0));            // -1/14!
    pc = fma(pc, z, K(1.0 / 479001600.0));               //  1/12!
    pc = fma(pc, z, K(-1.0 / 3628800.0));                // -1/10!
    pc = fma(pc, z, K(1.0 / 40320.0));                   //  1/8!
    pc = fma(pc, z, K(-1.0 / 720.0));                    // -1/6!
    pc = fma(pc, z, K(1.0 / 24.0));                      //  1/4!
    c = (1.0 - 0.5 * z) + (z * z) * pc;
}
// cosd / sind with MATLAB's reduction in degrees: n = round(x/90), x - 90 n in [-45, 45], quadrant m = mod(n, 4)
__device__ __forceinline__ void sincosd_small(double deg, double &s, double &c) {
    const double n = round(deg * (1.0 / 90.0));
    const double x = (3.14159265358979323846 / 180.0) * (deg - n * 90.0);
    const int m = (int)((long long)n & 3ll);
    double sx, cx;
    sincos_small<false>(x, sx, cx);
    s = (m == 0) ? sx : ((m == 1) ? cx : ((m == 2) ? -sx : -cx));
    c = (m == 0) ? cx : ((m == 1) ? -sx : ((m == 2) ? -cx : sx));
}

// run-time (wave-uniform) index into a tiny register array without scratch.  Written with bit
// masks on purpose: a ?: chain over a[q] is folded by LLVM into a variable-index load, which
// forces the whole array into scratch/LDS.
__device__ __forceinline__ uint32_t pick_bits(uint32_t v, bool sel) { return v & (sel ? 0xFFFFFFFFu : 0u); }
template <int N>
__device__ __forceinline__ int pick(const int (&a)[N], int idx) {
    uint32_t r = 0u;
#pragma unroll
    for (int q = 0; q < N; q++) r |= pick_bits((uint32_t)a[q], idx == q);
    return (int)r;
}
template <int N>
__device__ __forceinline__ double pick(const double (&a)[N], int idx) {
    uint32_t lo = 0u, hi = 0u;
#pragma unroll
    for (int q = 0; q < N; q++) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(a[q]);
        lo |= pick_bits((uint32_t)b, idx == q);
        hi |= pick_bits((uint32_t)(b >> 32), idx == q);
    }
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int N, typename T>
__device__ __forceinline__ void put(T (&a)[N], int idx, T v) {
#pragma unroll
    for (int q = 0; q < N; q++) a[q] = (idx == q) ? v : a[q];
}

__device__ __forceinline__ double round500(double num) { // UncorEncounterModel.m:196
    return 500.0 * (floor(num / 500.0) + ((fmod(num, 500.0) > 250.0) ? 1.0 : 0.0));
}

// The presets of one lane (bn_sample.m:44-50): sp[p] = the preset bin (1-based) of the node at position p or 0; a lane's own row of the
// start grid (entry of variable id, 0 = unset) wins over the model's start.  A preset node needs all its parents preset ('Attempt to
// preset a dependent variable', :47) and a bin inside 1..r: otherwise bit 2 of *status is raised (EMGPU_ERR_PRESET) and the preset is
// dropped.  Returns the lane's log-weight: the sum over its preset nodes of log P(preset | parents) read from `logp` (0 without a table).
template <int NI>
__device__ __forceinline__ double lane_presets(const EmgpuPlan &P, const int32_t *start_row, const double *logp, const uint32_t *lp_off, uint32_t *status, int (&sp)[NI]) {
    uint32_t mask = 0u;
    double lw = 0.0;
    uint32_t colbin[NI];
#pragma unroll
    for (int p = 0; p < NI; p++) {
        sp[p] = 0; colbin[p] = 0u;
        if (p >= P.ni) continue;
        int s = start_row ? start_row[P.i_var[p]] : 0;
        if (s == 0) s = (int)P.i_start[p];
        if (s == 0) continue;
        bool ok = s >= 1 && s <= (int)P.i_r[p];
        uint32_t col = 0u;
#pragma unroll
        for (int q = 0; q < p; q++) {
            if (P.i_stride[p][q] == 0u) continue;
            if (!((mask >> q) & 1u)) ok = false;
            col += P.i_stride[p][q] * colbin[q];
        }
        if (!ok) { atomicOr(status, 4u); continue; }
        sp[p] = s; colbin[p] = (uint32_t)(s - 1); mask |= 1u << p;
        if (logp) lw += logp[lp_off[p] + (size_t)col * (uint32_t)P.i_r[p] + (uint32_t)(s - 1)];
    }
    return lw;
}

// Initial network + dediscretize + rejection loop, shared by every DBN kernel.
// bn_sample.m:39-57, dbn_hierarchical_sample.m:25-31, UncorEncounterModel.m:248-281.
// bin[]: 0-based bins by topological position; val[]: dediscretised f64.  Returns the number of
// attempts used (>= 1) or -1 when max_attempts was reached; leaves rng.attempt at the accepted attempt.
template <int NI>
__device__ __forceinline__ int32_t init_network(const EmgpuPlan &P, const EmgpuRun &A, Rng &rng, int (&bin)[NI], double (&val)[NI]) {
    int32_t attempts_used = -1;
    const bool no_dedisc = (A.flags & EMGPU_FLAG_NO_DEDISC) != 0;
    for (uint32_t attempt = 0; attempt < (uint32_t)A.max_attempts; attempt++) {
        rng.attempt = attempt;
        uint4 wc = make_uint4(0, 0, 0, 0);
        int wblk = -1;
#pragma unroll
        for (int p = 0; p < NI; p++) {
            if (p < P.ni && P.i_start[p] != 0) { // bn_sample.m:44-50
                bin[p] = (int)P.i_start[p] - 1;
            } else if (p < P.ni) {
                uint32_t col = 0; // asub2ind.m:13-14 as strides
#pragma unroll
                for (int q = 0; q < p; q++) col += P.i_stride[p][q] * (uint32_t)bin[q];
                const int r = P.i_r[p];
                const int var = P.i_var[p];
                if ((var >> 2) != wblk) { wblk = var >> 2; wc = rng.block(EMGPU_SEC_INIT, 0u, (uint32_t)wblk); }
                bin[p] = draw_bin(P.thr + P.i_off[p] + (size_t)col * (uint32_t)(r - 1), r, word_of(wc, var & 3)); // bn_sample.m:55
            }
        }
        // dbn_hierarchical_sample.m:25-31
        wblk = -1;
#pragma unroll
        for (int p = 0; p < NI; p++) {
            double v = (double)(bin[p] + 1);
            if (p < P.ni && !no_dedisc && P.i_nb[p] != 0 && !P.i_skip[p]) {
                const int var = P.i_var[p];
                if ((var >> 2) != wblk) { wblk = var >> 2; wc = rng.block(EMGPU_SEC_DEDISC_INIT, 0u, (uint32_t)wblk); }
                v = (P.i_zero[p] == bin[p] + 1) ? 0.0 : dedisc_f64(P.bnd, P.i_boff[p], bin[p], word_of(wc, var & 3));
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

// The same with the lane's presets possibly coming from a start grid (Q->start, row `lane`) and its log-weight wanted (Q->log_weight):
// k_dbn_generic (Q null: the model's own start, like init_network).
template <int NI>
__device__ __forceinline__ int32_t init_network_ps(const EmgpuPlan &P, const EmgpuRun &A, const EmgpuPresets *Qp, Rng &rng, int (&bin)[NI], double (&val)[NI], int64_t lane) {
    constexpr bool PS = true;
    int32_t attempts_used = -1;
    const bool no_dedisc = (A.flags & EMGPU_FLAG_NO_DEDISC) != 0;
    int sp[NI];
    if constexpr (PS) {
        if (Qp) {
            const EmgpuPresets &Q = *Qp;
            const double lw = lane_presets<NI>(P, Q.start ? Q.start + (size_t)lane * (size_t)P.ni : nullptr, Q.log_weight ? Q.logp : nullptr, Q.lp_off, A.status, sp);
            if (Q.log_weight) Q.log_weight[lane] = lw;
        } else {
#pragma unroll
            for (int p = 0; p < NI; p++) sp[p] = p < P.ni ? (int)P.i_start[p] : 0;
        }
    }
    for (uint32_t attempt = 0; attempt < (uint32_t)A.max_attempts; attempt++) {
        rng.attempt = attempt;
        uint4 wc = make_uint4(0, 0, 0, 0);
        int wblk = -1;
#pragma unroll
        for (int p = 0; p < NI; p++) {
            const int preset = PS ? sp[p] : (p < P.ni ? (int)P.i_start[p] : 0);
            if (p < P.ni && preset != 0) { // bn_sample.m:44-50
                bin[p] = preset - 1;
            } else if (p < P.ni) {
                uint32_t col = 0; // asub2ind.m:13-14 as strides
#pragma unroll
                for (int q = 0; q < p; q++) col += P.i_stride[p][q] * (uint32_t)bin[q];
                const int r = P.i_r[p];
                const int var = P.i_var[p];
                if ((var >> 2) != wblk) { wblk = var >> 2; wc = rng.block(EMGPU_SEC_INIT, 0u, (uint32_t)wblk); }
                bin[p] = draw_bin(P.thr + P.i_off[p] + (size_t)col * (uint32_t)(r - 1), r, word_of(wc, var & 3)); // bn_sample.m:55
            }
        }
        // dbn_hierarchical_sample.m:25-31
        wblk = -1;
#pragma unroll
        for (int p = 0; p < NI; p++) {
            double v = (double)(bin[p] + 1);
            if (p < P.ni && !no_dedisc && P.i_nb[p] != 0 && !P.i_skip[p]) {
                const int var = P.i_var[p];
                if ((var >> 2) != wblk) { wblk = var >> 2; wc = rng.block(EMGPU_SEC_DEDISC_INIT, 0u, (uint32_t)wblk); }
                v = (P.i_zero[p] == bin[p] + 1) ? 0.0 : dedisc_f64(P.bnd, P.i_boff[p], bin[p], word_of(wc, var & 3));
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
