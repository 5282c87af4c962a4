// emgpu_events.h -- event lists from the 8-second-block kernels (k_uncor_fast_ev, k_dbn_step2 with EV).
//
// What UncorEncounterModel.sample returns (UncorEncounterModel.m:283-300) is the event list of dbn_hierarchical_sample.m:16-37:
// per second c = 1..T first the resample rows of resample_events.m:16-37 in ascending variable id -- of EVERY variable with a
// rate, also the static ones that the dense trace never shows -- then, for c < T, the transition rows of dbn_sample.m:151-161 (or
// :84-91 on the dependent branch) in ascending variable id, and the terminator row.  An 8-second block of the dense kernels
// already holds all of it but two things: the re-draws of static variables (their own RES slots: one more Philox call per such
// variable and block, the same packed compare) and the value of a resample row that a transition of the same second hides in
// the dense trace (its DEDISC_RES draw is made here, by the lane that needs it; ~1 per trajectory).  The flag streams of a
// block -- 8 - ND resample streams, ND transition streams, MSB-first -- are transposed into one 64-bit mask whose leading bit is
// the lane's next event in list order.
#pragma once
#include "emgpu_coop.h"
#include "emgpu_device.h"

namespace emgpu {

// 8 x 8 bit-matrix transpose (bit 8r + c <-> bit 8c + r): three delta swaps
__device__ __forceinline__ uint64_t transpose8x8(uint64_t x) {
    uint64_t t;
    t = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAull; x ^= t ^ (t << 7);
    t = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCull; x ^= t ^ (t << 14);
    t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ull; x ^= t ^ (t << 28);
    return x;
}

// the eight resample Bernoullis of one block of a STATIC variable (resample_events.m:24): high halfwords two seconds per
// instruction like the dynamic variables' pass; a tie with R's high half anywhere in the wave redoes the lane's eight on 32 bits.
__device__ __forceinline__ uint32_t static_hits8(const Rng &rng, uint32_t var, int g8, uint32_t R, uint32_t RR1) {
    const uint4 rh = rng.block(EMGPU_SEC_RES, var, (uint32_t)g8);
    uint32_t hitA = 0u;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        uint32_t u;
        asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(u) : "s"(RR1), "v"(word_of(rh, p)));
        asm("v_pk_min_u16 %0, %0, 2 op_sel_hi:[1,0]" : "+v"(u));
        hitA = p ? ((hitA << 2) | u) : u;
    }
    uint32_t hit8 = (hitA & 0xAAu) | ((hitA >> 17) & 0x55u);
    if (__ballot((hitA & 0x00550055u) != 0u) != 0ull) {
        const uint4 rl = rng.block(EMGPU_SEC_RES_LO, var, (uint32_t)g8);
        hit8 = 0u;
#pragma unroll
        for (int j = 0; j < 8; j++) hit8 = (hit8 << 1) | (clamp32(split_draw(rh, rl, j)) < R ? 1u : 0u);
    }
    return hit8;
}

// wave-uniform description of the eight event streams of a block, byte b of every word for stream b: resample of the b-th variable
// with a rate (b < 8 - ND, ascending variable id), transition of the (b - (8 - ND))-th dynamic variable in ascending variable id
struct EvPlan {
    uint64_t var1;  // 1-based variable id of the row
    uint64_t kdyn;  // dynamic-variable index k, 0xFF: a static variable (or no stream)
    uint64_t zero;  // zero bin (0: none)
    uint64_t nb;    // number of boundaries (0: categorical, the value is the bin)
    uint64_t boff;  // boundaries offset of streams 0-3 (16 bits each); boff4: of stream 4
    uint32_t boff4, nact;
    uint32_t R[5];    // resample hit thresholds
    uint32_t RR1[5];  // (R >> 16) + 1 in both halfwords
};
template <int NI, int ND>
__device__ __forceinline__ EvPlan ev_plan_of(const EmgpuPlan &P) {
    constexpr int NRES = 8 - ND;
    EvPlan E{};
    E.nact = (uint32_t)P.nact;
#pragma unroll
    for (int b = 0; b < NRES; b++) {
        E.R[b] = 0u; E.RR1[b] = 0x00010001u;
        uint64_t kd = 0xFFull;
        if (b < P.nact) {
            const uint32_t pos = P.a_pos[b];
            E.var1 |= (uint64_t)(P.a_var[b] + 1u) << (8 * b);
            kd = (uint64_t)(uint8_t)P.a_dyn[b];
            E.zero |= (uint64_t)P.i_zero[pos] << (8 * b);
            E.nb |= (uint64_t)P.i_nb[pos] << (8 * b);
            if (b < 4) E.boff |= (uint64_t)P.i_boff[pos] << (16 * b); else E.boff4 = P.i_boff[pos];
            E.R[b] = P.a_R[b]; E.RR1[b] = ((P.a_R[b] >> 16) + 1u) * 0x00010001u;
        }
        E.kdyn |= kd << (8 * b);
    }
#pragma unroll
    for (int e = 0; e < ND; e++) {
        if (e >= P.nd) { E.kdyn |= 0xFFull << (8 * (NRES + e)); continue; }   // an instance built for more dynamic variables than the model has
        const uint32_t k = P.d_emit[e];
        E.var1 |= (uint64_t)(P.d_ivar[k] + 1u) << (8 * (NRES + e));
        E.kdyn |= (uint64_t)k << (8 * (NRES + e));
        E.zero |= (uint64_t)P.d_zero[k] << (8 * (NRES + e));
        E.nb |= (uint64_t)P.d_nb[k] << (8 * (NRES + e));
    }
    return E;
}

// per-lane state of the event list
struct EvState {
    uint64_t *ev;      // the lane's list
    uint32_t count, last_t, cap;
    uint64_t sbins;    // byte b: 1-based bin of static stream b
    __device__ __forceinline__ void emit(uint32_t at, uint32_t var1, uint32_t bin1, float v) {   // dbn_hierarchical_sample.m:33-37 row [dt var value], bin alongside
        const uint32_t dt = at - last_t;
        last_t = at;
        if (count < cap) ev[count] = (uint64_t)(dt & 0xFFFFu) | ((uint64_t)var1 << 16) | ((uint64_t)bin1 << 24) | ((uint64_t)__float_as_uint(v) << 32);
        count++;
    }
};
template <int NI, int ND>
__device__ __forceinline__ EvState ev_state_of(const EmgpuPlan &P, const EmgpuRun &A, const int (&bin)[NI], bool valid, int64_t i) {
    EvState S{};
    S.ev = A.events + (size_t)(valid ? i : 0) * (size_t)A.event_cap;
    S.cap = valid ? (uint32_t)A.event_cap : 0u;
#pragma unroll
    for (int b = 0; b < 8 - ND; b++)
        if (b < P.nact && P.a_dyn[b] < 0) S.sbins |= (uint64_t)(uint32_t)(pick<NI>(bin, P.a_pos[b]) + 1) << (8 * b);
    return S;
}
// host side: can a plan's event list be written by the block kernels?  At most 8 - nd variables with a rate, every rate below
// the packed compare's limit
inline bool ev_plan_ok(const EmgpuPlan &P, const EmgpuRun &A) {
    if (P.nact > 8 - P.nd || P.nact > 5 || A.event_cap < 1) return false;
    for (int a = 0; a < P.nact; a++)
        if (P.a_R[a] >= 0xFFFF0000u) return false;
    return true;
}

// dediscretize.m:33-39 for a resample row whose draw the cooperative pass did not make (a static variable, or a row hidden by a
// transition of the same second): slot (DEDISC_RES, variable, second c), made by the lane itself
__device__ __attribute__((noinline)) float ev_draw(uint32_t c0, uint32_t c1, uint32_t attempt, uint32_t k0, uint32_t k1, const double *bnd,
                                                   uint32_t var0, uint32_t c, uint32_t bin0, uint32_t boff) {
    const uint4 r4 = philox4x32(c0, c1, attempt, (EMGPU_SEC_DEDISC_RES << 28) | (var0 << 20) | (c >> 2), k0, k1);
    const uint32_t w = c & 3u;
    return (float)dedisc_f64(bnd, (int)boff, (int)bin0, w == 0 ? r4.x : (w == 1 ? r4.y : (w == 2 ? r4.z : r4.w)));
}

// The rows of one 8-second block, after the cooperative dediscretize of the block (result slots and published bins in the lane's LDS row).
// hitp / chgp: the resample-hit and changed streams of the dynamic variables, byte k = variable k, MSB-first (bit 7 - j <-> second j);
// prevp: byte k = the bin variable k had when the block began.
template <int ND, bool LB>
__device__ __forceinline__ void ev_emit_block(const CoopLds<ND, LB> &W, int lane, const EvPlan &E, EvState &S, const Rng &rng, const double *bnd,
                                              int g8, int T, bool valid, uint32_t hitp, uint32_t chgp, uint32_t prevp,
                                              bool values_are_bins = false /* EMGPU_FLAG_NO_DEDISC (plain dbn_sample.m) */) {
    constexpr int NRES = 8 - ND;
    // streams: byte 7 - b of `in` = stream b, bit 7 - j = second j; after the transpose bit 63 - (8 j + b) is event (j, b)
    const uint32_t live8 = (g8 == 0 ? 0x7Fu : 0xFFu) & (8 * g8 + 7 < T ? 0xFFu : (0xFF00u >> (T - 8 * g8)) & 0xFFu);   // seconds 1 <= c < T
    uint64_t in = 0ull;
#pragma unroll
    for (int b = 0; b < NRES; b++) {
        if (b >= (int)E.nact) continue;                       // wave-uniform
        const uint32_t kd = (uint32_t)(E.kdyn >> (8 * b)) & 0xFFu;
        uint32_t st;
        if (kd != 0xFFu) st = (hitp >> (8u * kd)) & 0xFFu;    // a dynamic variable: decided by its 8-second pass
        else st = static_hits8(rng, ((uint32_t)(E.var1 >> (8 * b)) & 0xFFu) - 1u, g8, E.R[b], E.RR1[b]) & live8;
        in |= (uint64_t)st << (8 * (7 - b));
    }
#pragma unroll
    for (int e = 0; e < ND; e++) {
        const uint32_t kd = (uint32_t)(E.kdyn >> (8 * (NRES + e))) & 0xFFu;
        if (kd != 0xFFu) in |= (uint64_t)((chgp >> (8u * kd)) & 0xFFu) << (8 * (7 - (NRES + e)));   // wave-uniform
    }
    uint64_t pend = valid ? transpose8x8(in) : 0ull;
    const uint8_t *bins8 = reinterpret_cast<const uint8_t *>(&W.res[lane * CoopLds<ND, LB>::kStride + CoopLds<ND, LB>::kBins]);
    const float *res32 = &W.res[lane * CoopLds<ND, LB>::kStride];
    while (__ballot(pend != 0ull) != 0ull) {
        if (pend != 0ull) {
            const uint32_t pos = (uint32_t)__clzll((long long)pend);
            pend &= ~(0x8000000000000000ull >> pos);
            const uint32_t j = pos >> 3, b = pos & 7u, sh = 8u * b;
            const uint32_t var1 = (uint32_t)(E.var1 >> sh) & 0xFFu, kd = (uint32_t)(E.kdyn >> sh) & 0xFFu, zb = (uint32_t)(E.zero >> sh) & 0xFFu;
            const uint32_t c = 8u * (uint32_t)g8 + j;
            uint32_t bin1;
            float v = 0.f;
            bool draw = false;
            if (b >= (uint32_t)NRES) {                          // a transition row: the new bin and its value
                bin1 = bins8[8u * kd + j];
                if (bin1 != zb) v = res32[8u * kd + j];
            } else if (kd != 0xFFu) {                           // a resample row of a dynamic variable: the bin BEFORE this second's transition
                bin1 = j ? bins8[8u * kd + j - 1u] : ((prevp >> (8u * kd)) & 0xFFu);
                const bool hidden = ((chgp >> (8u * kd + 7u - j)) & 1u) != 0u;
                if (bin1 != zb) { if (hidden) draw = true; else v = res32[8u * kd + j]; }
            } else {                                            // a resample row of a static variable
                bin1 = (uint32_t)(S.sbins >> sh) & 0xFFu;
                const uint32_t nb = (uint32_t)(E.nb >> sh) & 0xFFu;
                if (nb == 0u) v = (float)bin1; else if (bin1 != zb) draw = true;
            }
            if (values_are_bins) { draw = false; v = (float)bin1; }
            if (draw) {
                const uint32_t boff = b < 4u ? (uint32_t)(E.boff >> (16u * b)) & 0xFFFFu : E.boff4;
                v = ev_draw(rng.c0, rng.c1, rng.attempt, rng.k0, rng.k1, bnd, var1 - 1u, c, bin1 - 1u, boff);
            }
            S.emit(c, var1, bin1, v);
        }
    }
}

// resample_events.m:16-37 also covers second T (the list runs to sum(dt) = T; the dense trace has no column T), then the terminator
// row (dbn_hierarchical_sample.m:15-19) and the count.  curp: byte k = the final bin of dynamic variable k.
template <int ND>
__device__ __forceinline__ void ev_tail(const EvPlan &E, EvState &S, const Rng &rng, const double *bnd, int T, uint32_t curp, const EmgpuRun &A, bool valid, int64_t i) {
    constexpr int NRES = 8 - ND;
    const uint32_t Tu = (uint32_t)T;
#pragma unroll
    for (int b = 0; b < NRES; b++) {
        if (b >= (int)E.nact) continue;
        const uint32_t var0 = ((uint32_t)(E.var1 >> (8 * b)) & 0xFFu) - 1u, kd = (uint32_t)(E.kdyn >> (8 * b)) & 0xFFu;
        const uint4 rh = rng.block(EMGPU_SEC_RES, var0, Tu >> 3), rl = rng.block(EMGPU_SEC_RES_LO, var0, Tu >> 3);
        uint32_t x = 0u;
#pragma unroll
        for (int j = 0; j < 8; j++) x = ((Tu & 7u) == (uint32_t)j) ? split_draw(rh, rl, j) : x;
        if (clamp32(x) < E.R[b]) {
            const uint32_t zb = (uint32_t)(E.zero >> (8 * b)) & 0xFFu, nb = (uint32_t)(E.nb >> (8 * b)) & 0xFFu;
            const uint32_t bin1 = kd != 0xFFu ? ((curp >> (8u * kd)) & 0xFFu) : ((uint32_t)(S.sbins >> (8 * b)) & 0xFFu);
            float v = 0.f;
            if (nb == 0u) v = (float)bin1;
            else if (bin1 != zb) v = ev_draw(rng.c0, rng.c1, rng.attempt, rng.k0, rng.k1, bnd, var0, Tu, bin1 - 1u, b < 4 ? (uint32_t)(E.boff >> (16 * b)) & 0xFFFFu : E.boff4);
            S.emit(Tu, var0 + 1u, bin1, v);
        }
    }
    if (!(A.flags & EMGPU_FLAG_NO_TERMINATOR)) S.emit(Tu, 0u, 0u, 0.f);
    if (valid) {
        A.ev_count[i] = S.count;
        if (S.count > (uint32_t)A.event_cap) atomicOr(A.status, 2u);
    }
}


// ---- WIDE lists (round 4): more variables with a resample rate than the 8 - ND streams above hold (haa_v1: 7 of its 9 variables).
// The rule of resample_events.m:16-37 is unchanged -- per second the resample rows of EVERY rated variable in ascending id, then the
// transition rows -- only the block's flag matrix grows to 16 streams per second: stream b < 16 - ND = the b-th rated variable (ascending
// id, static or dynamic alike), the last ND = the transitions.  Two 8 x 8 transposes, byte-interleaved per second, give a 128-bit mask
// whose leading bit is the lane's next row; what a stream is (variable, zero bin, boundaries, threshold) sits in LDS (EvStream, filled
// once per workgroup) instead of packed registers.
struct EvStream { uint8_t var1, kdyn, zero, nb; uint16_t boff, pad; uint32_t R, RR1; };
template <int ND>
__device__ __forceinline__ void ev_wide_plan(const EmgpuPlan &P, EvStream *s_ev /* [16] in LDS; call from every thread, then barrier */) {
    constexpr int NRES = 16 - ND;
    const int b = threadIdx.x;
    if (b < 16) {
        EvStream E{0, 0xFF, 0, 0, 0, 0, 0u, 0x00010001u};
        if (b < NRES) {
            if (b < P.nact) {
                const uint32_t pos = P.a_pos[b];
                E.var1 = (uint8_t)(P.a_var[b] + 1u); E.kdyn = (uint8_t)P.a_dyn[b]; E.zero = P.i_zero[pos]; E.nb = P.i_nb[pos]; E.boff = P.i_boff[pos];
                E.R = P.a_R[b]; E.RR1 = ((P.a_R[b] >> 16) + 1u) * 0x00010001u;
            }
        } else if (b - NRES < P.nd) {
            const uint32_t k = P.d_emit[b - NRES];
            E.var1 = (uint8_t)(P.d_ivar[k] + 1u); E.kdyn = (uint8_t)k; E.zero = P.d_zero[k]; E.nb = P.d_nb[k]; E.boff = P.d_boff[k];
        }
        s_ev[b] = E;
    }
}
struct EvStateW {
    EvState S;
    uint64_t sb_lo, sb_hi;   // byte b: 1-based bin of rated variable b when it is static
};
template <int NI, int ND>
__device__ __forceinline__ EvStateW ev_state_w_of(const EmgpuPlan &P, const EmgpuRun &A, const int (&bin)[NI], bool valid, int64_t i) {
    EvStateW W{};
    W.S.ev = A.events + (size_t)(valid ? i : 0) * (size_t)A.event_cap;
    W.S.cap = valid ? (uint32_t)A.event_cap : 0u;
#pragma unroll
    for (int b = 0; b < 16 - ND; b++)
        if (b < P.nact && P.a_dyn[b] < 0) {
            const uint64_t v = (uint64_t)(uint32_t)(pick<NI>(bin, P.a_pos[b]) + 1);
            if (b < 8) W.sb_lo |= v << (8 * b); else W.sb_hi |= v << (8 * (b - 8));
        }
    return W;
}
// bytes a3 a2 a1 a0 and c3 c2 c1 c0 -> a3 c3 a2 c2 a1 c1 a0 c0
__device__ __forceinline__ uint64_t ev_interleave_bytes(uint32_t a, uint32_t c) {
    const uint32_t hi = __builtin_amdgcn_perm(a, c, 0x07030602u), lo = __builtin_amdgcn_perm(a, c, 0x05010400u);
    return ((uint64_t)hi << 32) | lo;
}
inline bool ev_plan_wide_ok(const EmgpuPlan &P, const EmgpuRun &A) {
    if (P.nact > 16 - P.nd || A.event_cap < 1) return false;
    for (int a = 0; a < P.nact; a++)
        if (P.a_R[a] >= 0xFFFF0000u) return false;
    return true;
}
// the rows of one block as a 128-bit mask in list order: bit 63 - (16 j + b) of `hi` = (second j < 4, stream b), `lo` the same for seconds 4-7
template <int ND>
__device__ __forceinline__ void ev_wide_mask(const EvStream *s_ev, int nact /* 0: no resample rows (plain dbn_sample.m) */, const Rng &rng, int g8, int T, bool valid,
                                             uint32_t hitp, uint32_t chgp, uint64_t &pend_hi, uint64_t &pend_lo) {
    constexpr int NRES = 16 - ND;
    const uint32_t live8 = (g8 == 0 ? 0x7Fu : 0xFFu) & (8 * g8 + 7 < T ? 0xFFu : (0xFF00u >> (T - 8 * g8)) & 0xFFu);   // seconds 1 <= c < T
    uint64_t in1 = 0ull, in2 = 0ull;   // streams 0-7 / 8-15: byte 7 - (b & 7) = stream b, bit 7 - j = second j
    for (int b = 0; b < nact; b++) {                           // wave-uniform trip count
        const uint32_t kd = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ev[b].kdyn);
        uint32_t st;
        if (kd != 0xFFu) st = (hitp >> (8u * kd)) & 0xFFu;
        else {
            const uint32_t var0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ev[b].var1) - 1u;
            const uint32_t Rb = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ev[b].R), RR1b = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ev[b].RR1);
            st = static_hits8(rng, var0, g8, Rb, RR1b) & live8;
        }
        if (b < 8) in1 |= (uint64_t)st << (8 * (7 - b)); else in2 |= (uint64_t)st << (8 * (15 - b));
    }
#pragma unroll
    for (int e = 0; e < ND; e++) {
        const uint32_t kd = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ev[NRES + e].kdyn);
        if (kd != 0xFFu) in2 |= (uint64_t)((chgp >> (8u * kd)) & 0xFFu) << (8 * (15 - (NRES + e)));
    }
    const uint64_t t1 = valid ? transpose8x8(in1) : 0ull, t2 = valid ? transpose8x8(in2) : 0ull;   // byte 7 - j = second j, bit 7 - (b & 7) = stream b
    pend_hi = ev_interleave_bytes((uint32_t)(t1 >> 32), (uint32_t)(t2 >> 32));                     // seconds 0-3: bit 63 - (16 j + b)
    pend_lo = ev_interleave_bytes((uint32_t)t1, (uint32_t)t2);                                     // seconds 4-7
}
template <int ND, bool LB>
__device__ __forceinline__ void ev_emit_block_wide(const CoopLds<ND, LB> &W, int lane, const EvStream *s_ev, int nact, EvStateW &SW, const Rng &rng, const double *bnd,
                                                   int g8, int T, bool valid, uint32_t hitp, uint32_t chgp, uint32_t prevp) {
    constexpr int NRES = 16 - ND;
    uint64_t pend_hi, pend_lo;
    ev_wide_mask<ND>(s_ev, nact, rng, g8, T, valid, hitp, chgp, pend_hi, pend_lo);
    const uint8_t *bins8 = reinterpret_cast<const uint8_t *>(&W.res[lane * CoopLds<ND, LB>::kStride + CoopLds<ND, LB>::kBins]);
    const float *res32 = &W.res[lane * CoopLds<ND, LB>::kStride];
    while (__ballot((pend_hi | pend_lo) != 0ull) != 0ull) {
        if ((pend_hi | pend_lo) != 0ull) {
            uint32_t pos;
            if (pend_hi != 0ull) { pos = (uint32_t)__clzll((long long)pend_hi); pend_hi &= ~(0x8000000000000000ull >> pos); }
            else { pos = (uint32_t)__clzll((long long)pend_lo); pend_lo &= ~(0x8000000000000000ull >> pos); pos += 64u; }
            const uint32_t j = pos >> 4, b = pos & 15u;
            const EvStream E = s_ev[b];
            const uint32_t var1 = E.var1, kd = E.kdyn, zb = E.zero;
            const uint32_t c = 8u * (uint32_t)g8 + j;
            uint32_t bin1;
            float v = 0.f;
            bool draw = false;
            if (b >= (uint32_t)NRES) {                          // a transition row: the new bin and its value
                bin1 = bins8[8u * kd + j];
                if (bin1 != zb) v = res32[8u * kd + j];
            } else if (kd != 0xFFu) {                           // a resample row of a dynamic variable: the bin BEFORE this second's transition
                bin1 = j ? bins8[8u * kd + j - 1u] : ((prevp >> (8u * kd)) & 0xFFu);
                const bool hidden = ((chgp >> (8u * kd + 7u - j)) & 1u) != 0u;
                if (bin1 != zb) { if (hidden) draw = true; else v = res32[8u * kd + j]; }
            } else {                                            // a resample row of a static variable
                bin1 = (uint32_t)((b < 8u ? SW.sb_lo >> (8u * b) : SW.sb_hi >> (8u * (b - 8u))) & 0xFFull);
                if (E.nb == 0u) v = (float)bin1; else if (bin1 != zb) draw = true;
            }
            if (draw) v = ev_draw(rng.c0, rng.c1, rng.attempt, rng.k0, rng.k1, bnd, var1 - 1u, c, bin1 - 1u, E.boff);
            SW.S.emit(c, var1, bin1, v);
        }
    }
}
// ---- ROWS BY THE WAVE (round 4): an events-only call of a wide model.  ev_emit_block_wide walks each lane's mask row by row: the wave runs
// at the pace of its longest list (haa_v1: 14 rows per lane and block on average, 22+ at most), every row that needs a draw of its own
// (a static variable, a hidden resample row) is a Philox call for the few lanes that have one, and the dynamic variables' values took a
// cooperative pass of their own before.  Here a row IS the unit of work: the set bits of all 64 masks are queued (positions from one prefix
// sum, like coop_dedisc; a lane's requests in list order), and a worker lane builds one row from the owner's published block state -- its
// place in the list from its distance to the owner's first request, dt from the request before it, the bin by the stream's kind, ONE dediscretize draw in the row's own
// slot (DEDISC_TRANS for a transition row, DEDISC_RES for every resample row: no row shares a result slot, so nothing is "hidden") -- and
// stores it.  No result slots, no per-lane loop; 64 rows per pass whatever the lists' lengths.
// Published per lane (the result slots are free in this mode), words of the lane's CoopLds row: 0 rows so far, 1 time of the last row,
// 2 the bins the block began with, 3 the queue position of the lane's first request, 4-7 the static
// variables' bins (ev_rows_publish_static); the mask itself stays with its lane.
// (Measured and dropped: rows held back in LDS until their aligned group of eight is complete, so that a list is written 64 bytes at a
// time -- 10-15 % slower on every model: what looked like 6x write amplification on cor_v1's lists was the scratch traffic of a private
// copy of the plan, emgpu_kernels_step2.h.)
// LDS of the instances that build rows by the wave, and of no other instance: the stream table and a request queue per wave
template <bool ON>
__device__ __forceinline__ EvStream *ev_rows_stream_lds() {
    if constexpr (ON) { __shared__ EvStream s[16]; return s; }
    else return nullptr;
}
// requests per compaction round (uint16_t each; entry 0 of the array = the last request of the round before): 1024, or 512 where the
// workgroup's LDS decides the occupancy (the 4-variable instances: 48 KB of cooperative state per workgroup, three workgroups per CU)
template <int ND> constexpr int kEvRowsQueue = ND >= 4 ? 512 : 1024;
template <bool ON, int ND>
__device__ __forceinline__ uint16_t *ev_rows_queue_lds(int wave) {
    if constexpr (ON) { __shared__ uint16_t q[4][kEvRowsQueue<ND> + 2]; return q[wave]; }
    else return nullptr;
}
template <int ND>
__device__ __forceinline__ void ev_rows_publish_static(CoopLds<ND, true> &W, int lane, const EvStateW &SW) {
    uint32_t *row = reinterpret_cast<uint32_t *>(&W.res[lane * CoopLds<ND, true>::kStride]);
    row[4] = (uint32_t)SW.sb_lo; row[5] = (uint32_t)(SW.sb_lo >> 32); row[6] = (uint32_t)SW.sb_hi; row[7] = (uint32_t)(SW.sb_hi >> 32);
}
// One pass: worker lane l builds the row of request q0 + l.  A lane's requests sit in the queue in list order, one after the other, from
// position row[7] on: the row's rank in its block is its distance from there, and the row before it is the request before it.
template <int ND>
__device__ __forceinline__ void ev_rows_worker(const CoopLds<ND, true> &W, const uint16_t *queue /* entry k + 1 = request rb + k */, int lane, uint32_t rb, uint32_t q0,
                                               uint32_t cnt, const EvStream *s_ev, const Rng &rng, const double *bnd, int g8, const EmgpuRun &A, int64_t i) {
    using L = CoopLds<ND, true>;
    constexpr uint32_t NRES = 16 - ND;
    const uint32_t q = q0 + (uint32_t)lane;
    if (q < cnt) {
        const uint32_t d = queue[q + 1u];
        const uint32_t owner = d & 63u, pos = d >> 6;
        const uint32_t *orow = reinterpret_cast<const uint32_t *>(&W.res[owner * L::kStride]);
        const uint32_t rank = rb + q - orow[3];
        const uint32_t j = pos >> 4, b = pos & 15u;
        const uint32_t c = 8u * (uint32_t)g8 + j;
        const uint32_t dt = rank ? j - ((uint32_t)queue[q] >> 10) : c - orow[1];
        const uint32_t nrow = orow[0] + rank;
        const EvStream E = s_ev[b];
        const uint32_t var1 = E.var1, kd = E.kdyn, zb = E.zero;
        const uint8_t *bins8 = reinterpret_cast<const uint8_t *>(orow + L::kBins);
        uint32_t bin1, sec = EMGPU_SEC_DEDISC_RES;
        if (b >= NRES) { bin1 = bins8[8u * kd + j]; sec = EMGPU_SEC_DEDISC_TRANS; }                           // a transition row: the new bin
        else if (kd != 0xFFu) bin1 = j ? bins8[8u * kd + j - 1u] : ((orow[2] >> (8u * kd)) & 0xFFu);           // a resample row: the bin before this second's transition
        else bin1 = (orow[4u + (b >> 2)] >> (8u * (b & 3u))) & 0xFFu;                                          // ... of a static variable
        float v = 0.f;
        if ((b < NRES && kd == 0xFFu && E.nb == 0u) || (A.flags & EMGPU_FLAG_NO_DEDISC)) v = (float)bin1;   // (plain dbn_sample.m: the value of a row is its bin)
        else if (bin1 != zb) {
            const uint64_t go = *reinterpret_cast<const uint64_t *>(orow + L::kGidx);
            const uint4 r4 = philox4x32((uint32_t)go, (uint32_t)(go >> 32), W.attempt[owner], (sec << 28) | ((var1 - 1u) << 20) | (c >> 2), rng.k0, rng.k1);
            const uint32_t w = c & 3u;
            v = (float)dedisc_f64(bnd, (int)E.boff, (int)bin1 - 1, w == 0 ? r4.x : (w == 1 ? r4.y : (w == 2 ? r4.z : r4.w)));
        }
        if (nrow < (uint32_t)A.event_cap)
            A.events[(size_t)(i - lane + (int64_t)owner) * (size_t)A.event_cap + nrow] =
                (uint64_t)(dt & 0xFFFFu) | ((uint64_t)var1 << 16) | ((uint64_t)bin1 << 24) | ((uint64_t)__float_as_uint(v) << 32);
    }
}
template <int ND, int KQ = kEvRowsQueue<ND>>
__device__ __forceinline__ void ev_rows_block_wide(CoopLds<ND, true> &W, uint16_t *queue /* KQ + 2 entries */, int lane, const EvStream *s_ev, int nact, EvStateW &SW, const Rng &rng,
                                                   const double *bnd, int g8, int T, bool valid, uint32_t hitp, uint32_t chgp, uint32_t prevp, const EmgpuRun &A, int64_t i) {
    using L = CoopLds<ND, true>;
    uint64_t hi, lo;
    ev_wide_mask<ND>(s_ev, (A.flags & EMGPU_FLAG_NO_RESAMPLE) ? 0 : nact, rng, g8, T, valid, hitp, chgp, hi, lo);
    const uint32_t c = (uint32_t)__popcll(hi) + (uint32_t)__popcll(lo);
    const uint32_t inc = wave_inclusive_add(c);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    if (total == 0u) return;
    uint32_t *row = reinterpret_cast<uint32_t *>(&W.res[lane * L::kStride]);
    uint32_t a = inc - c;
    row[0] = SW.S.count; row[1] = SW.S.last_t; row[2] = prevp; row[3] = a;
    if (total <= (uint32_t)KQ) {
        // the usual case, one round: the four words of the mask one after the other, leading bit first (a lane's requests stay in list order)
        uint16_t *qp = queue + 1 + a;
        const uint32_t w4[4] = {(uint32_t)(hi >> 32), (uint32_t)hi, (uint32_t)(lo >> 32), (uint32_t)lo};
#pragma unroll
        for (int w = 0; w < 4; w++) {
            uint32_t m = w4[w];
            while (__ballot(m != 0u) != 0ull) {
                if (m != 0u) {
                    const uint32_t p = (uint32_t)__clz((int)m);
                    m &= ~(0x80000000u >> p);
                    *qp++ = (uint16_t)((uint32_t)lane | ((p + 32u * (uint32_t)w) << 6));
                }
            }
        }
        wave_sync();
        for (uint32_t q0 = 0u; q0 < total; q0 += 64u) ev_rows_worker<ND>(W, queue, lane, 0u, q0, total, s_ev, rng, bnd, g8, A, i);
        wave_sync();
    } else {
        const uint32_t aend = inc;
        uint64_t mh = hi, ml = lo;
        for (uint32_t rb = 0u; rb < total; rb += (uint32_t)KQ) {
            if (rb != 0u) { if (lane == 0) queue[0] = queue[KQ]; wave_sync(); }   // the request before this round's first
            const uint32_t lim = min(aend, rb + (uint32_t)KQ);   // a lane is active while a < lim
            while (__ballot(a < lim) != 0ull) {
                if (a < lim) {
                    uint32_t pos;
                    if (mh != 0ull) { pos = (uint32_t)__clzll((long long)mh); mh &= ~(0x8000000000000000ull >> pos); }
                    else { pos = (uint32_t)__clzll((long long)ml); ml &= ~(0x8000000000000000ull >> pos); pos += 64u; }
                    queue[1u + a - rb] = (uint16_t)((uint32_t)lane | (pos << 6));
                    a++;
                }
            }
            wave_sync();
            const uint32_t cnt = min(total - rb, (uint32_t)KQ);
            for (uint32_t q0 = 0u; q0 < cnt; q0 += 64u) ev_rows_worker<ND>(W, queue, lane, rb, q0, cnt, s_ev, rng, bnd, g8, A, i);
            wave_sync();
        }
    }
    SW.S.count += c;
    if (c != 0u) SW.S.last_t = 8u * (uint32_t)g8 + ((lo != 0ull ? 127u - (uint32_t)__builtin_ctzll(lo) : 63u - (uint32_t)__builtin_ctzll(hi)) >> 4);
}
template <int ND>
__device__ __forceinline__ void ev_tail_wide(const EvStream *s_ev, int nact, EvStateW &SW, const Rng &rng, const double *bnd, int T, uint32_t curp, const EmgpuRun &A, bool valid, int64_t i) {
    const uint32_t Tu = (uint32_t)T;
    if (A.flags & EMGPU_FLAG_NO_RESAMPLE) nact = 0;
    for (int b = 0; b < nact; b++) {
        const EvStream E = s_ev[b];
        const uint32_t var0 = (uint32_t)E.var1 - 1u, kd = E.kdyn;
        const uint4 rh = rng.block(EMGPU_SEC_RES, var0, Tu >> 3), rl = rng.block(EMGPU_SEC_RES_LO, var0, Tu >> 3);
        uint32_t x = 0u;
#pragma unroll
        for (int j = 0; j < 8; j++) x = ((Tu & 7u) == (uint32_t)j) ? split_draw(rh, rl, j) : x;
        if (clamp32(x) < E.R) {
            const uint32_t bin1 = kd != 0xFFu ? ((curp >> (8u * kd)) & 0xFFu) : (uint32_t)((b < 8 ? SW.sb_lo >> (8 * b) : SW.sb_hi >> (8 * (b - 8))) & 0xFFull);
            float v = 0.f;
            if (E.nb == 0u || (A.flags & EMGPU_FLAG_NO_DEDISC)) v = (float)bin1;
            else if (bin1 != (uint32_t)E.zero) v = ev_draw(rng.c0, rng.c1, rng.attempt, rng.k0, rng.k1, bnd, var0, Tu, bin1 - 1u, E.boff);
            SW.S.emit(Tu, var0 + 1u, bin1, v);
        }
    }
    if (!(A.flags & EMGPU_FLAG_NO_TERMINATOR)) SW.S.emit(Tu, 0u, 0u, 0.f);
    if (valid) {
        A.ev_count[i] = SW.S.count;
        if (SW.S.count > (uint32_t)A.event_cap) atomicOr(A.status, 2u);
    }
}

} // namespace emgpu
