#pragma once
// emgpu_kernels_step2.h -- (the kernel; its instances are spread over emgpu_kernels_step2.hip and emgpu_kernels_step2b.hip so that they compile side by side)
// k_dbn_step2 -- the per-timestep DBN (dbn_sample.m:65-93: dependent-branch models such as
// cor_v1 and the glider family, and EMGPU_TRANSITION_PER_STEP) with dense output, built like
// k_uncor_fast: one lane = one trajectory, 8 seconds per loop iteration, packed 16-bit compares
// on the primary (high) halfwords, MSB-first flag streams, wave-cooperative dediscretize.
//
// What differs from the fast-branch kernel: a variable's CPT column changes every second with the
// dynamic state (asub2ind.m:13-14 as strides over the current and the freshly drawn bins), so the
// column is fetched per draw with ONE 16-byte gather through L1/L2 (a buffer resource, byte offsets): its packed-compare form
// (EmgpuPlan::d_poffpk: three T' pairs + the nibble map by fired count), whatever the column's width.
// The columns of one dependency level are gathered together, the next second's level-0 columns as soon as this second's
// level 0 is decided.  The secondary (low) halfword block of a variable is generated only at a second where
// some lane of the wave met a tie between a draw's high halfword and a threshold's (p = 2^-16 per
// compare); the draw is then redone with the full 32 bits, in place, because later seconds depend on it.
// Interior blocks run unguarded; the first block of a trajectory and a partial last one take a rolled loop on full draws.
// Bound: VALU issue (Philox + ~16 instructions per draw, 84 % of the cycles on cor_v1) and one exposed L1/L2 round trip per second.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "emgpu_coop.h"
#include "emgpu_device.h"
#include "emgpu_events.h"
#include "emgpu_init_karg.h"
#include "emgpu_launch.h"

namespace emgpu {


struct Step2Args {
    uint32_t Rk[EMGPU_MAX_ND];   // resample hit threshold of dynamic variable k (0 = rate 0), < 0xFFFF0000
    uint32_t slot[EMGPU_MAX_ND]; // output row of dynamic variable k
    uint32_t RR1[EMGPU_MAX_ND];  // (Rk >> 16) + 1 in both halfwords: the packed resample compare of a whole block
};

// Dependency level of (t+1) node k among the dynamic variables: 0 when none of its parents is another (t+1) node, else one more than
// the deepest such parent (NEW: bit 4k+q <=> the (t+1) node of q is a parent of k, q < k).  The columns of one level are fetched
// together: their round trips through L1/L2 overlap instead of following each other.
template <uint32_t NEW>
constexpr int s2_level(int k) {
    int l = 0;
    for (int q = 0; q < k; q++)
        if ((NEW >> (4 * k + q)) & 1u) { const int lq = s2_level<NEW>(q) + 1; l = lq > l ? lq : l; }
    return l;
}

template <uint32_t CUR, uint32_t NEW>
constexpr bool s2_pre(int k) {
    if (s2_level<NEW>(k) != 0) return false;
    for (int q = 0; q < 4; q++)
        if (((CUR >> (4 * k + q)) & 1u) && s2_level<NEW>(q) != 0) return false;
    return true;
}

constexpr uint32_t kSelBase2 = 0x0c0c0c00u; // v_perm_b32 selector: bytes 1-3 zero, byte 0 <- table[borrows]

// ---- the packed compare of an interior second (EmgpuPlan::d_poffpk) ------------------------------------------------------
// One draw against a column's T' pairs: x_h = the half ODD of w goes to BOTH halves of a packed subtract, each against its own
// threshold; d = min(sat(x_h - T'), 2) is 0 not fired, 1 the low halfword decides, 2 fired; the pairs are added up and the two
// halves folded: the result is 2 * (thresholds fired), odd exactly when the draw needs its low halfword.  Six (NW = 3) or four
// (NW = 2: columns of at most 3 thresholds; T'3 = 0xFFFF never fires) thresholds in 3 NW + 1 instructions, no carry, no VCC,
// no wait states (the carry chain: 3 per threshold plus a min per threshold for the tie).
template <bool ODD, int NW>
__device__ __forceinline__ uint32_t pk_fired2(uint32_t w, uint32_t t01, uint32_t t23, uint32_t t45) {
    uint32_t d0, d1, d2 = 0u, acc;
    if (ODD) {
        asm("v_pk_sub_u16 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] clamp" : "=v"(d0) : "v"(w), "v"(t01));
        asm("v_pk_sub_u16 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] clamp" : "=v"(d1) : "v"(w), "v"(t23));
        if (NW == 3) asm("v_pk_sub_u16 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] clamp" : "=v"(d2) : "v"(w), "v"(t45));
    } else {
        asm("v_pk_sub_u16 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] clamp" : "=v"(d0) : "v"(w), "v"(t01));
        asm("v_pk_sub_u16 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] clamp" : "=v"(d1) : "v"(w), "v"(t23));
        if (NW == 3) asm("v_pk_sub_u16 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] clamp" : "=v"(d2) : "v"(w), "v"(t45));
    }
    asm("v_pk_min_u16 %0, %0, 2 op_sel_hi:[1,0]" : "+v"(d0));
    asm("v_pk_min_u16 %0, %0, 2 op_sel_hi:[1,0]" : "+v"(d1));
    asm("v_pk_add_u16 %0, %1, %2" : "=v"(acc) : "v"(d0), "v"(d1));
    if (NW == 3) {
        asm("v_pk_min_u16 %0, %0, 2 op_sel_hi:[1,0]" : "+v"(d2));
        asm("v_pk_add_u16 %0, %0, %1" : "+v"(acc) : "v"(d2));
    }
    uint32_t s;
    asm("v_add_u32_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(s) : "v"(acc));
    return s;
}
// The same decision for a column of at most 3 thresholds in its PLAIN form {H0, H1, H2, map} (EmgpuPlan::d_poffpk): a_t = H_t - x_h with plain
// subtracts; fired <=> a_t < 0, tie <=> a_t == 0 (x_h = 0 against H = 0 included).  Returns the bin's bit offset in the map (7 * fired);
// tie receives min(a_t) as unsigned: 0 exactly on a tie.
template <bool ODD>
__device__ __forceinline__ uint32_t plain_fired7(uint32_t w, uint32_t h0, uint32_t h1, uint32_t h2, uint32_t &tie) {
    const uint32_t xh = ODD ? (w >> 16) : (w & 0xFFFFu);
    const uint32_t a0 = h0 - xh, a1 = h1 - xh, a2 = h2 - xh;
    uint32_t off;
    asm("v_add3_u32 %0, %1, %2, %3" : "=v"(off) : "v"(a0 >> 29), "v"(a1 >> 29), "v"(a2 >> 29));
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(tie) : "v"(a0), "v"(a1), "v"(a2));
    return off;
}
// s | (the half ODD of z): z carries a 1 in the halves whose x_h is 0 (a tie with any threshold whose high half is 0)
template <bool ODD>
__device__ __forceinline__ uint32_t or_half(uint32_t s, uint32_t z) {
    uint32_t r;
    if (ODD) asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(s), "v"(z));
    else asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(s), "v"(z));
    return r;
}
// bin * stride + acc, everything in vector registers (no scalar operand: no wait states to respect)
__device__ __forceinline__ uint32_t mad24v(uint32_t bin, uint32_t stride, uint32_t acc) {
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(bin), "v"(stride), "v"(acc));
    return r;
}

// The rare exact redo of one draw, out of line so that the 32 copies of the hot per-second body stay small:
// the secondary block is generated and the compare repeated on the full 32-bit draw (select_random.m:19-20).
__device__ __attribute__((noinline)) uint32_t exact_borrows(uint32_t c0, uint32_t c1, uint32_t attempt, uint32_t k0, uint32_t k1,
                                                            uint32_t hi_word, uint32_t tvar, uint32_t g8, uint32_t j,
                                                            uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3, uint32_t t4, uint32_t t5) {
    const Rng rng{c0, c1, attempt, k0, k1};
    const uint4 tl = rng.block(EMGPU_SEC_TRANS_LO, tvar, g8);
    const uint32_t q = j >> 1;
    const uint32_t wl = q == 0 ? tl.x : (q == 1 ? tl.y : (q == 2 ? tl.z : tl.w));
    const uint32_t hi = (j & 1u) ? (hi_word & 0xFFFF0000u) : (hi_word << 16), lo = (j & 1u) ? (wl >> 16) : (wl & 0xFFFFu);
    const uint32_t x = clamp32(hi | lo);
    return (x < t0 ? 1u : 0u) + (x < t1 ? 1u : 0u) + (x < t2 ? 1u : 0u) + (x < t3 ? 1u : 0u) + (x < t4 ? 1u : 0u) + (x < t5 ? 1u : 0u);
}
__device__ __attribute__((noinline)) uint32_t exact_hit(uint32_t c0, uint32_t c1, uint32_t attempt, uint32_t k0, uint32_t k1,
                                                        uint32_t hi_word, uint32_t ivar, uint32_t g8, uint32_t j, uint32_t R) {
    const Rng rng{c0, c1, attempt, k0, k1};
    const uint4 rl = rng.block(EMGPU_SEC_RES_LO, ivar, g8);
    const uint32_t q = j >> 1;
    const uint32_t wl = q == 0 ? rl.x : (q == 1 ? rl.y : (q == 2 ? rl.z : rl.w));
    const uint32_t hi = (j & 1u) ? (hi_word & 0xFFFF0000u) : (hi_word << 16), lo = (j & 1u) ? (wl >> 16) : (wl & 0xFFFFu);
    return clamp32(hi | lo) < R ? 1u : 0u;                                                  // resample_events.m:24
}

// WMODE: 4 / 8 = every variable's columns are 4 / 8 words wide, 0 = decided per variable at run time, 16 + m = variable k is 4 words wide
// iff bit k of m is set (s2_w4).  A width left to run time is a wave-uniform branch per draw with both forms of the draw behind it.
// REG ("regular"): exactly ND dynamic variables, all with a resample rate > 0.  The specialised
// instances drop the wave-uniform tests and the code behind them (cor_v1: 25.2 -> 20.8 ms).
// CUR / NEW: which dynamic variables are parents of which (t+1) node (bit 4k+q; step_parent_masks): an instance built for a
// model's masks multiplies only the strides that exist (cor_v1: 6 of 22) and fetches the columns of a dependency level together.
// FRZ: the FAST branch of dbn_sample.m:95-166 on this kernel -- the parent configuration of every transition is frozen at the
// initial state (the column of a variable never changes along a trajectory).  For the fast-branch models k_uncor_fast does not
// take (four dynamic variables: littoral_cor_v1); the per-second gathers then hit the same line every time.
// EV: the event list as well (emgpu_events.h): what UncorEncounterModel.sample / dbn_hierarchical_sample return for these models.
template <int WMODE>
__device__ __forceinline__ bool s2_w4(const EmgpuPlan &P, int k) {
    return WMODE == 4 || (WMODE >= 16 ? (((WMODE - 16) >> k) & 1) != 0 : (WMODE == 0 && P.d_pw[k] == 4));
}



// EMGPU_DEBUG_EXTRA_LDS: bytes of unused dynamic LDS per workgroup (tools/occupancy_probe.sh: what does a kernel lose with one wave less?)
inline size_t step2_extra_lds() {
    static const int extra = getenv("EMGPU_DEBUG_EXTRA_LDS") ? atoi(getenv("EMGPU_DEBUG_EXTRA_LDS")) : 0;
    return (size_t)extra;
}

// The dense 16-variable instances take a fourth wave: self-contained requests (coop_dedisc_sc: LDS rows of 36 words instead of 44, four
// workgroups per CU) and the initial network through the kernel-argument segment (155 -> <= 128 registers).  Same box, tools/ab_bench.sh:
// cor_v1 `[cor] w4` 11.42 -> 10.98 ms, cor_v2p1_like `[cor] w8` 14.72 -> 14.48, littoral_cor_v1 `[frozen] w4` 11.02 -> 10.65, balloon_v1 /
// weatherballoon_v1 on the run-time-width instance 7.29 -> 6.13 / 7.48 -> 6.49.  (With the owner's bin looked up in registers -- a select
// over the variables -- instead of in its own, not yet used, result slots, only the first and the last of these gained.)
constexpr bool step2_sc_form(int NI, int ND, int WMODE, bool FRZ, int EV) {
    return ND == 4 && NI == 16 && EV == 0;
}

// EV: 0 the dense trace; 1 the event list as well (result slots + a row loop per lane, emgpu_events.h); 2 the list ALONE, its rows built
// by the wave ("ROWS BY THE WAVE": no result slots, no fill)
template <int NI, int ND, int WMODE, bool REG, uint32_t CUR, uint32_t NEW, bool FRZ = false, int EV = 0>
__global__ void __launch_bounds__(256, ((ND == 4 && !step2_sc_form(NI, ND, WMODE, FRZ, EV)) || EV == 1) ? 3 : 4) k_dbn_step2(const EmgpuPlan P, const EmgpuRun A, const Step2Args F) {
    static_assert(!FRZ || NEW == 0u, "a fast-branch model has no (t+1) parents");
    // the instances built for a model family's parent masks are only launched with both dense outputs (launch_masked): no null tests at the stores
    constexpr bool kBoth = !EV && CUR != 0x0777u && CUR != 0xFFFFu;
    // LBK: the workers look a request's bin up in the owner's LDS row (emgpu_coop.h).  The dense 16-variable instances carry it in the request
    // instead (coop_dedisc_sc): their rows are 36 words instead of 44, FOUR workgroups fit a CU's LDS instead of three, and with the initial
    // network read through the kernel-argument segment they have the registers for it
    constexpr bool LBK = !step2_sc_form(NI, ND, WMODE, FRZ, EV);
    __shared__ CoopLds<ND, LBK> s_wave[4];
    __shared__ double s_bnd[ND][16];
    const int tid = threadIdx.x, lane = tid & 63;
    CoopLds<ND, LBK> &W = s_wave[tid >> 6];
    const int64_t i = (int64_t)blockIdx.x * 256 + tid;
    const bool valid = i < A.n; // lanes past the end stay alive: they serve as workers for their wave
    const uint64_t gidx = A.first_index + (uint64_t)i;
    Rng rng{(uint32_t)gidx, (uint32_t)(gidx >> 32), 0u, (uint32_t)A.seed, (uint32_t)(A.seed >> 32)};
    const int T = A.T;
#pragma unroll
    for (int k = 0; k < ND; k++) // k stays a compile-time index into the plan (a per-lane index would force the kernarg struct into scratch)
        if ((tid >> 4) == k) {
            const int q = tid & 15;
            s_bnd[k][q] = (k < P.nd && q < (int)P.d_nb[k]) ? P.bnd[P.d_boff[k] + q] : 0.0;
        }

    uint32_t cur1[ND], basecol[ND];
    float cval[ND];
    EvPlan E{};
    EvState S{};
    EvStateW SW{};
    EvStream *const s_evs = ev_rows_stream_lds<EV == 2>();
    // the rows' request queue: the cooperative dediscretize's own (idle in this form, 254 requests per round: the 3-variable instances stay
    // within 40 KB of LDS and 128 registers, four waves per SIMD), or 512 requests in LDS of its own (the 4-variable instances)
    constexpr int KQ = ND == 4 ? kEvRowsQueue<4> : 254;
    uint16_t *const s_evq = ND == 4 ? ev_rows_queue_lds<EV == 2 && ND == 4, ND>(tid >> 6) : reinterpret_cast<uint16_t *>(W.queue);
    static_assert(sizeof(W.queue) >= (254 + 2) * sizeof(uint16_t), "rows queue");
    {
        int bin[NI];
        double val[NI];
#pragma unroll
        for (int p = 0; p < NI; p++) { bin[p] = 0; val[p] = 0.0; }
        int32_t attempts_used;
        if constexpr (!LBK) attempts_used = init_network_karg<NI>((KargPlan)__builtin_amdgcn_kernarg_segment_ptr(), A, rng, bin, val);
        else attempts_used = init_network<NI>(P, A, rng, bin, val);
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
        // The event-list set-up looks plan arrays up by run-time positions (P.i_zero[P.a_pos[b]], P.d_ivar[P.d_emit[e]]).  Done on the by-value
        // kernel argument itself that made the 16-variable instances keep a private copy of the whole plan (2.6 KB of scratch per lane,
        // every later P.x a scratch load, the table's buffer resource no longer provably uniform: a waterfall loop round every gather --
        // cor_v1's lists at a third of the dense trace's rate); read through the kernel-argument segment the plan stays where it is.
        const EmgpuPlan &Pk = *(const EmgpuPlan *)__builtin_amdgcn_kernarg_segment_ptr();
        if constexpr (EV == 1) {
            E = ev_plan_of<NI, ND>(Pk);
            S = ev_state_of<NI, ND>(Pk, A, bin, valid, i);
        }
        if constexpr (EV == 2) {
            ev_wide_plan<ND>(Pk, s_evs);          // (visible after the barrier below)
            SW = ev_state_w_of<NI, ND>(Pk, A, bin, valid, i);
            ev_rows_publish_static<ND>(W, lane, SW);
            coop_publish_gidx<ND>(W, lane, gidx);
        }
#pragma unroll
        for (int k = 0; k < ND; k++) {
            cur1[k] = 1u; cval[k] = 0.f; basecol[k] = 0u;
            if (k < P.nd) {
                cur1[k] = (uint32_t)pick<NI>(bin, P.d_ipos[k]) + 1u;
                cval[k] = (float)pick<NI>(val, P.d_ipos[k]);
                uint32_t b = 0;
#pragma unroll
                for (int p = 0; p < NI; p++) b += P.d_stride_static[k][p] * (uint32_t)bin[p];
                // the dynamic parents are added as stride * (1-based bin): take the "-1"s out here
#pragma unroll
                for (int q = 0; q < ND; q++) b -= P.d_stride_cur[k][q] + P.d_stride_new[k][q];
                basecol[k] = b;
            }
        }
    }
    W.attempt[lane] = rng.attempt;
    __syncthreads();
    // one 16-byte group of a padded column, gathered through L1/L2 (the tables are small)
    const uint32_t *__restrict__ gtab = P.pthr;
    // a buffer resource over the table: the gather's address is a 32-bit byte offset (one vector instruction, buffer_load ... offen)
    // instead of a 64-bit pointer per lane
    typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(gtab), 0, -1, 0x00020000);
    auto load4 = [&](uint32_t byte_off) {
        const v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 0);
        return make_uint4(v.x, v.y, v.z, v.w);
    };
    // columns are addressed in BYTES: the strides (scalar registers) and the lane's base are scaled by the column width once, so a
    // gather's address is the multiply-adds over the parents and nothing else.  full_of: the same column in the 8-word table.
    uint32_t wbytes[ND];
#pragma unroll
    for (int k = 0; k < ND; k++) {
        wbytes[k] = 16u; // the packed-compare form: 16 bytes per column (EmgpuPlan::d_poffpk)
        basecol[k] = P.d_poffpk[k] * 4u + basecol[k] * 16u;
    }
    // the strides of the parents that exist, in bytes, in vector registers (mad24v)
    uint32_t svc[ND][ND], svn[ND][ND];
#pragma unroll
    for (int k = 0; k < ND; k++)
#pragma unroll
        for (int q = 0; q < ND; q++) {
            svc[k][q] = svn[k][q] = 0u;
            if ((CUR >> (4 * k + q)) & 1u) asm volatile("v_mov_b32 %0, %1" : "=v"(svc[k][q]) : "s"(P.d_stride_cur[k][q] * 16u));
            if (q < k && ((NEW >> (4 * k + q)) & 1u)) asm volatile("v_mov_b32 %0, %1" : "=v"(svn[k][q]) : "s"(P.d_stride_new[k][q] * 16u));
        }
    uint32_t ivs[ND];
#pragma unroll
    for (int k = 0; k < ND; k++) ivs[k] = P.d_ivar[k];
    uint32_t frz[ND];   // FRZ: the initial bins, the only "current" bins a column ever sees (dbn_sample.m:110-135)
#pragma unroll
    for (int k = 0; k < ND; k++) frz[k] = cur1[k];

    // dependency levels as compile-time constants (a constexpr call with the loop variable is only folded after unrolling, too late
    // for the register allocator: the level loop would index its arrays dynamically)
    constexpr int kLev[4] = {s2_level<NEW>(0), s2_level<NEW>(1), s2_level<NEW>(2), s2_level<NEW>(3)};
    constexpr int kMaxLev = kLev[0] > kLev[1] ? (kLev[0] > kLev[2] ? (kLev[0] > kLev[3] ? kLev[0] : kLev[3]) : (kLev[2] > kLev[3] ? kLev[2] : kLev[3]))
                                             : (kLev[1] > kLev[2] ? (kLev[1] > kLev[3] ? kLev[1] : kLev[3]) : (kLev[2] > kLev[3] ? kLev[2] : kLev[3]));
    // level-0 nodes all of whose dynamic parents are level-0 nodes (through their current bins)
    constexpr bool kPre[4] = {s2_pre<CUR, NEW>(0), s2_pre<CUR, NEW>(1), s2_pre<CUR, NEW>(2), s2_pre<CUR, NEW>(3)};
    const int G4 = (T + 3) >> 2, G8 = (T + 7) >> 3;
    for (int g8 = 0; g8 < G8; g8++) {
        uint4 th[ND];
        uint32_t pbA[ND], pbB[ND], hit8[ND], chg8[ND], zer8[ND]; // flag streams MSB-first: bit 7-j <-> second j
        uint32_t prevp = 0u;                                     // EV: the bins the block starts from, one byte per variable
        if constexpr (EV) {
#pragma unroll
            for (int k = 0; k < ND; k++) prevp |= cur1[k] << (8 * k);
        }
        // seconds of this block that are draws at all (1 <= c < T), as an MSB-first mask
        uint32_t live8 = 0u;
#pragma unroll
        for (int j = 0; j < 8; j++) live8 |= (8 * g8 + j >= 1 && 8 * g8 + j < T) ? (0x80u >> j) : 0u;
#pragma unroll
        for (int k = 0; k < ND; k++) {
            th[k] = make_uint4(0, 0, 0, 0);
            pbA[k] = pbB[k] = hit8[k] = chg8[k] = zer8[k] = 0u;
            if (!REG && k >= P.nd) continue;
            th[k] = rng.block(EMGPU_SEC_TRANS, P.d_tvar[k], (uint32_t)g8);
            if (REG || F.Rk[k] != 0u) {
                // resample_events.m:24 for the whole block: the Bernoulli does not depend on the state, so its eight seconds are
                // decided two per instruction from the high halfwords (0 no hit, 1 tie, 2 hit; k_uncor_fast does the same)
                const uint4 rh = rng.block(EMGPU_SEC_RES, P.d_ivar[k], (uint32_t)g8);
                uint32_t hitA = 0u;
#pragma unroll
                for (int p2 = 0; p2 < 4; p2++) {
                    uint32_t u;
                    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(u) : "s"(F.RR1[k]), "v"(word_of(rh, p2)));
                    asm("v_pk_min_u16 %0, %0, 2 op_sel_hi:[1,0]" : "+v"(u));
                    hitA = p2 ? ((hitA << 2) | u) : u;
                }
                hit8[k] = (hitA & 0xAAu) | ((hitA >> 17) & 0x55u);
                if (__ballot((hitA & 0x00550055u) != 0u) != 0ull) {   // some lane ties with R's high half: this variable's block with 32 bits
                    uint32_t h = 0u;
#pragma unroll 1
                    for (int j = 0; j < 8; j++)
                        h = (h << 1) | exact_hit(rng.c0, rng.c1, rng.attempt, rng.k0, rng.k1, word_of(rh, j >> 1), P.d_ivar[k], (uint32_t)g8, (uint32_t)j, F.Rk[k]);
                    hit8[k] = h;
                }
                hit8[k] &= live8;
            }
        }
        // the address of (t+1) node k's column: asub2ind.m:13-14 over the current and the new bins of its parents
        auto column_of = [&](int k, const uint32_t (&nb1)[ND]) {
            uint32_t col = basecol[k];
#pragma unroll
            for (int q = 0; q < ND; q++)
                if ((CUR >> (4 * k + q)) & 1u) col = mad24v(FRZ ? frz[q] : cur1[q], svc[k][q], col);
#pragma unroll
            for (int q = 0; q < k; q++)
                if ((NEW >> (4 * k + q)) & 1u) col = mad24v(nb1[q], svn[k][q], col);
            return col;
        };
        // the same column in the padded table of full thresholds (4 or 8 words per column)
        auto full_of = [&](int k, uint32_t colpk) {
            const bool w4 = s2_w4<WMODE>(P, k);
            return (colpk - P.d_poffpk[k] * 4u) * (w4 ? 1u : 2u) + P.d_poff[k] * 4u;
        };
        if (8 * g8 + 7 < T) {
            // ---- full block: every second is a draw, nothing is guarded -- but second 0 of the trajectory (block 0), which is the
            // initial state and no draw (dbn_sample.m:77 starts at t = 2): a wave-uniform branch around that one second.
            // A level-0 node whose parents are level-0 nodes' current bins only (kPre) knows its NEXT second's column as soon as
            // this second's level 0 is decided: that gather is issued together with this second's level-1 gathers, so a second
            // exposes one round trip less through L1/L2.
            uint4 pre[ND];
            uint32_t precol[ND];
#pragma unroll
            for (int k = 0; k < ND; k++) { pre[k] = make_uint4(0, 0, 0, 0); precol[k] = 0u; }
            uint32_t cur_in[ND];   // the bins the block starts from (the "changed" stream of second 0 compares with them)
#pragma unroll
            for (int k = 0; k < ND; k++) cur_in[k] = cur1[k];
            // a 1 in every half of the block's draws that is 0: such a draw ties with any threshold whose high half is 0 (it has no T')
            uint4 zt[ND];
#pragma unroll
            for (int k = 0; k < ND; k++) {
                zt[k] = make_uint4(0u, 0u, 0u, 0u);
                if (s2_w4<WMODE>(P, k)) continue;   // (the plain form sees x_h = 0 against H = 0 as the tie it is)
                uint32_t z[4];
#pragma unroll
                for (int p2 = 0; p2 < 4; p2++) asm("v_pk_sub_u16 %0, 1, %1 op_sel_hi:[0,1] clamp" : "=v"(z[p2]) : "v"(word_of(th[k], p2)));
                zt[k] = make_uint4(z[0], z[1], z[2], z[3]);
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                uint32_t nb1[ND];
#pragma unroll
                for (int k = 0; k < ND; k++) nb1[k] = (j == 0) ? cur1[k] : 1u;
                const bool draw = j > 0 || g8 != 0;   // wave-uniform; folded for j > 0
#pragma unroll
                for (int lev = 0; lev <= kMaxLev; lev++) {
                    if (draw) {
                    uint4 ca[ND];   // live only across this level's gathers and draws
                    uint32_t sel[ND], dmin[ND], colv[ND];
#pragma unroll
                    for (int k = 0; k < ND; k++) { ca[k] = make_uint4(0, 0, 0, 0); sel[k] = 0u; dmin[k] = 0xFFFFFFFFu; colv[k] = 0u; }
                    // ---- the columns of this level, one 16-byte gather each
#pragma unroll
                    for (int k = 0; k < ND; k++) {
                        if (kLev[k] != lev || (!REG && k >= P.nd)) continue;
                        if (lev == 0 && j > 0 && kMaxLev >= 1 && kPre[k]) { colv[k] = precol[k]; ca[k] = pre[k]; continue; }   // requested a second ago
                        colv[k] = column_of(k, nb1);
                        ca[k] = load4(colv[k]);
                    }
                    // ---- the draws of this level from the high halfwords (dbn_sample.m:77): twice the number of thresholds that fired,
                    // bit 0 raised when the low halfword is needed
                    uint32_t tlev = 0u;
#pragma unroll
                    for (int k = 0; k < ND; k++) {
                        if (kLev[k] != lev || (!REG && k >= P.nd)) continue;
                        const uint32_t wt = word_of(th[k], j >> 1), wz = word_of(zt[k], j >> 1);
                        const uint4 a = ca[k];
                        if (s2_w4<WMODE>(P, k)) {   // the plain form: sel = 1 on a tie (the map offset goes straight to the lookup)
                            uint32_t tie;
                            const uint32_t off = (j & 1) ? plain_fired7<true>(wt, a.x, a.y, a.z, tie) : plain_fired7<false>(wt, a.x, a.y, a.z, tie);
                            nb1[k] = __builtin_amdgcn_ubfe(a.w, off, 4u);
                            sel[k] = tie == 0u ? 1u : 0u;
                        } else {
                            const uint32_t s2 = (j & 1) ? pk_fired2<true, 3>(wt, a.x, a.y, a.z) : pk_fired2<false, 3>(wt, a.x, a.y, a.z);
                            sel[k] = (j & 1) ? or_half<true>(s2, wz) : or_half<false>(s2, wz);
                            nb1[k] = __builtin_amdgcn_ubfe(a.w, sel[k] << 1, 4u);   // nibble (fired) of the column's map; overwritten below on a tie
                        }
                        tlev |= sel[k];
                    }
                    // one tie test per level; the draws that tied (in some lane) are repeated on the full 32-bit draw, out of line
                    if (__ballot((tlev & 1u) != 0u) != 0ull) {
#pragma unroll
                        for (int k = 0; k < ND; k++) {
                            if (kLev[k] != lev || (!REG && k >= P.nd)) continue;
                            if (__ballot((sel[k] & 1u) != 0u) == 0ull) continue;
                            const uint32_t cf = full_of(k, colv[k]);   // the full thresholds of the column, from the padded table
                            if (s2_w4<WMODE>(P, k)) {
                                const uint4 fa = load4(cf);
                                const uint32_t b = exact_borrows(rng.c0, rng.c1, rng.attempt, rng.k0, rng.k1, word_of(th[k], j >> 1), P.d_tvar[k], (uint32_t)g8, (uint32_t)j,
                                                                 fa.x, fa.y, fa.z, 0u, 0u, 0u);
                                nb1[k] = __builtin_amdgcn_perm(0u, fa.w, kSelBase2 + b);
                            } else {
                                const uint4 fa = load4(cf), fb = load4(cf + 16u);
                                const uint32_t b = exact_borrows(rng.c0, rng.c1, rng.attempt, rng.k0, rng.k1, word_of(th[k], j >> 1), P.d_tvar[k], (uint32_t)g8, (uint32_t)j,
                                                                 fa.x, fa.y, fa.z, fa.w, fb.x, fb.y);
                                nb1[k] = __builtin_amdgcn_perm(fb.w, fb.z, kSelBase2 + b);
                            }
                        }
                    }
                    }   // draw
                    if (lev == 0 && j < 7 && kMaxLev >= 1) {
#pragma unroll
                        for (int k = 0; k < ND; k++) {
                            if (!kPre[k] || (!REG && k >= P.nd)) continue;
                            uint32_t col = basecol[k];     // the column of second j + 1: the new bins are its current ones
#pragma unroll
                            for (int q = 0; q < ND; q++)
                                if ((CUR >> (4 * k + q)) & 1u) col = mad24v(nb1[q], svc[k][q], col);
                            precol[k] = col;
                            pre[k] = load4(col);
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < ND; k++) {
                    if (!REG && k >= P.nd) continue;
                    cur1[k] = nb1[k];                                                               // map back, dbn_sample.m:82
                    if (j < 4) pbA[k] = j ? (pbA[k] | (cur1[k] << (8 * (j & 3)))) : cur1[k];
                    else pbB[k] = (j & 3) ? (pbB[k] | (cur1[k] << (8 * (j & 3)))) : cur1[k];
                }
            }
            // the block's "changed" and "zero bin" streams from the packed bins (MSB-first: bit 7 - j <-> second j): a bin differs from
            // the second before it when the XOR's low nibble is not 0 (bins are < 16: + 0x0F carries into bit 4); it is the zero bin when
            // the XOR with that bin is 0 (+ 0x7F leaves bit 7 clear).  One multiply per word gathers the bits (byte_bit_stream).
#pragma unroll
            for (int k = 0; k < ND; k++) {
                if (!REG && k >= P.nd) continue;
                const uint32_t prevA = (pbA[k] << 8) | cur_in[k], prevB = __builtin_amdgcn_alignbit(pbB[k], pbA[k], 24);
                chg8[k] = byte_bit_stream<4>((pbA[k] ^ prevA) + 0x0F0F0F0Fu, (pbB[k] ^ prevB) + 0x0F0F0F0Fu);
                const uint32_t zz = (uint32_t)P.d_zero[k] * 0x01010101u;
                zer8[k] = ~byte_bit_stream<7>((pbA[k] ^ zz) + 0x7F7F7F7Fu, (pbB[k] ^ zz) + 0x7F7F7F7Fu) & 0xFFu;
            }
        } else {
            // ---- the first block of a trajectory (second 0 is the initial state, not a draw) and a partial last block: one rolled
            // loop over the seconds, every draw decided from its full 32 bits (no tie logic); same answers, 1 block in 30
            uint4 tl[ND];
#pragma unroll
            for (int k = 0; k < ND; k++) {
                tl[k] = make_uint4(0, 0, 0, 0);
                if (REG || k < P.nd) tl[k] = rng.block(EMGPU_SEC_TRANS_LO, P.d_tvar[k], (uint32_t)g8);
            }
#pragma unroll 1
            for (int j = 0; j < 8; j++) {
                const int c = 8 * g8 + j; // absolute event time == column produced
                if (c >= 1 && c < T) {    // wave-uniform
                    uint32_t nb1[ND];
#pragma unroll
                    for (int k = 0; k < ND; k++) nb1[k] = 1u;
#pragma unroll
                    for (int lev = 0; lev <= kMaxLev; lev++) {
#pragma unroll
                        for (int k = 0; k < ND; k++) {
                            if (kLev[k] != lev || (!REG && k >= P.nd)) continue;
                            const bool w4 = s2_w4<WMODE>(P, k);
                            const uint32_t col = full_of(k, column_of(k, nb1));   // the plain thresholds of the padded table
                            const uint4 a = load4(col);
                            uint4 b = make_uint4(0, 0, 0, 0);
                            if (!w4) b = load4(col + 16u);
                            const uint32_t wt = word_of(th[k], j >> 1), wl = word_of(tl[k], j >> 1);
                            const uint32_t x = clamp32((j & 1) ? ((wt & 0xFFFF0000u) | (wl >> 16)) : ((wt << 16) | (wl & 0xFFFFu)));
                            uint32_t borrows = (x < a.x ? 1u : 0u) + (x < a.y ? 1u : 0u) + (x < a.z ? 1u : 0u);   // select_random.m:19-20
                            if (!w4) borrows += (x < a.w ? 1u : 0u) + (x < b.x ? 1u : 0u) + (x < b.y ? 1u : 0u);
                            nb1[k] = __builtin_amdgcn_perm(w4 ? 0u : b.w, w4 ? a.w : b.z, kSelBase2 + borrows);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < ND; k++) {
                        chg8[k] = (chg8[k] << 1) | (nb1[k] != cur1[k] ? 1u : 0u);
                        zer8[k] = (zer8[k] << 1) | (nb1[k] == (uint32_t)P.d_zero[k] ? 1u : 0u);
                        if (REG || k < P.nd) cur1[k] = nb1[k];                                      // map back, dbn_sample.m:82
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < ND; k++) { chg8[k] += chg8[k]; zer8[k] += zer8[k]; }
                }
#pragma unroll
                for (int k = 0; k < ND; k++) {
                    const uint32_t b = (c < T) ? (cur1[k] << (8 * (j & 3))) : 0u;
                    if (j < 4) pbA[k] |= b; else pbB[k] |= b;
                }
            }
        }
        uint32_t need = 0u, kind = 0u, fill8[ND];
#pragma unroll
        for (int k = 0; k < ND; k++) {
            const uint32_t n8 = (REG || k < P.nd) ? ((hit8[k] | chg8[k]) & ~zer8[k] & 0xFFu) : 0u; // a dediscretize draw is due (REG: P.nd == ND)
            fill8[k] = n8 | (chg8[k] & 0xFFu);                                              // ... or 0 on a change into the zero bin
            need |= n8 << (8 * k);
            kind |= (chg8[k] & 0xFFu) << (8 * k);
        }
        if (!valid) need = 0u;
        if constexpr (EV == 1) { if (A.flags & EMGPU_FLAG_NO_DEDISC) need = 0u; }   // (plain dbn_sample.m: no draw is due)
        EMGPU_COUNT(5, lane, 1);
        if constexpr (EV == 2) {
            uint32_t hitp = 0u;
#pragma unroll
            for (int k = 0; k < ND; k++) hitp |= (hit8[k] & 0xFFu) << (8 * k);
            coop_publish_bins<ND>(W, lane, pbA, pbB);
            ev_rows_block_wide<ND, KQ>(W, s_evq, lane, s_evs, P.nact, SW, rng, P.bnd, g8, T, valid, hitp, kind, prevp, A, i);
        } else {
        if constexpr (LBK) {
            coop_zero_results<ND, LBK>(W, lane);
            coop_publish_bins<ND>(W, lane, pbA, pbB);
            coop_dedisc<ND, true, true>(W, lane, gidx, rng, g8, need, kind, pbA, pbB, ivs, s_bnd);   // dediscretize.m:39
        } else coop_dedisc_sc<ND, true>(W, lane, gidx, rng, g8, need, kind, pbA, pbB, ivs, s_bnd);
        if (!EV || A.dyn_bin != nullptr || A.dyn_val != nullptr)   // (an event-list call without the dense trace: no forward fill at all)
#pragma unroll
        for (int k = 0; k < ND; k++)
            if (REG || k < P.nd)
                coop_fill_store_msb<ND, LBK, kBoth>(W, lane, k, g8, T, G4, valid, fill8[k], cval[k], pbA[k], pbB[k],
                                              REG ? (uint32_t)ND : (uint32_t)P.nd, F.slot[k], (int64_t)blockIdx.x * 256, (uint32_t)tid, A.ld, A.dyn_bin, A.dyn_val);
        if constexpr (EV == 1) {
            uint32_t hitp = 0u;
#pragma unroll
            for (int k = 0; k < ND; k++) hitp |= (hit8[k] & 0xFFu) << (8 * k);
            ev_emit_block<ND, true>(W, lane, E, S, rng, P.bnd, g8, T, valid, hitp, kind, prevp, (A.flags & EMGPU_FLAG_NO_DEDISC) != 0);
        }
        }
        wave_sync();
    }
    if constexpr (EV) {
        uint32_t curp = 0u;
#pragma unroll
        for (int k = 0; k < ND; k++) curp |= cur1[k] << (8 * k);
        if constexpr (EV == 2) {
            ev_tail_wide<ND>(s_evs, P.nact, SW, rng, P.bnd, T, curp, A, valid, i);
        }
        else ev_tail<ND>(E, S, rng, P.bnd, T, curp, A, valid, i);
    }
}


// every parent / the full chain of dependencies: the instance any model can run on
constexpr uint32_t kCurAll3 = 0x0777u, kNewAll3 = 0x0310u, kCurAll4 = 0xFFFFu, kNewAll4 = 0x7310u;

// one instance, with or without the event list
#define EMGPU_S2_LAUNCH(NI_, ND_, W_, REG_, C_, N_, FRZ_)                                                                      \
    do {                                                                                                                       \
        if (A.ev_count != nullptr && step2_rows_by_wave(P, A)) hipLaunchKernelGGL((k_dbn_step2<NI_, ND_, W_, REG_, C_, N_, FRZ_, 2>), g, b, step2_extra_lds(), s, P, A, F); \
        else if (A.ev_count != nullptr) hipLaunchKernelGGL((k_dbn_step2<NI_, ND_, W_, REG_, C_, N_, FRZ_, 1>), g, b, step2_extra_lds(), s, P, A, F); \
        else hipLaunchKernelGGL((k_dbn_step2<NI_, ND_, W_, REG_, C_, N_, FRZ_, 0>), g, b, step2_extra_lds(), s, P, A, F);                        \
    } while (0)


// a list asked for alone: its rows are built by the wave (EMGPU_DEBUG_EVENT_ROWS=lane: tests keep the per-lane row loop reachable)
inline bool step2_rows_by_wave(const EmgpuPlan &P, const EmgpuRun &A) {
    static const char *rows_env = getenv("EMGPU_DEBUG_EVENT_ROWS");
    return A.dyn_bin == nullptr && A.dyn_val == nullptr && rows_env == nullptr && ev_plan_wide_ok(P, A);
}

// the instances built for the 3-variable families (emgpu_kernels_step2b.hip)
bool launch_masked3(const EmgpuPlan &P, const EmgpuRun &A, const Step2Args &F, hipStream_t s, uint32_t cur, uint32_t nw, const char **tag);

} // namespace emgpu
