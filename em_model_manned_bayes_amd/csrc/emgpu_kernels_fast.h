// emgpu_kernels_fast.h -- the benchmarked kernel (templates; launched from emgpu_kernels_fast.hip, the event-list forms from
// emgpu_kernels_fast_ev.hip): uncorrelated DBN, REFERENCE_AUTO semantics on a
// "fast-branch" model (dbn_sample.m:95-166: parent configuration frozen at the initial state),
// compact dense trace output.  One lane = one trajectory, 3 dynamic variables, 8 seconds per
// loop iteration.
//
// Work per trajectory and 8-second block
//   * 6 Philox4x32 calls: the PRIMARY (high) halfwords of the 24 transition draws
//     (dbn_sample.m:133,144) and 24 resample Bernoullis (resample_events.m:24) of the dynamic
//     variables.  Variables that are not dynamic cannot change the dense trace.
//     A draw is decided from its high 16 bits alone unless they tie with a threshold's high half
//     (p ~ 1e-4 per draw); only then are the SECONDARY (low) halfwords fetched and the 8 seconds
//     of that variable redone exactly -- the value drawn is the same 32-bit uniform either way.
//   * compares against register-resident quantile thresholds (select_random.m:17-20), two SECONDS per
//     instruction: the 16-bit primary halfwords of seconds 2p and 2p+1 share a Philox word, and packed
//     16-bit arithmetic (v_pk_sub_u16 clamp / v_pk_min_u16 / v_pk_add_u16) counts fired thresholds in
//     both halves at once; ties with a threshold's high half show up as an odd count.
//   * the rare dediscretize draws (dediscretize.m:39; ~0.3 per lane and block) are compacted
//     across the 64 lanes of the wave through LDS and computed by "worker" lanes: one Philox call
//     per wave serves them all instead of one divergent call per event.
//   * 2 x (4-byte + 16-byte) stores per variable: time-blocked SoA, 1 KiB contiguous per wave store.
// Bound: HBM writes (3635 B / trajectory) co-limited by the integer multiplies of Philox4x32
// (HISTORY.md section 5).  No MFMA: there is no contraction on this path.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "emgpu_coop.h"
#include "emgpu_device.h"
#include "emgpu_events.h"
#include "emgpu_init_karg.h"
#include "emgpu_launch.h"

namespace emgpu {

struct FastArgs {
    uint32_t Rk[3];   // resample hit threshold of dynamic variable k (0 = rate 0), < 0xFFFF0000
    uint32_t slot[3]; // output row of dynamic variable k
    uint32_t RR1[3];  // (Rk >> 16) + 1 in both halfwords: the packed resample compare (eight_seconds_pk)
};

#ifndef EMGPU_FAST_WAVES
#define EMGPU_FAST_WAVES 4 // waves per SIMD the register budget is set for (LDS allows 4 workgroups per CU)
#endif

// One compacted CPT column (EmgpuPlan::cthr): meff distinct thresholds, then the nibble map
// bin(n) = (map >> 4n) & 15 with n = #{t : x' >= threshold t}.  The kernel instance may be built for
// M >= meff thresholds: the extra ones repeat the last one (no new tie value).  The map is returned as a BYTE table indexed by
// the number of BORROWS b = M - n (what the compare chain counts), entries 0-3 in bml and 4-7 in
// bmh, so that one v_perm_b32 with selector kSelBase + b yields the 1-based bin.
constexpr uint32_t kSelBase = 0x0c0c0c00u; // v_perm_b32 selector: bytes 1-3 constant zero, byte 0 <- table[b]
template <int M>
__device__ __forceinline__ void load_cthr_full(uint32_t (&th)[M], const uint32_t *__restrict__ p, int meff) {
#pragma unroll
    for (int t = 0; t < M; t++) th[t] = p[t < meff ? t : meff - 1]; // the instance's extra thresholds repeat the last one
}
// The hot loop compares high halfwords only, so the registers hold two threshold halves each
// (threshold 2q in the low word, 2q+1 in the high word: SDWA selects the word); the rare exact pass
// reloads the full column from the table.
// Entries of the byte table carry bit 7 when the bin is the variable's zero bin: the "zero bin" flag of a second
// is then bit 7 of its packed byte, and the whole 8-bit stream is gathered from the two packed words with one
// multiply each (zero_stream below) instead of one compare + carry per second.
constexpr uint32_t kZeroFlag = 0x80u;
template <int M>
__device__ __forceinline__ void load_cthr(uint32_t (&tp)[(M + 1) / 2], uint32_t &bml, uint32_t &bmh, const uint32_t *__restrict__ p, int meff, uint32_t zbin1) {
    static_assert(M >= 1 && M <= 7, "the byte table has 8 entries");
    uint32_t th[M];
    load_cthr_full<M>(th, p, meff);
#pragma unroll
    for (int q = 0; q < (M + 1) / 2; q++) tp[q] = (th[2 * q] >> 16) | ((2 * q + 1 < M ? th[2 * q + 1] : 0xFFFFFFFFu) & 0xFFFF0000u);
    const uint32_t map = p[meff];
    uint32_t lo = 0u, hi = 0u;
#pragma unroll
    for (int b = 0; b <= M; b++) {
        const int n = M - b;                       // thresholds that fired
        const int nn = n < meff ? n : meff;        // the repeated ones fire with the last real one: n jumps to M
        uint32_t e = (map >> (4 * nn)) & 15u;
        e |= (e == zbin1) ? kZeroFlag : 0u;
        if (b < 4) lo |= e << (8 * b); else hi |= e << (8 * (b - 4));
    }
    bml = lo; bmh = hi;
}

__device__ __forceinline__ uint32_t zero_stream(uint32_t a, uint32_t b) { return byte_bit_stream<7>(a, b); }

// ---- the packed (two seconds per instruction) form of the same column ------------------------------------
// For the high-halfword compare every threshold X_t is represented by the 16-bit value T'_t such that
//     d = sat(x_h - T'_t)   is   0: not fired,   1: the low halfword decides (tie),   >= 2: fired,
// i.e. T'_t = H_t - 1 for H_t = X_t >> 16.  Two refinements keep "an odd sum of min(d, 2) <=> some tie" exact:
//   * the T' of a column are made strictly increasing (T'_t = max(H_t - 1, T'_{t-1} + 1)): thresholds that share a
//     high half would otherwise tie together and leave an even sum.  A shifted threshold can only be mis-decided
//     at an x_h where its predecessor in the chain reports a tie, so the block is redone exactly anyway;
//   * H_t = 0 has no T' (it would be -1): it gets T' = 0 and x_h = 0 is treated as a tie by a separate test of the
//     draws themselves (p = 2^-16 per draw, like any other tie).
// Thresholds beyond meff (the instance is built for M >= meff) get T' = 0xFFFF: never fired, never a tie.
// The byte table is indexed by the number of FIRED thresholds n (entries 0-3 in bnl, 4-7 in bnh; for M <= 3 by 2n, the sum of
// the min(d, 2) itself) and carries the zero-bin flag in bit 7 like the by-borrows table of the exact pass.
template <int M>
__device__ __forceinline__ void load_cthr_pk(uint32_t (&tp)[(M + 1) / 2], uint32_t &bnl, uint32_t &bnh, const uint32_t *__restrict__ p, int meff, uint32_t zbin1) {
    static_assert(M >= 1 && M <= 7, "the byte table has 8 entries");
    uint32_t tq[M];
    uint32_t prev = 0u;
#pragma unroll
    for (int t = 0; t < M; t++) {
        uint32_t v = 0xFFFFu;
        if (t < meff) {
            const uint32_t h = p[t] >> 16;
            v = h ? h - 1u : 0u;
            if (t > 0 && v <= prev) v = prev + 1u;
            v = v > 0xFFFFu ? 0xFFFFu : v;
        }
        tq[t] = v; prev = v;
    }
#pragma unroll
    for (int q = 0; q < (M + 1) / 2; q++) tp[q] = tq[2 * q] | ((2 * q + 1 < M ? tq[2 * q + 1] : 0xFFFFu) << 16);
    const uint32_t map = p[meff];
    uint32_t lo = 0u, hi = 0u;
#pragma unroll
    for (int n = 0; n <= M; n++) {
        uint32_t e = (map >> (4 * (n < meff ? n : meff))) & 15u;
        e |= (e == zbin1) ? kZeroFlag : 0u;
        const int at = (M <= 3) ? 2 * n : n;   // up to 3 thresholds: indexed by the sum itself (2 per fired threshold), no halving
        if (at < 4) lo |= e << (8 * at); else hi |= e << (8 * (at - 4));
    }
    bnl = lo; bnh = hi;
}

// Eight seconds of one dynamic variable, interior block (every second is a draw), decided from the high halfwords:
// same outputs as eight_seconds_pass<M, false, false>.  Returns bit 0 when a transition compare of this lane needs the low
// halfword, bit 1 when a resample compare does (the caller then redoes the block exactly).  No carries, no VCC: nothing here needs wait states.
template <int M>
__device__ __forceinline__ uint32_t eight_seconds_pk(const uint4 &th, const uint4 &rh, const uint32_t (&tp)[(M + 1) / 2], uint32_t bnl, uint32_t bnh,
                                                 uint32_t RR1, uint32_t cur_in, uint32_t &cur_out, uint32_t &pbA, uint32_t &pbB, uint32_t &hit8, uint32_t &chg8) {
    uint32_t nb2[4], par = 0u, hitA = 0u;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const uint32_t w = word_of(th, p), wr = word_of(rh, p);
        uint32_t acc = 0u;
#pragma unroll
        for (int t = 0; t < M; t++) {
            uint32_t d;                                                                      // select_random.m:19-20 for seconds 2p, 2p+1
            if (t & 1) asm("v_pk_sub_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] clamp" : "=v"(d) : "v"(w), "v"(tp[t >> 1]));
            else       asm("v_pk_sub_u16 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] clamp" : "=v"(d) : "v"(w), "v"(tp[t >> 1]));
            asm("v_pk_min_u16 %0, %0, 2 op_sel_hi:[1,0]" : "+v"(d));
            if (t == 0) acc = d; else asm("v_pk_add_u16 %0, %0, %1" : "+v"(acc) : "v"(d));
        }
        par |= acc;                                                                          // an odd count in either half: a tie
        uint32_t cnt = acc;
        if (M > 3) asm("v_pk_lshrrev_b16 %0, 1, %1 op_sel_hi:[0,1]" : "=v"(cnt) : "v"(acc));
        nb2[p] = __builtin_amdgcn_perm(bnh, bnl, cnt);                                       // dbn_sample.m:144: bins of 2p (byte 0) and 2p+1 (byte 2); bytes 1 and 3 (selector 0) are never read
        uint32_t u;                                                                          // resample_events.m:24: 0 no hit, 1 tie, 2 hit
        asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(u) : "s"(RR1), "v"(wr));
        asm("v_pk_min_u16 %0, %0, 2 op_sel_hi:[1,0]" : "+v"(u));
        hitA = p ? ((hitA << 2) | u) : u;
    }
    // x_h = 0 ties with a threshold whose high half is 0 (it has no T'): treat every such draw as a tie
    uint32_t mz, zt;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(mz) : "v"(th.x), "v"(th.y));
    asm("v_pk_min_u16 %0, %0, %1" : "+v"(mz) : "v"(th.z));
    asm("v_pk_min_u16 %0, %0, %1" : "+v"(mz) : "v"(th.w));
    asm("v_pk_sub_u16 %0, 1, %1 op_sel_hi:[0,1] clamp" : "=v"(zt) : "v"(mz));              // 1 in a half <=> that half of mz is 0
    pbA = __builtin_amdgcn_perm(nb2[1], nb2[0], 0x06040200u);
    pbB = __builtin_amdgcn_perm(nb2[3], nb2[2], 0x06040200u);
    hit8 = (hitA & 0xAAu) | ((hitA >> 17) & 0x55u);                                          // bit 1 of every 2-bit field, MSB-first
    // changed <=> the bin differs from the second before (dbn_sample.m:151-161); the low nibble alone tells
    const uint32_t prevA = (pbA << 8) | cur_in, prevB = __builtin_amdgcn_alignbit(pbB, pbA, 24);
    // bins are < 16 and only bit 7 (the zero flag) may be set above them: bit 4 of byte + 0x0F is the carry out of the low nibble
    const uint32_t yA = (pbA ^ prevA) + 0x0F0F0F0Fu, yB = (pbB ^ prevB) + 0x0F0F0F0Fu;
    chg8 = byte_bit_stream<4>(yA, yB);
    cur_out = pbB >> 24;
    return ((((par & 0x00010001u) | zt) != 0u) ? 1u : 0u) | (((hitA & 0x00550055u) != 0u) ? 2u : 0u);
}

// Eight seconds of one dynamic variable.  Outputs: bins packed 1-based 4 per word (pbA: seconds
// 0-3, pbB: 4-7) and three 8-bit flag streams, MSB-first (bit 7-j belongs to second j):
//   hit8  -- resample Bernoulli hit (resample_events.m:24)
//   chg8  -- the transition draw changed the bin (dbn_sample.m:151-161)
//   zer8  -- the bin after the draw is the zero bin (dediscretize.m:24-25); gathered from bit 7 of the packed bytes
// EXACT = false: decide from the high halfwords only and report `amb` when some compare could
// flip with the low halfword; EXACT = true: full 32-bit draws.
//
// The compare chains are written as carry arithmetic (two VOP2 instructions per compare, no
// select, no merge): a borrow out of x - X is "x < X"; flag streams take it with f = f + f + carry;
// the borrow count starts at kSelBase and feeds v_perm_b32 directly.  The differences x_h - X_h of
// the high halfwords are kept: a compare decided from them can only flip if the difference is 0.
// gfx950 needs two wait states between a VALU write of VCC and a VALU read of it (the assembler
// does not look inside asm blocks), hence the s_nop 1 in every pair; an SGPR operand is only read
// three wait states into a block, in case a VALU (v_readlane of a spilled SGPR) wrote it just before.
template <int M, bool EXACT, bool EDGE>
__device__ __forceinline__ bool eight_seconds_pass(const uint4 &th, const uint4 &rh, const uint4 &tl, const uint4 &rl, int g8, int T,
                                                   const uint32_t (&thr)[EXACT ? M : (M + 1) / 2], uint32_t bml, uint32_t bmh, uint32_t selbase,
                                                   uint32_t Rres, uint32_t cur_in,
                                                   uint32_t &cur_out, uint32_t &pbA, uint32_t &pbB, uint32_t &hit8, uint32_t &chg8) {
    uint32_t c1 = cur_in, dmin = 0xFFFFFFFFu;
    pbA = pbB = hit8 = chg8 = 0u;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int c = 8 * g8 + j; // absolute event time == column produced
        if (!EDGE || (c >= 1 && c < T)) {   // wave-uniform; interior blocks need no guard
            uint32_t d[M + 1], sel;
            if (EXACT) {
                const uint32_t xr = half_hi(rh, j) | half_lo(rl, j), xt = clamp32(half_hi(th, j) | half_lo(tl, j));
                // first threshold, then the resample Bernoulli: the SGPR operand is read three wait states into the block
                asm("v_sub_co_u32 %0, vcc, %4, %5\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %6, vcc\n\t"       // select_random.m:19-20
                    "v_subrev_co_u32 %2, vcc, %8, %7\n\ts_nop 1\n\tv_addc_co_u32 %3, vcc, %3, %3, vcc"         // resample_events.m:24
                    : "=&v"(d[0]), "=&v"(sel), "=&v"(d[M]), "+v"(hit8) : "v"(xt), "v"(thr[0]), "v"(selbase), "v"(xr), "s"(Rres) : "vcc");
#pragma unroll
                for (int t = 1; t < M; t++)
                    asm("v_sub_co_u32 %0, vcc, %2, %3\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
                        : "=&v"(d[t]), "+v"(sel) : "v"(xt), "v"(thr[t]) : "vcc");
            } else {
                // High halfwords only, read in place through SDWA operand selects (no extraction):
                // x_h - X_h borrows <=> x_h < X_h; the difference is 0 exactly when the low halfword decides.
                const uint32_t wt = word_of(th, j >> 1), wr = word_of(rh, j >> 1);
#define EMGPU_PAIR0(SELX)                                                                                                        \
                asm("v_sub_co_u32_sdwa %0, vcc, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" SELX " src1_sel:WORD_0\n\t"   \
                    "s_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %6, vcc\n\t"                                                           \
                    "v_subrev_co_u32_sdwa %2, vcc, %8, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:" SELX "\n\t" \
                    "s_nop 1\n\tv_addc_co_u32 %3, vcc, %3, %3, vcc"                                                               \
                    : "=&v"(d[0]), "=&v"(sel), "=&v"(d[M]), "+v"(hit8) : "v"(wt), "v"(thr[0]), "v"(selbase), "v"(wr), "s"(Rres) : "vcc")
#define EMGPU_PAIRT(SELX, SELT)                                                                                                  \
                asm("v_sub_co_u32_sdwa %0, vcc, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" SELX " src1_sel:" SELT "\n\t" \
                    "s_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"                                                                \
                    : "=&v"(d[t]), "+v"(sel) : "v"(wt), "v"(thr[t >> 1]) : "vcc")
                if (j & 1) {
                    EMGPU_PAIR0("WORD_1");
#pragma unroll
                    for (int t = 1; t < M; t++) { if (t & 1) EMGPU_PAIRT("WORD_1", "WORD_1"); else EMGPU_PAIRT("WORD_1", "WORD_0"); }
                } else {
                    EMGPU_PAIR0("WORD_0");
#pragma unroll
                    for (int t = 1; t < M; t++) { if (t & 1) EMGPU_PAIRT("WORD_0", "WORD_1"); else EMGPU_PAIRT("WORD_0", "WORD_0"); }
                }
#undef EMGPU_PAIR0
#undef EMGPU_PAIRT
#pragma unroll
                for (int t = 0; t + 1 <= M; t += 2) dmin = min(min(dmin, d[t]), d[t + 1]);
                if (!(M & 1)) dmin = min(dmin, d[M]);
            }
            const uint32_t nb1 = __builtin_amdgcn_perm(bmh, bml, sel);                          // dbn_sample.m:144
            asm("v_cmp_ne_u32 vcc, %1, %2\n\ts_nop 1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(chg8) : "v"(nb1), "v"(c1) : "vcc");
            c1 = nb1;                                                                           // map back, dbn_sample.m:149
        } else {
            hit8 += hit8; chg8 += chg8;
        }
        const uint32_t b = (!EDGE || c < T) ? (c1 << (8 * (j & 3))) : 0u;
        if (j < 4) pbA |= b; else pbB |= b;
    }
    cur_out = c1;
    return !EXACT && dmin == 0u;
}

// rare path, kept out of line so that the hot loop stays small in the instruction cache
template <int M>
__device__ __attribute__((noinline)) void eight_seconds_exact(uint32_t c0, uint32_t c1r, uint32_t attempt, uint32_t k0, uint32_t k1,
                                                              uint4 th, uint4 rh, uint32_t tvar, uint32_t ivar, int g8, int T,
                                                              const uint32_t *thr_col /* this lane's column in EmgpuPlan::cthr */, int meff,
                                                              uint32_t zbin1, uint32_t Rres, uint32_t cur_in, uint32_t which /* wave-uniform: 1 transition, 2 resample low halfwords needed */,
                                                              uint32_t *out /* cur, pbA, pbB, hit8, chg8 */) {
    const Rng rng{c0, c1r, attempt, k0, k1};
    // a low-halfword block that no lane of the wave needs is not generated: with zeros in its place every compare
    // that the high halfword decides (all of them, then) comes out the same
    uint4 tl = make_uint4(0u, 0u, 0u, 0u), rl = make_uint4(0u, 0u, 0u, 0u);
    if (which & 1u) tl = rng.block(EMGPU_SEC_TRANS_LO, tvar, (uint32_t)g8);
    if (which & 2u) rl = rng.block(EMGPU_SEC_RES_LO, ivar, (uint32_t)g8);
    uint32_t thr[M], bml, bmh;
    {   // the column again, in full and with the by-borrows byte table (rare path: nothing of this stays in registers)
        uint32_t tph[(M + 1) / 2];
        load_cthr<M>(tph, bml, bmh, thr_col, meff, zbin1);
    }
    load_cthr_full<M>(thr, thr_col, meff);
    uint32_t cur, a, b, h, c;
    eight_seconds_pass<M, true, true>(th, rh, tl, rl, g8, T, thr, bml, bmh, kSelBase, Rres, cur_in, cur, a, b, h, c);
    out[0] = cur; out[1] = a; out[2] = b; out[3] = h; out[4] = c;
}

template <int M>
__device__ __forceinline__ void eight_seconds(const Rng &rng, uint32_t tvar, uint32_t ivar, int g8, int T,
                                              const uint32_t *ctab /* the variable's compacted table */, int meff, const uint32_t *col_slot /* LDS: this lane's column */,
                                              const uint32_t (&thr)[(M + 1) / 2], uint32_t bnl, uint32_t bnh, uint32_t zbin1, uint32_t Rres, uint32_t RR1, uint32_t &cur1,
                                              uint32_t &pbA, uint32_t &pbB, uint32_t &hit8, uint32_t &chg8, uint32_t &zer8) {
    const uint4 th = rng.block(EMGPU_SEC_TRANS, tvar, (uint32_t)g8);
    const uint4 rh = rng.block(EMGPU_SEC_RES, ivar, (uint32_t)g8);
    uint32_t cur_out = cur1;
    // Interior blocks (every second 1 <= c < T) run the unguarded high-halfword pass inline.  The
    // last (partial) block of a trajectory, and any block in which SOME lane of the wave met a tie,
    // take the out-of-line exact pass (full 32-bit draws, guarded): lanes without a tie get the
    // same answers again, so control flow stays wave-uniform.
    const bool edge = 8 * g8 + 7 >= T; // the block runs past the end of the trajectory
    uint32_t redo = edge ? 3u : 0u;
    if (!edge) {
        const uint32_t amb = eight_seconds_pk<M>(th, rh, thr, bnl, bnh, RR1, cur1, cur_out, pbA, pbB, hit8, chg8);
        redo = (__ballot(amb & 1u) != 0ull ? 1u : 0u) | (__ballot(amb & 2u) != 0ull ? 2u : 0u);
        if (g8 == 0) {
            // Second 0 of a trajectory is the initial state, not a draw (slot 0 is never used,
            // dbn_sample.m:133,138).  The unguarded pass treated it as one: put the initial bin back,
            // clear its flags (streams are MSB-first: second j is bit 7-j) and re-derive "changed" of second 1.
            const uint32_t nb_1 = (pbA >> 8) & 0xFFu;
            pbA = (pbA & 0xFFFFFF00u) | cur1;
            hit8 &= 0x7Fu;
            chg8 = (chg8 & 0x3Fu) | ((nb_1 != cur1) ? 0x40u : 0u);
        }
    }
    if (redo) {
        EMGPU_COUNT(0, (int)(threadIdx.x & 63), 1);
        uint32_t out[5];
        const uint32_t *thr_col = ctab + (size_t)(*col_slot) * (uint32_t)(meff + 1);
        eight_seconds_exact<M>(rng.c0, rng.c1, rng.attempt, rng.k0, rng.k1, th, rh, tvar, ivar, g8, T, thr_col, meff, zbin1, Rres, cur1, redo, out);
        cur_out = out[0]; pbA = out[1]; pbB = out[2]; hit8 = out[3]; chg8 = out[4];
    }
    cur1 = cur_out;                      // still carries the zero-bin flag
    zer8 = zero_stream(pbA, pbB);        // dediscretize.m:24-25 (the bit of second 0 of a trajectory is never consumed)
    pbA &= 0x7F7F7F7Fu; pbB &= 0x7F7F7F7Fu;
}

// The whole kernel as a function of (plan, run, workgroup number within the run): k_uncor_fast runs it on the kernel's own arguments,
// k_uncor_fast_mixed on the entry of the model block its workgroup belongs to.
// MIXED: the plan is read from device memory (not from the kernel arguments): what the 8-second loop uses of it is pinned in
// scalar registers up front (readfirstlane) -- left to the compiler these became vector loads inside the loop, each waiting
// (vmcnt counts stores too on gfx9) for the block's trace stores.
// EV: the event list of dbn_hierarchical_sample.m:33-60 is written as well (uncor_fast_events below).
// IDX: an index list may be in use (emgpu_sample_params.indices): the workers read the owner's global index from LDS instead of
// deriving it from their own (kept out of the plain instance: the benchmark kernel pays 1 % for the possibility).
// the stream table of a wide event list lives in LDS of the instances that write one, and nowhere else
template <int EVW>
__device__ __forceinline__ EvStream *evw_lds() {
    if constexpr (EVW != 0) { __shared__ EvStream s[16]; return s; }
    else return nullptr;
}
// The request queue of the rows-by-the-wave form: EVW 2 -- the cooperative dediscretize's own queue, idle in that form: 254 requests per
// round, the workgroup stays within 40 KB of LDS (four per CU) and 128 registers (four waves per SIMD); EVW 3 -- 1 024 requests per round in
// LDS of its own (three workgroups per CU), for lists of several hundred rows per wave and block (haa_v1: ~900), where 254 mean four rounds
constexpr int kFastRowsQueue = 254;
template <int EVW>
__device__ __forceinline__ uint16_t *evu_lds(int wave) {
    if constexpr (EVW == 3) { __shared__ uint16_t q[4][kEvRowsQueue<3> + 2]; return q[wave]; }
    else return nullptr;
}
// EVW: the event list of a model with more rated variables than the eight streams of EV hold (haa_v1): emgpu_events.h "WIDE lists"
// (1: result slots + a row loop per lane, any outputs; 2, 3: events only, the rows built by the wave -- "ROWS BY THE WAVE")
template <int NI, int M0, int M1, int M2, bool MIXED = false, bool EV = false, bool IDX = false, int EVW = 0>
__device__ __forceinline__ void uncor_fast_body(const EmgpuPlan &P, const EmgpuRun &A, const FastArgs &F, const int64_t i0 /* trajectory of lane 0: wave-uniform, may be < 0 */) {
    // workers look the bin of a request up in LDS (the owner does not encode it into the request): -1.8 % on the
    // <7,4,6,6> instance too since the packed compare pass freed its registers
    constexpr bool LB = true;
    __shared__ CoopLds<3, LB> s_wave[4];
    __shared__ double s_bnd[3][16];
    const int tid = threadIdx.x, lane = tid & 63;
    CoopLds<3, LB> &W = s_wave[tid >> 6];
    const int64_t i = i0 + tid;
    const bool valid = i >= 0 && i < A.n; // lanes outside the run stay alive: they serve as workers for their wave
    uint64_t gidx = A.first_index + (uint64_t)i;
    if constexpr (IDX) {
        if (A.indices != nullptr && valid) gidx = A.indices[i];   // (wave-uniform pointer test)
        coop_publish_gidx<3>(W, lane, gidx);
    }
    Rng rng{(uint32_t)gidx, (uint32_t)(gidx >> 32), 0u, (uint32_t)A.seed, (uint32_t)(A.seed >> 32)};
    const int T = A.T;
#pragma unroll
    for (int k = 0; k < 3; k++) // k stays a compile-time index into the plan (a per-lane index would force it into scratch)
        if ((tid >> 4) == k) {
            const int q = tid & 15;
            s_bnd[k][q] = (q < (int)P.d_nb[k]) ? P.bnd[P.d_boff[k] + q] : 0.0;
        }

    int bin[NI];
    double val[NI];
#pragma unroll
    for (int p = 0; p < NI; p++) { bin[p] = 0; val[p] = 0.0; }
#ifdef EMGPU_FAST_INIT_KARG   // measuring variant (HISTORY.md section 9): -1.9 % vector instructions, judged in both box states
    const int32_t attempts_used = init_network_karg<NI>((KargPlan)__builtin_amdgcn_kernarg_segment_ptr(), A, rng, bin, val);
#else
    const int32_t attempts_used = init_network<NI>(P, A, rng, bin, val);
#endif
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
    __syncthreads(); // s_bnd visible to every wave (the only block-wide barrier)

    // frozen parent configuration -> one CPT column per dynamic variable (dbn_sample.m:110-135)
    uint32_t cur1[3];
    float cval[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        cur1[k] = (uint32_t)pick<NI>(bin, P.d_ipos[k]) + 1u;
        cval[k] = (float)pick<NI>(val, P.d_ipos[k]);
    }
    uint32_t th0[(M0 + 1) / 2], th1[(M1 + 1) / 2], th2[(M2 + 1) / 2], bl0, bl1, bl2, bh0, bh1, bh2;
    {
        uint32_t col[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            uint32_t c = 0;
#pragma unroll
            for (int p = 0; p < NI; p++) c += P.d_stride_static[k][p] * (uint32_t)bin[p];
#pragma unroll
            for (int q = 0; q < 3; q++) c += P.d_stride_cur[k][q] * (uint32_t)(cur1[q] - 1);
            col[k] = c;
        }
        load_cthr_pk<M0>(th0, bl0, bh0, P.cthr + P.d_coff[0] + (size_t)col[0] * (uint32_t)(P.d_meff[0] + 1), P.d_meff[0], (uint32_t)P.d_zero[0]);
        load_cthr_pk<M1>(th1, bl1, bh1, P.cthr + P.d_coff[1] + (size_t)col[1] * (uint32_t)(P.d_meff[1] + 1), P.d_meff[1], (uint32_t)P.d_zero[1]);
        load_cthr_pk<M2>(th2, bl2, bh2, P.cthr + P.d_coff[2] + (size_t)col[2] * (uint32_t)(P.d_meff[2] + 1), P.d_meff[2], (uint32_t)P.d_zero[2]);
        // the exact pass finds its column again through the lane's spare LDS words
#pragma unroll
        for (int k = 0; k < 3; k++) reinterpret_cast<uint32_t *>(&W.res[lane * CoopLds<3, LB>::kStride + CoopLds<3, LB>::kSpare])[k] = col[k];
    }
    // from here on the current bin carries the zero-bin flag like the entries of the byte tables
#pragma unroll
    for (int k = 0; k < 3; k++) cur1[k] |= (cur1[k] == (uint32_t)P.d_zero[k]) ? kZeroFlag : 0u;
    auto U = [](uint32_t v) -> uint32_t { return MIXED ? (uint32_t)__builtin_amdgcn_readfirstlane((int)v) : v; };
    const uint32_t iv0 = U(P.d_ivar[0]), iv1 = U(P.d_ivar[1]), iv2 = U(P.d_ivar[2]);
    const uint32_t ivs[3] = {iv0, iv1, iv2};
    uint32_t h_tvar[3], h_meff[3], h_zero[3], h_Rk[3], h_RR1[3], h_slot[3];
    const uint32_t *h_ctab[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        h_tvar[k] = U(P.d_tvar[k]); h_meff[k] = U(P.d_meff[k]); h_zero[k] = U(P.d_zero[k]);
        h_Rk[k] = U(F.Rk[k]); h_RR1[k] = U(F.RR1[k]); h_slot[k] = U(F.slot[k]);
        const uint64_t a = (uint64_t)(P.cthr + P.d_coff[k]);
        h_ctab[k] = reinterpret_cast<const uint32_t *>(((uint64_t)U((uint32_t)(a >> 32)) << 32) | U((uint32_t)a));
    }
    const uint32_t *col_slot = reinterpret_cast<const uint32_t *>(&W.res[lane * CoopLds<3, LB>::kStride + CoopLds<3, LB>::kSpare]);
    EvPlan E{};
    EvState S{};
    EvStateW SW{};
    EvStream *s_evw = evw_lds<EVW>();
    if constexpr (EV && EVW == 0) {
        E = ev_plan_of<NI, 3>(P);
        S = ev_state_of<NI, 3>(P, A, bin, valid, i);
    }
    uint16_t *s_evq = EVW == 3 ? evu_lds<EVW>(tid >> 6) : reinterpret_cast<uint16_t *>(W.queue);
    static_assert(sizeof(W.queue) >= (kFastRowsQueue + 2) * sizeof(uint16_t), "rows queue");
    if constexpr (EVW != 0) {
        ev_wide_plan<3>(P, s_evw);
        SW = ev_state_w_of<NI, 3>(P, A, bin, valid, i);
        if constexpr (EVW >= 2) ev_rows_publish_static<3>(W, lane, SW);
        __syncthreads();
    }
    const int G4 = (T + 3) >> 2, G8 = (T + 7) >> 3;
    for (int g8 = 0; g8 < G8; g8++) {
        uint32_t pbA[3], pbB[3], need8[3], kind8[3], fill8[3];
        uint32_t prevw = 0u, hit24 = 0u;   // EV: the bins the block starts from, the resample hits of the dynamic variables
        if constexpr (EV) prevw = (cur1[0] & 0x7Fu) | ((cur1[1] & 0x7Fu) << 8) | ((cur1[2] & 0x7Fu) << 16);
        {
            uint32_t hit8[3], chg8[3], zer8[3];
            eight_seconds<M0>(rng, h_tvar[0], iv0, g8, T, h_ctab[0], (int)h_meff[0], col_slot + 0, th0, bl0, bh0, h_zero[0], h_Rk[0], h_RR1[0], cur1[0], pbA[0], pbB[0], hit8[0], chg8[0], zer8[0]);
            eight_seconds<M1>(rng, h_tvar[1], iv1, g8, T, h_ctab[1], (int)h_meff[1], col_slot + 1, th1, bl1, bh1, h_zero[1], h_Rk[1], h_RR1[1], cur1[1], pbA[1], pbB[1], hit8[1], chg8[1], zer8[1]);
            eight_seconds<M2>(rng, h_tvar[2], iv2, g8, T, h_ctab[2], (int)h_meff[2], col_slot + 2, th2, bl2, bh2, h_zero[2], h_Rk[2], h_RR1[2], cur1[2], pbA[2], pbB[2], hit8[2], chg8[2], zer8[2]);
#pragma unroll
            for (int k = 0; k < 3; k++) {      // the streams stay MSB-first: bit (7-j) <-> second j
                need8[k] = (hit8[k] | chg8[k]) & ~zer8[k];   // a dediscretize draw is due (dediscretize.m:24-39)
                kind8[k] = chg8[k];                          // 1 = transition event (it hides a resample event of the same second)
                fill8[k] = need8[k] | chg8[k];               // the value changes: a draw, or 0 on a change into the zero bin
            }
            if constexpr (EV) hit24 = hit8[0] | (hit8[1] << 8) | (hit8[2] << 16);
        }
        EMGPU_COUNT(5, lane, 1);
        const uint32_t need24 = valid ? (need8[0] | (need8[1] << 8) | (need8[2] << 16)) : 0u;
        const uint32_t kind24 = kind8[0] | (kind8[1] << 8) | (kind8[2] << 16);
        if constexpr (EVW >= 2) {   // events only: no result slots, no fill -- every row is one request of the wave's queue
            coop_publish_bins<3>(W, lane, pbA, pbB);
            ev_rows_block_wide<3, EVW == 3 ? kEvRowsQueue<3> : kFastRowsQueue>(W, s_evq, lane, s_evw, P.nact, SW, rng, P.bnd, g8, T, valid, hit24, kind24, prevw, A, i);
        } else {
        coop_zero_results<3, LB>(W, lane);
        if constexpr (LB) coop_publish_bins<3>(W, lane, pbA, pbB);
        coop_dedisc<3, true, LB, IDX>(W, lane, gidx, rng, g8, need24, kind24, pbA, pbB, ivs, s_bnd);   // dediscretize.m:39
        if (!EV || A.dyn_bin != nullptr || A.dyn_val != nullptr)   // (an event-list call without the dense trace: no forward fill at all)
#pragma unroll
        for (int k = 0; k < 3; k++)
            coop_fill_store_msb<3, LB, !EV && !IDX, true>(W, lane, k, g8, T, G4, valid, fill8[k], cval[k], pbA[k], pbB[k],   // (the plain and the mixed kernel are only launched with both dense outputs)
                                   3u, h_slot[k], i0, (uint32_t)tid, A.ld, A.dyn_bin, A.dyn_val);
        if constexpr (EV && EVW == 0) ev_emit_block<3, LB>(W, lane, E, S, rng, P.bnd, g8, T, valid, hit24, kind24, prevw);
        if constexpr (EVW == 1) ev_emit_block_wide<3, LB>(W, lane, s_evw, P.nact, SW, rng, P.bnd, g8, T, valid, hit24, kind24, prevw);
        }
        wave_sync(); // results of this block are consumed before the next block's workers overwrite them
    }
    if constexpr (EV) {
        const uint32_t curp = (cur1[0] & 0x7Fu) | ((cur1[1] & 0x7Fu) << 8) | ((cur1[2] & 0x7Fu) << 16);
        if constexpr (EVW != 0) ev_tail_wide<3>(s_evw, P.nact, SW, rng, P.bnd, T, curp, A, valid, i);
        else ev_tail<3>(E, S, rng, P.bnd, T, curp, A, valid, i);
    }
}

template <int NI, int M0, int M1, int M2>
__global__ void __launch_bounds__(256, EMGPU_FAST_WAVES) k_uncor_fast(const EmgpuPlan P, const EmgpuRun A, const FastArgs F) {
    // workgroup w covers columns [256 w, 256 w + 256) of the TRACE: a shard that starts at column col0 leads with col0 mod 256 idle lanes
    uncor_fast_body<NI, M0, M1, M2>(P, A, F, (int64_t)blockIdx.x * 256 - (A.col0 & 255));
}

// The same kernel with the event list written as well (dense outputs are optional here): UncorEncounterModel.sample's own output
// type, served at the dense kernel's pace instead of k_dbn_generic's.  Three waves per SIMD: the list's state does not fit the
// dense instance's 128 registers.
template <int NI, int M0, int M1, int M2>
__global__ void __launch_bounds__(256, 3) k_uncor_fast_ev(const EmgpuPlan P, const EmgpuRun A, const FastArgs F) {
    uncor_fast_body<NI, M0, M1, M2, false, true, true>(P, A, F, (int64_t)blockIdx.x * 256 - (A.col0 & 255));
}
// ... and for a model with up to 13 rated variables (haa_v1 has 7: it ran on k_dbn_generic)
template <int NI, int M0, int M1, int M2>
__global__ void __launch_bounds__(256, 3) k_uncor_fast_evw(const EmgpuPlan P, const EmgpuRun A, const FastArgs F) {
    uncor_fast_body<NI, M0, M1, M2, false, true, true, 1>(P, A, F, (int64_t)blockIdx.x * 256 - (A.col0 & 255));
}
// ... and its events-only form: the rows of a block built 64 at a time by the wave (emgpu_events.h "ROWS BY THE WAVE")
template <int NI, int M0, int M1, int M2>
__global__ void __launch_bounds__(256, 4) k_uncor_fast_evu(const EmgpuPlan P, const EmgpuRun A, const FastArgs F) {
    uncor_fast_body<NI, M0, M1, M2, false, true, true, 2>(P, A, F, (int64_t)blockIdx.x * 256 - (A.col0 & 255));
}
template <int NI, int M0, int M1, int M2>
__global__ void __launch_bounds__(256, 3) k_uncor_fast_evu_long(const EmgpuPlan P, const EmgpuRun A, const FastArgs F) {
    uncor_fast_body<NI, M0, M1, M2, false, true, true, 3>(P, A, F, (int64_t)blockIdx.x * 256 - (A.col0 & 255));
}
// the dense kernel for an index list (the later rounds of UncorEncounterModel.track: the trajectories still rejected)
template <int NI, int M0, int M1, int M2>
__global__ void __launch_bounds__(256, EMGPU_FAST_WAVES) k_uncor_fast_idx(const EmgpuPlan P, const EmgpuRun A, const FastArgs F) {
    uncor_fast_body<NI, M0, M1, M2, false, false, true>(P, A, F, (int64_t)blockIdx.x * 256 - (A.col0 & 255));
}

// Mixed-model batch in ONE launch (RUN_1_emsample.m:13,24-47 shards by model file; SURVEY.md 8e: "model id per block"): the models
// of the batch share this kernel instance, every model block of the batch owns a contiguous range of workgroups
// [wg_begin[b], wg_begin[b+1]) -- lined up with the trace's columns like a single-model launch, so a workgroup at a model
// boundary exists twice with complementary live lanes and no workgroup meets two table sets.  What differs between the blocks
// of one call travels in the kernel arguments (index range, first column in the shared trace); a model's plan and resample
// thresholds sit in device memory next to its tables (uploaded once with them) and are read through the constant address space:
// the same scalar loads that fetch a single-model launch's kernel arguments.
struct PlanF {
    EmgpuPlan P;
    FastArgs F;
};
struct MixedBlock {
    const PlanF *pf;
    uint64_t first_index;
    int64_t n, col; // trajectories; first column of the block in the call's trace
};
struct MixedHead {
    EmgpuRun A; // the call's run: outputs at column 0, n / first_index unused
    int32_t nb, _pad;
    uint32_t wg_begin[EMGPU_MAX_MIXED + 1];
    MixedBlock blk[EMGPU_MAX_MIXED];
};
template <int NI, int M0, int M1, int M2>
__global__ void __launch_bounds__(256, EMGPU_FAST_WAVES) k_uncor_fast_mixed(const MixedHead H) {
    uint32_t b = 0;
#pragma unroll
    for (int q = 1; q < EMGPU_MAX_MIXED; q++) b += (q < H.nb && blockIdx.x >= H.wg_begin[q]) ? 1u : 0u;
    const MixedBlock B = H.blk[b];
    EmgpuRun A = H.A;
    A.first_index = B.first_index; A.n = B.n;
    const size_t c = (size_t)B.col;
    A.init_bin = A.init_bin ? A.init_bin + c : nullptr;
    A.init_val = A.init_val ? A.init_val + c : nullptr;
    A.dyn_bin = A.dyn_bin ? A.dyn_bin + c : nullptr;
    A.dyn_val = A.dyn_val ? A.dyn_val + 4 * c : nullptr;
    A.attempts = A.attempts ? A.attempts + c : nullptr;
    typedef const __attribute__((address_space(4))) PlanF *CPlanF;
    const PlanF &E = *(const PlanF *)((CPlanF)B.pf);
    uncor_fast_body<NI, M0, M1, M2, true>(E.P, A, E.F, (int64_t)(blockIdx.x - H.wg_begin[b]) * 256 - ((H.A.col0 + B.col) & 255));
}

// Kernel instances by the number of DISTINCT thresholds per column of the three dynamic variables
// (EmgpuPlan::d_meff).  A model runs on the first instance that covers it.
struct FastShape { int ni, m0, m1, m2; };
static const FastShape kFastShapes[] = {
    {7, 2, 2, 2}, {7, 2, 4, 2}, {7, 2, 4, 4}, {7, 4, 2, 4}, {7, 4, 6, 4}, {7, 4, 6, 6}, {7, 6, 6, 6}, {9, 6, 6, 6},
};

static inline int fast_shape_of(const EmgpuPlan &P) {
    for (size_t q = 0; q < sizeof kFastShapes / sizeof kFastShapes[0]; q++) {
        const FastShape &f = kFastShapes[q];
        if (P.ni <= f.ni && P.d_meff[0] <= f.m0 && P.d_meff[1] <= f.m1 && P.d_meff[2] <= f.m2) return (int)q;
    }
    return -1;
}

static inline FastArgs fast_args_of(const EmgpuPlan &P) {
    FastArgs F{};
    for (int k = 0; k < 3; k++) {
        F.slot[k] = P.d_row[k];
        for (int a = 0; a < P.nact; a++)
            if (P.a_dyn[a] == k) F.Rk[k] = P.a_R[a];
        F.RR1[k] = ((F.Rk[k] >> 16) + 1u) * 0x00010001u;
    }
    return F;
}


} // namespace emgpu
