// emgpu_coop.h -- wave-cooperative dediscretize and the dense-trace store, shared by the
// 8-second-block kernels (k_uncor_fast, k_dbn_step2, k_dbn_step).
//
// A dediscretize draw (dediscretize.m:39) is due only at an event (~0.3 per lane and 8-second
// block), but a per-lane `if (event) philox()` makes the whole wave pay for a Philox call whenever
// ANY of its 64 lanes has an event.  Instead the flagged positions s = 8k + j of all lanes are
// compacted into a per-wave LDS queue (positions from one prefix sum over the lanes' request counts,
// no atomics), lanes 0..R-1 each serve one request (one Philox call per wave serves up to 64 events),
// and the f32 results go back through LDS to the owners.  Everything is wave-local: no __syncthreads, only a wave fence.
#pragma once
#include <type_traits>

#include "emgpu_device.h"

namespace emgpu {

// -DEMGPU_DEBUG_COUNTERS: wave-level event counts of the data-dependent paths (diagnostic builds only, read
// through emgpu_debug_counters): 0 exact redos, 1 compaction rounds, 2 compaction steps, 3 worker passes, 4 requests, 5 blocks
#ifdef EMGPU_DEBUG_COUNTERS
__device__ unsigned long long g_dbg[8];
#define EMGPU_COUNT(slot, lane, val) do { if ((lane) == 0) atomicAdd(&g_dbg[slot], (unsigned long long)(val)); } while (0)
#else
#define EMGPU_COUNT(slot, lane, val) do { } while (0)
#endif

// Bit B (4..7) of the 8 packed bytes (seconds 0-3 in a, 4-7 in b) as an MSB-first stream: bit 7-j <-> second j.  One mask and
// one v_mul_hi_u32 per word: the multiplier moves bit 8i + B of the masked word to bit 35 - i of the 64-bit product (the 16
// partial products land on 16 different bits: no carries), so the low nibble of the high word is the stream of that word.
template <int B>
__device__ __forceinline__ uint32_t byte_bit_stream(uint32_t a, uint32_t b) {
    static_assert(B >= 4 && B <= 7, "the shifts 35 - B - 9i must fit a 32-bit multiplier");
    constexpr uint32_t K = (1u << (35 - B)) | (1u << (26 - B)) | (1u << (17 - B)) | (1u << (8 - B));
#ifdef EMGPU_OLD_STREAM
    const uint32_t na = (((a >> B) & 0x01010101u) * 0x80402010u) >> 28, nb = (((b >> B) & 0x01010101u) * 0x80402010u) >> 28;
    return (na << 4) | nb;
#endif
    const uint32_t ha = __umulhi(a & (0x01010101u << B), K), hb = __umulhi(b & (0x01010101u << B), K);
    uint32_t t;
    asm("v_lshl_or_b32 %0, %1, 4, %2" : "=v"(t) : "v"(ha), "v"(hb)); // (the compiler splits the mask over both operands: one more instruction)
    return t & 0xFFu; // other partial products sit at bits >= 9 of the high words
}
// LB: the lanes also publish the packed bins of the block (2 words per variable, after the 8*ND result
// slots) and their kind mask so that a worker looks the bin and kind of a request up itself: a request is
// then 11 bits (owner lane | bit of the owner's need mask << 6) and the queue holds 256 of them in the
// 512 bytes that hold 128 of the self-contained 32-bit requests of the other form.
template <int ND, bool LB = false>
struct CoopLds {
    static constexpr int kBins = 8 * ND;                          // word offset of the published bins
    static constexpr int kSpare = 8 * ND + (LB ? 2 * ND : 0);     // word offset of the lane's spare words (>= 4 of them)
    static constexpr int kKind = kSpare + 3;                      // the lane's kind mask of the block, read by the workers (LB)
    static constexpr int kStride = ((kSpare + 3 + 3) / 8) * 8 + 4; // words per lane; = 4 (mod 8) keeps the b128 reads conflict-free
    static_assert(kStride % 8 == 4 && kStride >= kSpare + 4, "lane stride");
    // the lane's global sample index (two words), read by the worker that serves one of its requests: with an index list
    // (emgpu_sample_params.indices) a neighbour's index is not this lane's plus the lane distance.  After the kind word where the
    // row has room (3 variables), else in the column slots (only k_uncor_fast, a 3-variable kernel, uses those).
    static constexpr int kGidx = (kSpare + 6 <= kStride) ? kSpare + 4 : kSpare;
    static constexpr int kCap = LB ? 256 : 128;                   // requests per compaction round
    using Request = typename std::conditional<LB, uint16_t, uint32_t>::type;
    Request queue[kCap];
    float res[64 * kStride];
    uint32_t attempt[64];
};

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int ND>
__device__ __forceinline__ uint32_t pick_word(const uint32_t (&a)[ND], uint32_t k) {
    uint32_t r = 0u; // bit masks, not ?: -- see pick() in emgpu_device.h
#pragma unroll
    for (int q = 0; q < ND; q++) r |= pick_bits(a[q], k == (uint32_t)q);
    return r;
}

// inclusive prefix sum over the 64 lanes of a (fully active) wave: four row-local DPP steps, two row broadcasts
__device__ __forceinline__ uint32_t wave_inclusive_add(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false); // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false); // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false); // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false); // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2, 3
    return x;
}

// One worker pass: lanes 0..cnt-1 each serve the request queue[q0 + lane].  A request is owner | sb << 6 (sb = the
// bit of the owner's need mask); with LB the worker reads the owner's bin and kind word from the owner's LDS row,
// without it the owner encoded them (kind << 11, bin << 12).
template <int ND, bool MSBFIRST, bool LB, bool IDX = false>
__device__ __forceinline__ void coop_worker_pass(CoopLds<ND, LB> &W, int lane, uint32_t q0, uint32_t cnt, uint64_t gidx, const Rng &rng, int g8,
                                                 uint32_t ivpack, const double (*s_bnd)[16]) {
    using L = CoopLds<ND, LB>;
    const uint32_t q = q0 + (uint32_t)lane;
    if (q < cnt) {
        const uint32_t d = W.queue[q];
        const uint32_t owner = d & 63u, sb = (d >> 6) & 31u;
        const uint32_t s = MSBFIRST ? (sb ^ 7u) : sb;
        const uint32_t kind = LB ? ((reinterpret_cast<const uint32_t *>(&W.res[owner * L::kStride + L::kKind])[0] >> sb) & 1u) : ((d >> 11) & 1u);
        const uint32_t b1 = LB ? reinterpret_cast<const uint8_t *>(&W.res[owner * L::kStride + L::kBins])[s] : ((d >> 12) & 63u);
        const uint32_t k = s >> 3, j = s & 7u;
        uint64_t go = gidx - (uint64_t)lane + (uint64_t)owner;
        if constexpr (LB && IDX) go = *reinterpret_cast<const uint64_t *>(&W.res[owner * L::kStride + L::kGidx]);   // an index list may be in use (coop_publish_gidx)
        const uint32_t iv = (ivpack >> (8u * k)) & 0xFFu;
        const uint32_t sec = kind ? EMGPU_SEC_DEDISC_TRANS : EMGPU_SEC_DEDISC_RES;
        const uint4 r4 = philox4x32((uint32_t)go, (uint32_t)(go >> 32), W.attempt[owner],
                                       (sec << 28) | (iv << 20) | (uint32_t)(2 * g8 + (int)(j >> 2)), rng.k0, rng.k1);
        const uint32_t w = j & 3u;
        const uint32_t x = w == 0 ? r4.x : (w == 1 ? r4.y : (w == 2 ? r4.z : r4.w));
        double v;
        {
#pragma clang fp contract(off)
            const double a = s_bnd[k][b1 - 1u], b = s_bnd[k][b1];
            const double dd = b - a;
            const double mm = dd * uniform32(x);
            v = a + mm;
        }
        W.res[owner * L::kStride + s] = (float)v;
    }
}

// needmask / kindmask: bit (8k + j) for dynamic variable k, second j of the block; with MSBFIRST
// the byte of a variable is an MSB-first stream instead: bit (8k + 7 - j).
// pbA / pbB: packed 1-based bins of seconds 0-3 / 4-7.  s_bnd[k][]: boundaries of variable k.
//
// LB: the requests of the wave get their queue positions from ONE prefix sum over the lanes' request counts; a
// lane then writes its own requests to consecutive slots (find-first-set, one LDS store, clear the bit: no ballot
// or rank per request), kCap positions per round, and the workers serve exactly ceil(requests / 64) passes.
// Without LB (k_dbn_step, the fallback kernel) the round-1 scheme stays: one ballot + rank per compaction step.
template <int ND, bool MSBFIRST = false, bool LB = false, bool IDX = false>
__device__ __forceinline__ void coop_dedisc(CoopLds<ND, LB> &W, int lane, uint64_t gidx, const Rng &rng, int g8,
                                            uint32_t needmask, uint32_t kindmask, const uint32_t (&pbA)[ND], const uint32_t (&pbB)[ND],
                                            const uint32_t (&ivar)[ND], const double (*s_bnd)[16]) {
    using L = CoopLds<ND, LB>;
    uint32_t ivpack = 0u; // wave-uniform byte table of the variables' RNG ids
#pragma unroll
    for (int q = 0; q < ND; q++) ivpack |= ivar[q] << (8 * q);
    uint32_t m = needmask;
    if constexpr (LB) {
        if (__ballot(m != 0u) == 0ull) return;
        const uint32_t c = (uint32_t)__popc(m);
        const uint32_t inc = wave_inclusive_add(c);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        // byte positions of this lane's requests in the (virtual, unbounded) queue: [a, aend)
        constexpr uint32_t kReq = (uint32_t)sizeof(typename L::Request);
        uint32_t a = (inc - c) * kReq;
        const uint32_t aend = inc * kReq;
        reinterpret_cast<uint32_t *>(&W.res[lane * L::kStride + L::kKind])[0] = kindmask;
        EMGPU_COUNT(4, lane, total);
        if (total <= (uint32_t)L::kCap) {
            // the usual case, one round: find-first-set, one LDS store, clear the bit
            EMGPU_COUNT(1, lane, 1);
            typename L::Request *qp = reinterpret_cast<typename L::Request *>(reinterpret_cast<char *>(W.queue) + a);
            while (__ballot(m != 0u) != 0ull) {
                EMGPU_COUNT(2, lane, 1);
                if (m != 0u) {
                    *qp++ = (typename L::Request)((uint32_t)lane | (((uint32_t)__ffs((int)m) - 1u) << 6));
                    m &= m - 1u;
                }
            }
            wave_sync();
            for (uint32_t q0 = 0u; q0 < total; q0 += 64u) {
                EMGPU_COUNT(3, lane, 1);
                coop_worker_pass<ND, MSBFIRST, LB, IDX>(W, lane, q0, total, gidx, rng, g8, ivpack, s_bnd);
            }
            wave_sync();
        } else {
            // more than kCap requests in the wave-block: rounds of kCap positions
            for (uint32_t rb = 0u; rb < total; rb += (uint32_t)L::kCap) {
                EMGPU_COUNT(1, lane, 1);
                const uint32_t lim = min(aend, (rb + (uint32_t)L::kCap) * kReq); // a lane is active while a < lim
                char *const qb = reinterpret_cast<char *>(W.queue) - rb * kReq;
                while (__ballot(a < lim) != 0ull) {
                    EMGPU_COUNT(2, lane, 1);
                    if (a < lim) {
                        *reinterpret_cast<typename L::Request *>(qb + a) = (typename L::Request)((uint32_t)lane | (((uint32_t)__ffs((int)m) - 1u) << 6));
                        a += kReq;
                        m &= m - 1u;
                    }
                }
                wave_sync();
                const uint32_t cnt = min(total - rb, (uint32_t)L::kCap);
                for (uint32_t q0 = 0u; q0 < cnt; q0 += 64u) {
                    EMGPU_COUNT(3, lane, 1);
                    coop_worker_pass<ND, MSBFIRST, LB, IDX>(W, lane, q0, cnt, gidx, rng, g8, ivpack, s_bnd);
                }
                wave_sync();
            }
        }
    } else {
        unsigned long long bal = __ballot(m != 0u);
        while (bal != 0ull) {
            uint32_t base = 0u; // wave-uniform number of queued requests in this round
            EMGPU_COUNT(1, lane, 1);
            // a compaction step joins the round only while the round still fits ONE worker pass of 64 requests (the first step always does)
            while (bal != 0ull && base + (uint32_t)__popcll(bal) <= 64u) {
                EMGPU_COUNT(2, lane, 1);
                if (m != 0u) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    const uint32_t sb = (uint32_t)__ffs((int)m) - 1u;
                    const uint32_t s = MSBFIRST ? (sb ^ 7u) : sb;
                    const uint32_t k = s >> 3, j = s & 7u;
                    const uint32_t wA = pick_word<ND>(pbA, k), wB = pick_word<ND>(pbB, k);
                    const uint32_t b1 = (((j & 4u) ? wB : wA) >> (8u * (j & 3u))) & 0xFFu;
                    W.queue[base + rank] = (uint32_t)lane | (sb << 6) | (((kindmask >> sb) & 1u) << 11) | (b1 << 12);
                    m &= m - 1u;
                }
                base += (uint32_t)__popcll(bal);
                bal = __ballot(m != 0u);
            }
            wave_sync();
            EMGPU_COUNT(4, lane, base);
            EMGPU_COUNT(3, lane, 1);
            coop_worker_pass<ND, MSBFIRST, LB>(W, lane, 0u, base, gidx, rng, g8, ivpack, s_bnd);
            wave_sync();
            bal = __ballot(m != 0u);
        }
    }
}

// ---- Self-contained requests on the prefix-sum queue (round 4; the dense 16-variable instances of k_dbn_step2).  A request is 16 bits:
// owner lane (6) | bit of the owner's need mask (5) | kind (1) | bin - 1 (4: at most 15 bins, step2_eligible) -- the owner looks its own
// bin up when it writes the request, nothing is published per lane, and the lane's LDS row is its 8 ND result slots + 4 words: 36 words
// for ND = 4 where the published-bins form needs 44.  Queue positions come from ONE prefix sum like the LB form (CoopLds<ND, false>'s 512
// queue bytes hold 256 of these requests).
template <int ND, bool MSBFIRST>
__device__ __forceinline__ void coop_worker_pass_sc(CoopLds<ND, false> &W, const uint16_t *q16, int lane, uint32_t q0, uint32_t cnt, uint64_t gidx, const Rng &rng, int g8,
                                                    uint32_t ivpack, const double (*s_bnd)[16]) {
    using L = CoopLds<ND, false>;
    const uint32_t q = q0 + (uint32_t)lane;
    if (q < cnt) {
        const uint32_t d = q16[q];
        const uint32_t owner = d & 63u, sb = (d >> 6) & 31u, kind = (d >> 11) & 1u, b1 = ((d >> 12) & 15u) + 1u;
        const uint32_t s = MSBFIRST ? (sb ^ 7u) : sb;
        const uint32_t k = s >> 3, j = s & 7u;
        const uint64_t go = gidx - (uint64_t)lane + (uint64_t)owner;
        const uint32_t iv = (ivpack >> (8u * k)) & 0xFFu;
        const uint32_t sec = kind ? EMGPU_SEC_DEDISC_TRANS : EMGPU_SEC_DEDISC_RES;
        const uint4 r4 = philox4x32((uint32_t)go, (uint32_t)(go >> 32), W.attempt[owner],
                                       (sec << 28) | (iv << 20) | (uint32_t)(2 * g8 + (int)(j >> 2)), rng.k0, rng.k1);
        const uint32_t w = j & 3u;
        const uint32_t x = w == 0 ? r4.x : (w == 1 ? r4.y : (w == 2 ? r4.z : r4.w));
        double v;
        {
#pragma clang fp contract(off)
            const double a = s_bnd[k][b1 - 1u], b = s_bnd[k][b1];
            const double dd = b - a;
            const double mm = dd * uniform32(x);
            v = a + mm;
        }
        W.res[owner * L::kStride + s] = (float)v;
    }
}
// (clears the lane's result slots itself: while the requests are written, the slots -- not yet anybody's -- hold the lane's packed bins, so
// that the owner's look-up of a request's bin is one byte read instead of a select over the variables' registers)
template <int ND, bool MSBFIRST>
__device__ __forceinline__ void coop_dedisc_sc(CoopLds<ND, false> &W, int lane, uint64_t gidx, const Rng &rng, int g8,
                                               uint32_t needmask, uint32_t kindmask, const uint32_t (&pbA)[ND], const uint32_t (&pbB)[ND],
                                               const uint32_t (&ivar)[ND], const double (*s_bnd)[16]) {
    using L = CoopLds<ND, false>;
    constexpr uint32_t kCap = 256u;
    static_assert(sizeof(W.queue) >= kCap * sizeof(uint16_t), "queue bytes");
    uint32_t ivpack = 0u; // wave-uniform byte table of the variables' RNG ids
#pragma unroll
    for (int q = 0; q < ND; q++) ivpack |= ivar[q] << (8 * q);
    uint32_t m = needmask;
    float4 *const rp = reinterpret_cast<float4 *>(&W.res[lane * L::kStride]);
    if (__ballot(m != 0u) == 0ull) {
#pragma unroll
        for (int q = 0; q < 2 * ND; q++) rp[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const uint32_t c = (uint32_t)__popc(m);
    const uint32_t inc = wave_inclusive_add(c);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    uint16_t *const q16 = reinterpret_cast<uint16_t *>(W.queue);
    uint32_t a = inc - c;
    EMGPU_COUNT(4, lane, total);
    if (total <= kCap) {
        // the usual case, one round
        EMGPU_COUNT(1, lane, 1);
        uint2 *const bp = reinterpret_cast<uint2 *>(rp);
#pragma unroll
        for (int k = 0; k < ND; k++) bp[k] = make_uint2(pbA[k], pbB[k]);    // byte s = 8 k + j: the bin of (variable k, second j)
        const uint8_t *const mybins = reinterpret_cast<const uint8_t *>(rp);
        uint16_t *qp = q16 + a;
        while (__ballot(m != 0u) != 0ull) {
            EMGPU_COUNT(2, lane, 1);
            if (m != 0u) {
                const uint32_t sb = (uint32_t)__ffs((int)m) - 1u;
                const uint32_t b1 = mybins[MSBFIRST ? (sb ^ 7u) : sb];
                *qp++ = (uint16_t)((uint32_t)lane | (sb << 6) | (((kindmask >> sb) & 1u) << 11) | ((b1 - 1u) << 12));
                m &= m - 1u;
            }
        }
#pragma unroll
        for (int q = 0; q < 2 * ND; q++) rp[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        wave_sync();
        for (uint32_t q0 = 0u; q0 < total; q0 += 64u) {
            EMGPU_COUNT(3, lane, 1);
            coop_worker_pass_sc<ND, MSBFIRST>(W, q16, lane, q0, total, gidx, rng, g8, ivpack, s_bnd);
        }
        wave_sync();
        return;
    }
    // more than kCap requests in the wave-block: rounds of kCap positions, the bins looked up in the registers
#pragma unroll
    for (int q = 0; q < 2 * ND; q++) rp[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t aend = inc;
    for (uint32_t rb = 0u; rb < total; rb += kCap) {
        EMGPU_COUNT(1, lane, 1);
        const uint32_t lim = min(aend, rb + kCap);   // a lane is active while a < lim
        while (__ballot(a < lim) != 0ull) {
            EMGPU_COUNT(2, lane, 1);
            if (a < lim) {
                const uint32_t sb = (uint32_t)__ffs((int)m) - 1u;
                const uint32_t s = MSBFIRST ? (sb ^ 7u) : sb;
                const uint32_t k = s >> 3, j = s & 7u;
                const uint32_t wd = (j & 4u) ? pick_word<ND>(pbB, k) : pick_word<ND>(pbA, k);
                const uint32_t b1 = (wd >> (8u * (j & 3u))) & 0xFFu;
                q16[a - rb] = (uint16_t)((uint32_t)lane | (sb << 6) | (((kindmask >> sb) & 1u) << 11) | ((b1 - 1u) << 12));
                a++;
                m &= m - 1u;
            }
        }
        wave_sync();
        const uint32_t cnt = min(total - rb, kCap);
        for (uint32_t q0 = 0u; q0 < cnt; q0 += 64u) {
            EMGPU_COUNT(3, lane, 1);
            coop_worker_pass_sc<ND, MSBFIRST>(W, q16, lane, q0, cnt, gidx, rng, g8, ivpack, s_bnd);
        }
        wave_sync();
    }
}

// Forward fill of variable k across the 8 seconds of the block (value changes only where a draw was
// due or the bin became the zero bin) and the two 4-second output blocks of the time-blocked SoA.
template <int ND>
__device__ __forceinline__ void coop_fill_store(const CoopLds<ND> &W, int lane, int k, int g8, int T, int G4, bool valid,
                                                uint32_t need8, uint32_t zero8, float &cval, uint32_t pbA, uint32_t pbB,
                                                uint32_t nd, uint32_t slot, int64_t i, int64_t n, uint32_t *dyn_bin, float *dyn_val) {
    const float4 *rp = reinterpret_cast<const float4 *>(&W.res[lane * CoopLds<ND>::kStride]);
    const float4 ra = rp[2 * k], rb = rp[2 * k + 1];
    const float r[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
    float pv[8];
    float v = cval;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        v = ((need8 >> j) & 1u) ? r[j] : (((zero8 >> j) & 1u) ? 0.f : v);
        pv[j] = (8 * g8 + j < T) ? v : 0.f;
    }
    cval = v;
    if (valid) {
        const size_t o = ((size_t)(2 * g8) * nd + slot) * (size_t)n + (size_t)i;
        if (dyn_bin) dyn_bin[o] = pbA;
        if (dyn_val) reinterpret_cast<float4 *>(dyn_val)[o] = make_float4(pv[0], pv[1], pv[2], pv[3]);
        if (2 * g8 + 1 < G4) {
            const size_t o2 = o + (size_t)nd * (size_t)n;
            if (dyn_bin) dyn_bin[o2] = pbB;
            if (dyn_val) reinterpret_cast<float4 *>(dyn_val)[o2] = make_float4(pv[4], pv[5], pv[6], pv[7]);
        }
    }
}

// The same for MSB-first flag streams, written as carry arithmetic: fill8 (bit 7-j <-> second j) marks
// the seconds whose value is replaced by the lane's result slot -- a dediscretize draw, or 0 for a
// change into the zero bin (coop_zero_results cleared the slots before the workers wrote).
// One v_add_co (shift the stream, carry = the flag) + one v_cndmask per second; two wait states
// between the VCC write and its read (see emgpu_kernels_fast.hip).
template <int ND, bool LB>
__device__ __forceinline__ void coop_zero_results(CoopLds<ND, LB> &W, int lane) {
    float4 *rp = reinterpret_cast<float4 *>(&W.res[lane * CoopLds<ND, LB>::kStride]);
#pragma unroll
    for (int q = 0; q < 2 * ND; q++) rp[q] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// once per trajectory: the lane's global sample index for the workers (CoopLds<ND, true>::kGidx)
template <int ND>
__device__ __forceinline__ void coop_publish_gidx(CoopLds<ND, true> &W, int lane, uint64_t gidx) {
    *reinterpret_cast<uint64_t *>(&W.res[lane * CoopLds<ND, true>::kStride + CoopLds<ND, true>::kGidx]) = gidx;
}

// publish the packed bins of this lane's block for the workers (CoopLds<ND, true>)
template <int ND>
__device__ __forceinline__ void coop_publish_bins(CoopLds<ND, true> &W, int lane, const uint32_t (&pbA)[ND], const uint32_t (&pbB)[ND]) {
    uint2 *bp = reinterpret_cast<uint2 *>(&W.res[lane * CoopLds<ND, true>::kStride + CoopLds<ND, true>::kBins]);
#pragma unroll
    for (int k = 0; k < ND; k++) bp[k] = make_uint2(pbA[k], pbB[k]);
}

// BOTH: dyn_bin and dyn_val are both there (the launch code checked): no null tests -- eight wave-uniform branches less per variable
// and block, and the basic blocks they cut the fill of the variables into (k_dbn_step2 on cor_v1: 12.9 -> 11.7 ms)
// SADDR (with BOTH): the stores take the row segment's address as a scalar base + the thread's 32-bit byte offset (no 64-bit vector add
// per store): -1 % on k_uncor_fast; the four scalar pairs it keeps alive cost the 4-variable k_dbn_step2 6 %, which does not ask for it
template <int ND, bool LB, bool BOTH = false, bool SADDR = false>
__device__ __forceinline__ void coop_fill_store_msb(const CoopLds<ND, LB> &W, int lane, int k, int g8, int T, int G4, bool valid,
                                                    uint32_t fill8, float &cval, uint32_t pbA, uint32_t pbB,
                                                    uint32_t nd, uint32_t slot, int64_t i_wg, uint32_t tid, int64_t n,
                                                    uint32_t *dyn_bin, float *dyn_val) {
    const float4 *rp = reinterpret_cast<const float4 *>(&W.res[lane * CoopLds<ND, LB>::kStride]);
    const float4 ra = rp[2 * k], rb = rp[2 * k + 1];
    const float r[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
    float pv[8];
    uint32_t f = fill8 << 24;
#pragma unroll
    for (int j = 0; j < 8; j++)
        asm("v_add_co_u32 %0, vcc, %0, %0\n\ts_nop 1\n\tv_cndmask_b32 %1, %2, %3, vcc"
            : "+v"(f), "=v"(pv[j]) : "v"(j ? pv[j - 1] : cval), "v"(r[j]) : "vcc");
    cval = pv[7];
    if (8 * g8 + 7 >= T) { // the last block of the trajectory: nothing past T
#pragma unroll
        for (int j = 0; j < 8; j++) pv[j] = (8 * g8 + j < T) ? pv[j] : 0.f;
    }
    if (valid) {
        // wave-uniform base (scalar registers) + the thread's 32-bit offset: no per-store 64-bit vector arithmetic
        const size_t o = ((size_t)(2 * g8) * nd + slot) * (size_t)n + (size_t)i_wg;
      if constexpr (BOTH && SADDR) {
        typedef float v4f_t __attribute__((ext_vector_type(4)));
        const uint32_t *bb = dyn_bin + o;
        const float4 *vb = reinterpret_cast<const float4 *>(dyn_val) + o;
        const uint32_t o4 = tid * 4u, o16 = tid * 16u;
        const v4f_t a = {pv[0], pv[1], pv[2], pv[3]}, b = {pv[4], pv[5], pv[6], pv[7]};
        asm volatile("global_store_dword %0, %1, %2" : : "v"(o4), "v"(pbA), "s"(bb) : "memory");
        asm volatile("global_store_dwordx4 %0, %1, %2" : : "v"(o16), "v"(a), "s"(vb) : "memory");
        if (2 * g8 + 1 < G4) {
            const size_t o2 = (size_t)nd * (size_t)n;
            const uint32_t *bb2 = bb + o2;
            const float4 *vb2 = vb + o2;
            asm volatile("global_store_dword %0, %1, %2" : : "v"(o4), "v"(pbB), "s"(bb2) : "memory");
            asm volatile("global_store_dwordx4 %0, %1, %2" : : "v"(o16), "v"(b), "s"(vb2) : "memory");
        }
      } else if constexpr (BOTH) {
        uint32_t *__restrict__ bb = dyn_bin + o;
        float4 *__restrict__ vb = reinterpret_cast<float4 *>(dyn_val) + o;
        bb[tid] = pbA;
        vb[tid] = make_float4(pv[0], pv[1], pv[2], pv[3]);
        if (2 * g8 + 1 < G4) {
            const size_t o2 = (size_t)nd * (size_t)n;
            bb[o2 + tid] = pbB;
            vb[o2 + tid] = make_float4(pv[4], pv[5], pv[6], pv[7]);
        }
      } else {
        uint32_t *__restrict__ bb = dyn_bin ? dyn_bin + o : nullptr;
        float4 *__restrict__ vb = dyn_val ? reinterpret_cast<float4 *>(dyn_val) + o : nullptr;
        if (bb) bb[tid] = pbA;
        if (vb) vb[tid] = make_float4(pv[0], pv[1], pv[2], pv[3]);
        if (2 * g8 + 1 < G4) {
            const size_t o2 = (size_t)nd * (size_t)n;
            if (bb) bb[o2 + tid] = pbB;
            if (vb) vb[o2 + tid] = make_float4(pv[4], pv[5], pv[6], pv[7]);
        }
      }
    }
}

} // namespace emgpu
