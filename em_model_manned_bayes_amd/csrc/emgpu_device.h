// emgpu_device.h -- device-side primitives shared by every kernel: Philox4x32-10, the uniform,
// the threshold draw and dediscretize.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "emgpu_plan.h"

namespace emgpu {

// Philox4x32-10 (Salmon et al. SC'11).  The key is wave-uniform (the seed), so the key schedule
// lives in SGPRs / literals; each round is two v_mad_u64_u32 (32x32->64) and four XORs.
__device__ __forceinline__ uint4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ (k0 + (uint32_t)r * 0x9E3779B9u);
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ (k1 + (uint32_t)r * 0xBB67AE85u);
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
        return philox4x32_10(c0, c1, attempt, (section << 28) | (a << 20) | blk, k0, k1);
    }
};

// word w of a block; w is compile-time or wave-uniform at every call site
__device__ __forceinline__ uint32_t word_of(const uint4 &v, int w) { return w == 0 ? v.x : (w == 1 ? v.y : (w == 2 ? v.z : v.w)); }

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

} // namespace emgpu
