// tools/ubench/rng_ubench.hip -- VALU cost of counter-based RNG candidates on gfx950.
// Each thread runs ITER dependent-free calls; reports ns per call per wave-slot and derived
// "lane-calls/s" for the whole chip.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int ROUNDS>
__device__ __forceinline__ uint4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ (k0 + (uint32_t)r * 0x9E3779B9u);
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ (k1 + (uint32_t)r * 0xBB67AE85u);
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
    }
    return make_uint4(c0, c1, c2, c3);
}

__device__ __forceinline__ uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
template <int ROUNDS>
__device__ __forceinline__ uint4 threefry(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t k0, uint32_t k1) {
    // Threefry4x32 with key {k0,k1,0,0}
    const int R[8][2] = {{10, 26}, {11, 21}, {13, 27}, {23, 5}, {6, 20}, {17, 11}, {25, 10}, {18, 20}};
    uint32_t ks[5] = {k0, k1, 0u, 0u, 0x1BD11BDAu ^ k0 ^ k1};
    x0 += ks[0]; x1 += ks[1]; x2 += ks[2]; x3 += ks[3];
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        if (r % 2 == 0) { x0 += x1; x1 = rotl(x1, R[r % 8][0]); x1 ^= x0; x2 += x3; x3 = rotl(x3, R[r % 8][1]); x3 ^= x2; }
        else { x0 += x3; x3 = rotl(x3, R[r % 8][0]); x3 ^= x0; x2 += x1; x1 = rotl(x1, R[r % 8][1]); x1 ^= x2; }
        if (r % 4 == 3) {
            const int s = r / 4 + 1;
            x0 += ks[s % 5]; x1 += ks[(s + 1) % 5]; x2 += ks[(s + 2) % 5]; x3 += ks[(s + 3) % 5] + s;
        }
    }
    return make_uint4(x0, x1, x2, x3);
}

template <int KIND, int ROUNDS, int ITER, int LDSB = 0>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t k0, uint32_t k1) {
    __shared__ uint32_t pad[LDSB / 4 + 1];
    if (LDSB > 0 && k0 == 0xdeadbeefu) pad[threadIdx.x] = k1;   // keep the allocation (occupancy limiter)
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
#pragma unroll 4
    for (int it = 0; it < ITER; it++) {
        uint4 v = KIND == 0 ? philox<ROUNDS>(i, 0, 0, it, k0, k1) : threefry<ROUNDS>(i, 0, 0, it, k0, k1);
        acc += (v.x < 0x1234567u) + (v.y < 0x2345678u) + (v.z < 0x3456789u) + (v.w < 0x456789Au);
    }
    out[i] = acc + ((LDSB > 0 && k0 == 0xdeadbeefu) ? pad[(threadIdx.x + 1) & 255] : 0u);
}

template <int KIND, int ROUNDS, int LDSB = 0>
void run(const char *name) {
    const int ITER = 512, blocks = 256 * 16;
    uint32_t *d;
    hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<KIND, ROUNDS, ITER, LDSB><<<blocks, 256>>>(d, 1, 2);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) k<KIND, ROUNDS, ITER, LDSB><<<blocks, 256>>>(d, 1, 2);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double calls = 5.0 * blocks * 256.0 * ITER;
    double rate = calls / (ms * 1e-3);
    // lane-slot model: 256 CU * 4 SIMD * 32 lanes * 2.4e9 = 78.6e12 lane-instr/s
    printf("%-18s %8.3f ms  %.3e calls/s  %.3e words/s  ~%.1f lane-slots/call (at 78.6e12/s)\n", name, ms / 5, rate, rate * 4, 78.6e12 / rate);
    hipFree(d);
}

// xoshiro128++ (Blackman & Vigna 2018): sequential, 4 words of state, ~10 full-rate VALU ops per word
template <int ITER>
__global__ void __launch_bounds__(256) kx(uint32_t *out, uint32_t k0, uint32_t k1) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint4 st = philox<10>(i, 0, 0, 0, k0, k1);
    uint32_t s0 = st.x, s1 = st.y, s2 = st.z, s3 = st.w, acc = 0;
#pragma unroll 8
    for (int it = 0; it < ITER * 4; it++) {
        const uint32_t r = rotl(s0 + s3, 7) + s0;
        const uint32_t t = s1 << 9;
        s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3; s2 ^= t; s3 = rotl(s3, 11);
        acc += (r < 0x1234567u);
    }
    out[i] = acc;
}
void runx(const char *name) {
    const int ITER = 512, blocks = 256 * 16;
    uint32_t *d;
    hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    kx<ITER><<<blocks, 256>>>(d, 1, 2);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) kx<ITER><<<blocks, 256>>>(d, 1, 2);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double words = 5.0 * blocks * 256.0 * ITER * 4;
    printf("%-18s %8.3f ms  %.3e words/s  ~%.1f lane-slots/word\n", name, ms / 5, words / (ms * 1e-3), 78.6e12 / (words / (ms * 1e-3)));
    hipFree(d);
}

// three interleaved xoshiro128++ streams per lane (one per dynamic variable): ILP for the serial chains
struct Xo { uint32_t s0, s1, s2, s3; };
__device__ __forceinline__ uint32_t xo_next(Xo &x) {
    const uint32_t r = rotl(x.s0 + x.s3, 7) + x.s0;
    const uint32_t t = x.s1 << 9;
    x.s2 ^= x.s0; x.s3 ^= x.s1; x.s1 ^= x.s2; x.s0 ^= x.s3; x.s2 ^= t; x.s3 = rotl(x.s3, 11);
    return r;
}
template <int ITER>
__global__ void __launch_bounds__(256) kx3(uint32_t *out, uint32_t k0, uint32_t k1) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint4 a = philox<10>(i, 0, 0, 0, k0, k1), b = philox<10>(i, 0, 0, 1, k0, k1), c = philox<10>(i, 0, 0, 2, k0, k1);
    Xo x0{a.x, a.y, a.z, a.w}, x1{b.x, b.y, b.z, b.w}, x2{c.x, c.y, c.z, c.w};
    uint32_t acc = 0;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int w = 0; w < 8; w++) {   // one 8-second block: 8 words per variable
            acc += (xo_next(x0) < 0x1234567u) + (xo_next(x1) < 0x2345678u) + (xo_next(x2) < 0x3456789u);
        }
    }
    out[i] = acc;
}
void runx3(const char *name) {
    const int ITER = 128, blocks = 256 * 16;
    uint32_t *d;
    hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    kx3<ITER><<<blocks, 256>>>(d, 1, 2);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) kx3<ITER><<<blocks, 256>>>(d, 1, 2);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double words = 5.0 * blocks * 256.0 * ITER * 24;
    printf("%-18s %8.3f ms  %.3e words/s  ~%.1f lane-slots/word\n", name, ms / 5, words / (ms * 1e-3), 78.6e12 / (words / (ms * 1e-3)));
    hipFree(d);
}

int main() {
    runx3("xoshiro128++ x3");
    runx("xoshiro128++");
    run<0, 10>("philox4x32-10");
    run<0, 10, 20000>("philox-10 8blk/CU");   // 8 waves/SIMD
    run<0, 10, 40000>("philox-10 4blk/CU");   // 4 waves/SIMD
    run<0, 10, 80000>("philox-10 2blk/CU");   // 2 waves/SIMD
    run<0, 10, 160000>("philox-10 1blk/CU");  // 1 wave/SIMD
    run<0, 7>("philox4x32-7");
    run<1, 20>("threefry4x32-20");
    run<1, 12>("threefry4x32-12");
    return 0;
}
