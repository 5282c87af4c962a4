// tools/ubench/exec_ubench.hip -- what a vector instruction costs when only some lanes of the wave are active (gfx950): the same chain of
// v_fma_f64 / v_mad_u64_u32 / v_xor_b32 under exec masks with 64, 32 (lower half), 16 (one quarter), 5 scattered and 1 active lane(s),
// 4 waves per SIMD.  Prints SIMD cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int KIND>
__global__ void __launch_bounds__(256) k(double *out, uint64_t mask, int iters, double a) {
    const int lane = threadIdx.x & 63;
    double x0 = a + lane, x1 = a * 2 + lane, x2 = a * 3 + lane, x3 = a * 5 + lane;
    uint32_t u0 = lane * 3 + 1, u1 = lane * 5 + 2, u2 = lane * 7 + 3, u3 = lane * 11 + 4;
    if ((mask >> lane) & 1ull) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                if (KIND == 0) { asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(x0) : "v"(x1)); asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(x2) : "v"(x3));
                                 asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(x1) : "v"(x0)); asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(x3) : "v"(x2)); }
                if (KIND == 1) { asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u0) : "v"(u1)); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u2) : "v"(u3));
                                 asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u1) : "v"(u0)); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u3) : "v"(u2)); }
                if (KIND == 2) { uint64_t p, q;
                                 asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, 0" : "=v"(p) : "v"(u0), "v"(u1) : "s10", "s11"); u0 = (uint32_t)(p >> 32) ^ (uint32_t)p;
                                 asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, 0" : "=v"(q) : "v"(u2), "v"(u3) : "s10", "s11"); u2 = (uint32_t)(q >> 32) ^ (uint32_t)q;
                                 asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, 0" : "=v"(p) : "v"(u1), "v"(u0) : "s10", "s11"); u1 = (uint32_t)(p >> 32);
                                 asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, 0" : "=v"(q) : "v"(u3), "v"(u2) : "s10", "s11"); u3 = (uint32_t)(q >> 32); }
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + u0 + u1 + u2 + u3;
}
template <int KIND> double run(const char *name, uint64_t mask, double *d_out) {
    int dev = 0, cus = 0; hipGetDevice(&dev); hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int blocks = cus * 4, iters = 2000;   // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, d_out, mask, 10, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, d_out, mask, iters, 1.0);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, dev);
    // per SIMD: 4 waves x iters x 16 x 4 (+ 8 for the xors of KIND 2: reported as is) instructions
    const double inst = 4.0 * iters * 16 * 4, cyc = ms * 1e-3 * clk * 1e3 / inst;
    printf("%-14s mask %016llx : %.2f cycles per wave-instruction (at the nominal %d MHz)\n", name, (unsigned long long)mask, cyc, clk / 1000);
    return cyc;
}
int main() {
    double *d_out; hipMalloc(&d_out, 256 * 4096 * sizeof(double));
    const uint64_t masks[] = {~0ull, 0xFFFFFFFFull, 0xFFFFull, 0x0000000100010101ull | (1ull << 40), 0x1111111111111111ull, 1ull};
    for (uint64_t m : masks) { run<0>("v_fma_f64", m, d_out); }
    for (uint64_t m : masks) { run<1>("v_xor_b32", m, d_out); }
    for (uint64_t m : masks) { run<2>("v_mad_u64_u32", m, d_out); }
    return 0;
}
