// tools/ubench/f64_ubench.hip -- issue cost of the f64 vector instructions the track / terminal kernels lean on (gfx950): 8 independent
// chains per lane, 8 waves per SIMD, inline asm.  Prints SIMD-cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int KIND>
__global__ void __launch_bounds__(256) k(double *out, double a0, double b0, int iters) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    double x0 = a0 + i, x1 = a0 + 2 * i, x2 = a0 + 3 * i, x3 = a0 + 4 * i, x4 = a0 + 5 * i, x5 = a0 + 6 * i, x6 = a0 + 7 * i, x7 = a0 + 8 * i;
    double c = b0 + 1e-9 * i, d = a0 * 0.5 + 1e-7 * i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#define FMA(n) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x##n) : "v"(c), "v"(d));
#define FMAS(n) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x##n) : "v"(c), "s"(b0));
#define MUL(n) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x##n) : "v"(c));
#define ADD(n) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x##n) : "v"(c));
#define RCP(n) asm volatile("v_rcp_f64 %0, %0" : "+v"(x##n));
#define SQRT(n) asm volatile("v_sqrt_f64 %0, %0" : "+v"(x##n));
#define RNDNE(n) asm volatile("v_rndne_f64 %0, %0" : "+v"(x##n));
#define CMPSEL(n) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(x##n), "v"(c) : "vcc");
#define MIN(n) asm volatile("v_min_f64 %0, %0, %1" : "+v"(x##n) : "v"(c));
#define CVTI(n) { int q; asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(q) : "v"(x##n)); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(x##n) : "v"(q)); }
#define LDEXP(n) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(x##n));
#define MOV64(n) asm volatile("v_mov_b64 %0, %1" : "=v"(x##n) : "v"(c));
            if (KIND == 0) { REP8(FMA) }
            if (KIND == 1) { REP8(FMAS) }
            if (KIND == 2) { REP8(MUL) }
            if (KIND == 3) { REP8(ADD) }
            if (KIND == 4) { REP8(RCP) }
            if (KIND == 5) { REP8(SQRT) }
            if (KIND == 6) { REP8(RNDNE) }
            if (KIND == 7) { REP8(CMPSEL) }
            if (KIND == 8) { REP8(MIN) }
            if (KIND == 9) { REP8(CVTI) }
            if (KIND == 10) { REP8(LDEXP) }
            if (KIND == 11) { REP8(MOV64) }
        }
    }
    out[i] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int KIND>
static void run(const char *name, int per_rep) {
    const int blocks = 256 * 8, iters = 2000;   // 8 waves per SIMD on 256 CUs
    double *out; hipMalloc(&out, (size_t)blocks * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.5, 0.999, 10);
    hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.5, 0.999, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)blocks * 4 / (256 * 4) * iters * 8 * 8 * per_rep;   // waves per SIMD x instructions per wave
    printf("%-28s %.2f cycles per wave-instruction (at 2.4 GHz)\n", name, ms * 1e-3 * 2.4e9 / insts_per_simd);
    hipFree(out);
}
int main() {
    run<0>("v_fma_f64", 1); run<1>("v_fma_f64 (sgpr operand)", 1); run<2>("v_mul_f64", 1); run<3>("v_add_f64", 1); run<4>("v_rcp_f64", 1);
    run<5>("v_sqrt_f64", 1); run<6>("v_rndne_f64", 1); run<7>("v_cmp_lt_f64", 1); run<8>("v_min_f64", 1); run<9>("cvt i32<->f64 pair", 2);
    run<10>("v_ldexp_f64", 1); run<11>("v_mov_b64", 1);
    return 0;
}
