// tools/ubench/store_ubench.hip -- what the benchmark's OUTPUT alone costs on this box: kernels that do nothing but store the
// dense trace of 10 M trajectories x 240 s x 3 variables (36.35 GB: u32 packed bins + float4 values per 4-second block, variable and
// trajectory -- the layout of DESIGN.md section 4), in the order k_uncor_fast stores it (a workgroup walks its 256 columns through the
// 60 four-second blocks) and, for comparison, as one flat fill.  Prints ms and TB/s per variant (median of 20 launches).
//   0 twin       : the sampler's store pattern, one lane = one trajectory, 30 iterations x 3 variables x (2 x u32 + 2 x float4)
//   1 twin + nt  : the same with nontemporal stores
//   2 flat       : every workgroup fills one contiguous 256 x 16 B x 8 chunk after another (grid-stride)
//   3 flat + nt
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

template <bool NT>
__device__ __forceinline__ void st4(float4 *p, float4 v) {
    if (NT) {
        __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y);
        __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w);
    } else *p = v;
}
template <bool NT>
__device__ __forceinline__ void st1(uint32_t *p, uint32_t v) {
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

template <bool NT>
__global__ void __launch_bounds__(256, 4) k_twin(uint32_t *dyn_bin, float4 *dyn_val, int64_t n, int64_t ld, int G4, uint32_t seedish) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t x = (uint32_t)i * 2654435761u + seedish;
    for (int g8 = 0; g8 < G4 / 2; g8++) {
        x = x * 1664525u + 1013904223u;   // (something to store that the compiler cannot hoist)
        const float f = __uint_as_float(0x3f800000u | (x >> 9));
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const size_t o = ((size_t)(2 * g8) * 3 + k) * (size_t)ld + (size_t)i;
            st1<NT>(dyn_bin + o, x + k);
            st4<NT>(dyn_val + o, make_float4(f, f + k, f, f));
            const size_t o2 = o + (size_t)3 * (size_t)ld;
            st1<NT>(dyn_bin + o2, x ^ k);
            st4<NT>(dyn_val + o2, make_float4(f, f, f + k, f));
        }
    }
}

// the same stores in a TILE-major layout: everything a workgroup (256 trajectories) writes is one contiguous range that it fills from
// front to back ([tile][4-second block][variable][256]): does the order of the planes cost anything?
__global__ void __launch_bounds__(256, 4) k_twin_tile(uint32_t *dyn_bin, float4 *dyn_val, int64_t n, int G4, uint32_t seedish) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t x = (uint32_t)i * 2654435761u + seedish;
    const size_t base = (size_t)blockIdx.x * (size_t)G4 * 3 * 256 + threadIdx.x;
    for (int g8 = 0; g8 < G4 / 2; g8++) {
        x = x * 1664525u + 1013904223u;
        const float f = __uint_as_float(0x3f800000u | (x >> 9));
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const size_t o = base + ((size_t)(2 * g8) * 3 + k) * 256;
            dyn_bin[o] = x + k;
            dyn_val[o] = make_float4(f, f + k, f, f);
            dyn_bin[o + 3 * 256] = x ^ k;
            dyn_val[o + 3 * 256] = make_float4(f, f, f + k, f);
        }
    }
}

template <bool NT>
__global__ void __launch_bounds__(256, 4) k_flat(float4 *p, size_t n16, uint32_t seedish) {
    const float f = __uint_as_float(0x3f800000u | (seedish >> 9));
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) st4<NT>(p + i, make_float4(f, f, f, f));
}

template <typename F>
static void run(const char *name, double bytes, F launch) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 5; w++) launch(w);
    hipDeviceSynchronize();
    std::vector<float> ms;
    for (int r = 0; r < 20; r++) {
        hipEventRecord(a); launch(100 + r); hipEventRecord(b);
        hipEventSynchronize(b);
        float t; hipEventElapsedTime(&t, a, b); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    printf("%-12s median %.3f ms  best %.3f ms  %.3f TB/s (median)\n", name, ms[10], ms[0], bytes / ms[10] / 1e9);
}

int main() {
    const int64_t n = 10000000, ld = ((n + 1023) / 1024) * 1024;
    const int T = 240, G4 = T / 4;
    uint32_t *db; float4 *dv;
    const size_t nb = (size_t)G4 * 3 * ld;
    if (hipMalloc(&db, nb * 4) != hipSuccess || hipMalloc(&dv, nb * 16) != hipSuccess) { printf("no device memory\n"); return 1; }
    const double bytes = (double)n * 3 * T * 5;   // 5 B per variable-second: the benchmark's dynamic part (3 600 of its 3 635 B per trajectory)
    const unsigned blocks = (unsigned)((n + 255) / 256);
    run("twin", bytes, [&](int r) { hipLaunchKernelGGL(k_twin<false>, dim3(blocks), dim3(256), 0, 0, db, dv, n, ld, G4, (uint32_t)r); });
    run("twin+nt", bytes, [&](int r) { hipLaunchKernelGGL(k_twin<true>, dim3(blocks), dim3(256), 0, 0, db, dv, n, ld, G4, (uint32_t)r); });
    run("twin tile", bytes, [&](int r) { hipLaunchKernelGGL(k_twin_tile, dim3(blocks), dim3(256), 0, 0, db, dv, n, G4, (uint32_t)r); });
    const size_t n16 = nb;                 // the value planes as one contiguous range (28.8 GB)
    const double fbytes = (double)n16 * 16;
    run("flat", fbytes, [&](int r) { hipLaunchKernelGGL(k_flat<false>, dim3(256 * 32), dim3(256), 0, 0, dv, n16, (uint32_t)r); });
    run("flat+nt", fbytes, [&](int r) { hipLaunchKernelGGL(k_flat<true>, dim3(256 * 32), dim3(256), 0, 0, dv, n16, (uint32_t)r); });
    run("hipMemset", fbytes, [&](int r) { hipMemsetAsync(dv, r & 0xFF, (size_t)fbytes, 0); });
    return 0;
}
