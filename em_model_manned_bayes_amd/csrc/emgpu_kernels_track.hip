// emgpu_kernels_track.hip -- sample2track.m:183-237: the 1 Hz dead-reckoning track that the
// reference builds from the files em_sample wrote, and its rejection tests, for n trajectories.
// One lane = one trajectory.  Two input forms:
//   DENSE  -- the sampler's device output as it lies in HBM (init_val rows + the time-blocked
//             dyn_val float4 blocks): the device consumer of the hot path;
//   PLANAR -- f64 columns parsed from initial.txt / transition.txt, laid out [T][3][n].
// Per trajectory and second: one f64 sin/cos pair (quadrant-exact like cosd/sind, Horner sums), 5 adds, 5 multiplies;
// reads 12 B (DENSE) and writes 24 B of f64 track.  Bound: HBM.
#include <hip/hip_runtime.h>

#include "emgpu_device.h"
#include "emgpu_launch.h"
#include "emgpu_plan.h"

namespace emgpu {

// cosd / sind: MATLAB's reduction in degrees + Horner sums on the reduced angle (emgpu_device.h; within 2 ulp of the library)
__device__ __forceinline__ void k_sincosd(double deg, double &s, double &c) { sincosd_small(deg, s, c); }

template <bool DENSE>
__global__ void __launch_bounds__(256) k_sample2track(const EmgpuTrackRun A) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n) return;
    const size_t n = (size_t)A.n;
    // sample2track.m:184-189 (initial position) and :126-128 (unit conversion of the initial columns)
    double z = DENSE ? (double)A.alt0_f[i] : A.alt0_d[i];
    double sp = (DENSE ? (double)A.speed0_f[i] : A.speed0_d[i]) * A.ur_speed;
    double x = 0.0, y = 0.0, hd = 0.0;
    const double vmin = A.min_speed * A.ur_speed, vmax = A.max_speed * A.ur_speed;   // :138-139
    double lo = sp, hi = sp;
    uint32_t fl = 0u;
    if (z < 0.0) fl |= 1u;                               // :234-237 CFIT
    if (sp <= vmin || sp >= vmax) fl |= 2u;              // :240
    if (A.xyz) { A.xyz[0 * n + i] = x; A.xyz[1 * n + i] = y; A.xyz[2 * n + i] = z; }
    const float4 *dv4 = reinterpret_cast<const float4 *>(A.dyn_val);
    for (int t0 = 0; t0 < A.T; t0 += 4) {
        double uvr[4], uac[4], utr[4];
        if (DENSE) {
            const size_t b = (size_t)(t0 >> 2) * (size_t)A.nd;
            const float4 a = dv4[(b + A.s_vr) * n + i], c = dv4[(b + A.s_acc) * n + i], d = dv4[(b + A.s_tr) * n + i];
            uvr[0] = a.x; uvr[1] = a.y; uvr[2] = a.z; uvr[3] = a.w;
            uac[0] = c.x; uac[1] = c.y; uac[2] = c.z; uac[3] = c.w;
            utr[0] = d.x; utr[1] = d.y; utr[2] = d.z; utr[3] = d.w;
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int t = min(t0 + q, A.T - 1);
                uvr[q] = A.upd[((size_t)t * 3 + 0) * n + i];
                uac[q] = A.upd[((size_t)t * 3 + 1) * n + i];
                utr[q] = A.upd[((size_t)t * 3 + 2) * n + i];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int t = t0 + q;
            if (t >= A.T) break;
            // :131-133 unit conversion of the transition columns, :201-212 the update
            const double dz = uvr[q] * A.ur_vertrate, dsp = uac[q] * A.ur_speed, dhd = utr[q] * A.ur_heading;
            double sh, ch;
            k_sincosd(hd, sh, ch);
            const double xn = x + sp * ch, yn = y + sp * sh;
            z = z + dz; sp = sp + dsp; hd = hd + dhd;
            x = xn; y = yn;
            if (z < 0.0) fl |= 1u;
            if (sp <= vmin || sp >= vmax) fl |= 2u;
            lo = sp < lo ? sp : lo; hi = sp > hi ? sp : hi;
            if (A.xyz) {
                const size_t o = (size_t)(t + 1) * 3 * n + i;
                A.xyz[o] = x; A.xyz[o + n] = y; A.xyz[o + 2 * n] = z;
            }
        }
    }
    if (A.flags) A.flags[i] = (uint8_t)fl;
    if (A.vmm) { A.vmm[i] = lo; A.vmm[n + i] = hi; }
}

hipError_t launch_sample2track(const EmgpuTrackRun &A, bool dense, hipStream_t s, const char **name) {
    *name = dense ? "k_sample2track<dense>" : "k_sample2track<planar>";
    if (A.n <= 0) return hipSuccess;
    const int64_t blocks = (A.n + 255) / 256;
    if (dense) hipLaunchKernelGGL((k_sample2track<true>), dim3((unsigned)blocks), dim3(256), 0, s, A);
    else hipLaunchKernelGGL((k_sample2track<false>), dim3((unsigned)blocks), dim3(256), 0, s, A);
    return hipGetLastError();
}

} // namespace emgpu
