// emgpu_kernels_tfilter.hip -- the two small kernels around k_terminal_propagate that make CorTerminalModel.track
// (@CorTerminalModel/track.m:45-150) a device pipeline:
//   k_terminal_geo     the geometry sample -> the inputs of createEncounter (createEncounter.m:21-49: x0 y0 from distance and
//                      bearing with cosd / sind, z0 v0 heading0 intent, and the trajectory model of each of the four tracks);
//   k_terminal_filter  one lane per encounter: forward + backward tracks merged and ordered in time (:74-84), then the filters
//                      of track.m:62-145 through the static checks of CorTerminalModel.m:117-316 (miss distance and CPA time,
//                      overlap, runway proximity, vertical intent, dynamic limits with the cumulative-turn test).
// em-core's computeVerticalRate / computeHeadingRate are not vendored by the reference ("dynamics unpinned"): forward
// differences over the 1 s samples, the last value repeated, heading differences wrapped to (-pi, pi].  Reference defects
// kept or decided: `isClimb` (track.m:122,134) is undefined there -- read as is_climb; the turn-rate test compares rad/s
// with a deg/s limit (CorTerminalModel.m:296) and the runway distance uses 1.68781 as nm -> ft (:215): both kept.
// Not a hot kernel: it reads the tracks once (20 B per track-second) behind a propagation that costs 100x more.
#include <hip/hip_runtime.h>

#include "../../include/emgpu.h"
#include "emgpu_launch.h"
#include "emgpu_plan.h"

namespace emgpu {

__device__ __forceinline__ double f_sign(double x) { return (double)((x > 0) - (x < 0)); }
__device__ __forceinline__ double f_wrapTo360(double lon) {
    const bool positive = lon > 0;
    lon = lon - floor(lon / 360.0) * 360.0;
    return (lon == 0 && positive) ? 360.0 : lon;
}
__device__ __forceinline__ double f_wrapTo180(double x) { return (x < -180 || 180 < x) ? f_wrapTo360(x + 180) - 180 : x; }
__device__ __forceinline__ double f_wrapToPi(double x) {
    const double pi = 3.14159265358979323846;
    if (x < -pi || pi < x) {
        const bool pos = (x + pi) > 0;
        double y = (x + pi) - floor((x + pi) / (2 * pi)) * (2 * pi);
        if (y == 0 && pos) y = 2 * pi;
        return y - pi;
    }
    return x;
}
__device__ __forceinline__ void f_sincosd(double deg, double &s, double &c) { // MATLAB's reduction in degrees (see emgpu_kernels_term.hip)
    const double n = round(deg / 90.0);
    const double x = (3.14159265358979323846 / 180.0) * (deg - n * 90.0);
    const int m = (int)((long long)n & 3ll);
    const double sx = sin(x), cx = cos(x);
    s = (m == 0) ? sx : ((m == 1) ? cx : ((m == 2) ? -sx : -cx));
    c = (m == 0) ? cx : ((m == 1) ? -sx : ((m == 2) ? -cx : sx));
}

__global__ void __launch_bounds__(256) k_terminal_geo(const EmgpuTGeoRun A) {
#pragma clang fp contract(off)
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= A.n) return;
    double *g = A.geo + e * 12;
    int intent[2];
    for (int a = 0; a < 2; a++) {
        const int32_t *ix = A.idx + 6 * a;
        auto v = [&](int k) { return (double)A.val[(size_t)ix[k] * (size_t)A.n + (size_t)e]; };
        double s, c;
        f_sincosd(v(1), s, c);
        g[6 * a] = v(0) * c; g[6 * a + 1] = v(0) * s; g[6 * a + 2] = v(2); g[6 * a + 3] = v(3); g[6 * a + 4] = v(4); g[6 * a + 5] = v(5);
        intent[a] = (int)v(5);
    }
    int32_t *mo = A.model_of + 4 * e;      // CorTerminalModel.m:84-100 order: own landing / take-off, int landing / take-off / transit; fwd, bck
    mo[0] = 2 * (intent[0] - 1); mo[1] = mo[0] + 1; mo[2] = 4 + 2 * (intent[1] - 1); mo[3] = mo[2] + 1;
}

// the joined, time-ordered track of one aircraft (createEncounter.m:74-84), read in place from k_terminal_propagate's track-major output:
// element i is row lo + i of the aircraft's block, lo = C - (rb - 1) (the backward track's last second), its time lo + i - C
struct Track {
    const float *base;        // row 0 of the aircraft's block [EMGPU_TERMINAL_BLOCK_ROWS(cap)][5]
    int lo, C;
    int rf, rb, n;
    __device__ __forceinline__ double at(int field, int i) const {
        return field == 0 ? (double)(lo + i - C) : (double)base[(size_t)(lo + i) * 5 + (size_t)(field - 1)];
    }
};

__device__ bool f_check_cum_turn(const Track &T, double limit) {                             // CorTerminalModel.m:135-185
    if (!(limit < INFINITY) || T.n < 2) return false;                                        // |cumsum| > inf never holds
    const int m = T.n - 1;
    if (m > 248) return true;   // cannot happen: emgpu_track_terminal_host refuses tmax_s > 122 (a merged track has <= 2 tmax_s + 3 rows); fail closed
    float hd[248];
    short ts[128], te[128];
    int nts = 0, nte = 0;
    double prev = f_wrapTo180(T.at(4, 0));
    for (int i = 0; i < m; i++) {
        const double cur = f_wrapTo180(T.at(4, i + 1));
        hd[i] = (float)(round((cur - prev) * 10.0) / 10.0);                                  // one decimal: exact enough in f32 to test == 0 and the sign
        prev = cur;
    }
    for (int i = 0; i + 1 < m; i++) {
        if (hd[i] == 0.f && hd[i + 1] != 0.f && nts < 128) ts[nts++] = (short)(i + 2);
        if (hd[i] != 0.f && hd[i + 1] == 0.f && nte < 128) te[nte++] = (short)(i + 1);
    }
    if (nts == 0) { ts[0] = 1; nts = 1; }
    if (nte == 0) { te[0] = (short)m; nte = 1; }
    if (nts > nte) te[nte++] = (short)m;
    for (int i = 0; i < nts; i++) {
        double cum = 0, prevh = 0;
        for (int q = ts[i]; q <= te[i]; q++) {
            // re-derive the f64 difference: the f32 copy only located the turns
            const double h = f_wrapTo180(round((f_wrapTo180(T.at(4, q)) - f_wrapTo180(T.at(4, q - 1))) * 10.0) / 10.0);
            if (q > ts[i] && f_sign(h) != f_sign(prevh)) cum = 0;
            cum += h; prevh = h;
            if (fabs(cum) > limit) return true;
        }
    }
    return false;
}

__device__ bool f_check_dynamic_limits(const Track &T, const double *dl, double max_cum_turn, double pitch) { // CorTerminalModel.m:268-316
#pragma clang fp contract(off)
    const int n = T.n;
    if (n <= 1) return false;
    for (int i = 0; i < n; i++) {
        const double z = T.at(3, i), v = T.at(5, i);
        const int k = i + 1 < n ? i : n - 2;
        const double dh = T.at(3, k + 1) - T.at(3, k);
        const double rate = f_wrapToPi(T.at(4, k + 1) * (3.14159265358979323846 / 180.0) - T.at(4, k) * (3.14159265358979323846 / 180.0));
        bool ok = z > 0 && z <= dl[3] && v >= dl[0] && v <= dl[1] && fabs(dh) <= dl[4] && fabs(rate) <= dl[2];
        if (i > 0) {
            const double ratio = fabs(z - T.at(3, i - 1)) / v;
            ok = ok && (ratio <= 1 ? fabs(asin(ratio) * (180.0 / 3.14159265358979323846)) <= pitch : pitch == INFINITY);
        }
        if (!ok) return false;
    }
    return !f_check_cum_turn(T, max_cum_turn);
}

__global__ void __launch_bounds__(256) k_terminal_filter(const EmgpuTFilterRun A) {
#pragma clang fp contract(off)
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= A.n) return;
    const int64_t slot = A.slot ? A.slot[e] : e;
    bool good = true;
    Track T[2];
    for (int a = 0; a < 2; a++) {
        const int rf = A.rows[4 * e + 2 * a], rb = A.rows[4 * e + 2 * a + 1];
        if (rf < 1 || rb < 1) good = false;                    // a track hit the re-draw cap: the attempt is void
        T[a].base = A.tracks + (size_t)(2 * e + a) * (size_t)EMGPU_TERMINAL_BLOCK_ROWS(A.cap) * 5;
        T[a].C = EMGPU_TERMINAL_T0_ROW(A.cap);
        T[a].rf = rf < 1 ? 1 : rf; T[a].rb = rb < 1 ? 1 : rb; T[a].n = T[a].rf + T[a].rb - 1;
        T[a].lo = T[a].C - (T[a].rb - 1);
    }
    const int own_intent = (int)A.geo[e * 12 + 5], int_intent = (int)A.geo[e * 12 + 11];
    double tcpa = 0, hmd = INFINITY, vmd = 0;
    int nc = 0;
    if (good) {
        // getGeneratedMissDistance (CorTerminalModel.m:117-133): the times are consecutive integers
        const double t0 = fmax(T[0].at(0, 0), T[1].at(0, 0)), t1 = fmin(T[0].at(0, T[0].n - 1), T[1].at(0, T[1].n - 1));
        nc = (int)(t1 - t0) + 1;
        if (nc <= 0) good = false;
        else {
            const int ia = (int)(t0 - T[0].at(0, 0)), ib = (int)(t0 - T[1].at(0, 0));
            int best = 0;
            for (int k = 0; k < nc; k++) {
                const double dx = T[0].at(1, ia + k) - T[1].at(1, ib + k), dy = T[0].at(2, ia + k) - T[1].at(2, ib + k);
                const double d = sqrt(dx * dx + dy * dy) * 6076.1154855643;
                if (d < hmd) { hmd = d; best = k; }
            }
            tcpa = T[0].at(0, ia + best); vmd = T[1].at(3, ib + best) - T[0].at(3, ia + best);
            good = fabs(tcpa) <= 10;                                                         // track.m:84-87
        }
    }
    if (good) {
        const bool is_long = nc >= A.min_enc_time_s;                                         // :90-91
        bool close[2], low[2], climb[2], descend[2];
        for (int a = 0; a < 2; a++) {
            close[a] = low[a] = false;
            double zmax = T[a].at(3, 0), zmin = zmax;
            int nclimb = 0, ndesc = 0;
            const int n = T[a].n;
            for (int i = 0; i < n; i++) {
                const double z = T[a].at(3, i);
                const double d_ft = hypot(T[a].at(1, i), T[a].at(2, i)) * 1.68781;           // CorTerminalModel.m:213-215
                if (d_ft <= A.thres_dist_ft) { close[a] = true; if (z <= A.thres_alt_low_ft) low[a] = true; }
                zmax = fmax(zmax, z); zmin = fmin(zmin, z);
                double dh = 0;                                                               // forward difference, last repeated
                if (n > 1) { const int k = i + 1 < n ? i : n - 2; dh = T[a].at(3, k + 1) - T[a].at(3, k); }
                nclimb += dh >= A.thres_vertrate_ft_s; ndesc += dh <= -A.thres_vertrate_ft_s;
            }
            const double thr_time = (zmax - zmin) / A.thres_vertrate_ft_s, pth = fmin(0.2, thr_time / (double)n);   // :244-251
            climb[a] = (double)nclimb / n >= pth; descend[a] = (double)ndesc / n >= pth;
        }
        const bool prox1 = (close[0] && low[0]) || !close[0];                                // track.m:98-112
        const bool prox2 = int_intent == 3 ? !(close[1] && low[1]) : ((close[1] && low[1]) || !close[1]);
        const bool int_ok = int_intent == 1 ? descend[1] : (int_intent == 2 ? climb[1] : true);   // :119-127
        bool own_ok = false;                                                                 // :130-137
        if (own_intent == 1 || own_intent == 2) {
            const double c = own_intent == 1 ? 90.0 : 270.0;
            int ok = 0;
            for (int i = 0; i < T[0].n; i++) { const double h = T[0].at(4, i); ok += h >= c - 30 && h <= c + 30; }
            own_ok = (own_intent == 1 ? descend[0] : climb[0]) && ((double)ok / T[0].n >= .95);
        }
        good = is_long && prox1 && prox2 && own_ok && int_ok;
        if (good) good = f_check_dynamic_limits(T[0], A.dl[0], A.max_cum_turn[0], A.pitch[0]);   // :140-141
        if (good) good = f_check_dynamic_limits(T[1], A.dl[1], A.max_cum_turn[1], A.pitch[1]);
    }
    A.accepted[e] = good ? 1 : 0;
    if (A.attempts) {
        if (good) A.attempts[slot] = A.attempt_no;
        else if (A.last_round) A.attempts[slot] = -1;
    }
    if (!good) return;
    if (A.sample) for (int k = 0; k < A.n_i; k++) A.sample[(size_t)slot * A.n_i + k] = (double)A.val[(size_t)k * (size_t)A.n + (size_t)e];
    if (A.meta) { double *m = A.meta + 4 * slot; m[0] = tcpa; m[1] = hmd; m[2] = vmd; m[3] = (double)nc; }
    for (int a = 0; a < 2; a++) {
        if (A.len) A.len[2 * slot + a] = T[a].n;
        if (A.traj)
            for (int i = 0; i < T[a].n && i < A.cap2; i++) {
                double *q = A.traj + (((size_t)slot * 2 + a) * (size_t)A.cap2 + (size_t)i) * 6;
                for (int f = 0; f < 6; f++) q[f] = T[a].at(f, i);
            }
    }
}

// createEncounter.m:88-89, the stand-in for em-core's local_smooth (EMGPU_FLAG_LOCAL_SMOOTH in include/emgpu.h; oracle: em_local_smooth): one
// wave per joined track, its speed and altitude columns in LDS, row i = the mean of rows i-k..i+k summed in ascending order in f64.
__global__ void __launch_bounds__(256) k_terminal_smooth(float *traj, const int32_t *rows, int64_t n2, int32_t cap) {
#pragma clang fp contract(off)
    // the track's rows as they lie in memory (contiguous: read and written back as whole lines; a pass that touched only the two columns
    // would still move every 32-byte sector of the block both ways)
    __shared__ float s_rows[4][256 * 5];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t a = (int64_t)blockIdx.x * 4 + w;           // aircraft 2e + a
    if (a >= n2) return;
    const int rf = rows[2 * a], rb = rows[2 * a + 1];
    if (rf < 1 || rb < 1) return;                             // a void attempt: nobody reads it
    const int C0 = EMGPU_TERMINAL_T0_ROW(cap), W = 2 * C0;
    const int lo = C0 - (rb - 1), n = rf + rb - 1;
    float *base = traj + ((size_t)a * (size_t)W + (size_t)lo) * 5;
    float *sr = s_rows[w];
    for (int j = lane; j < 5 * n; j += 64) sr[j] = base[j];
    __builtin_amdgcn_wave_barrier();
    float out[4][2];
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const int i = lane + 64 * it;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            int k = c ? 7 : 2;                                // (w - 1) / 2 of w = 15 (altitude), 5 (speed)
            k = i < k ? i : k; k = n - 1 - i < k ? n - 1 - i : k;
            double sum = 0.0;
            if (i < n) for (int q = i - k; q <= i + k; q++) sum += (double)sr[q * 5 + (c ? 2 : 4)];
            out[it][c] = (float)(sum / (double)(2 * k + 1));
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const int i = lane + 64 * it;
        if (i < n) { sr[i * 5 + 4] = out[it][0]; sr[i * 5 + 2] = out[it][1]; }
    }
    __builtin_amdgcn_wave_barrier();
    for (int j = lane; j < 5 * n; j += 64) __builtin_nontemporal_store(sr[j], &base[j]);
}

hipError_t launch_terminal_smooth(float *traj, const int32_t *rows, int64_t n2, int32_t cap, hipStream_t s) {
    if (n2 <= 0) return hipSuccess;
    if (EMGPU_TERMINAL_BLOCK_ROWS(cap) > 256) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_terminal_smooth, dim3((unsigned)((n2 + 3) / 4)), dim3(256), 0, s, traj, rows, n2, cap);
    return hipGetLastError();
}

hipError_t launch_terminal_geo(const EmgpuTGeoRun &A, hipStream_t s) {
    if (A.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_terminal_geo, dim3((unsigned)((A.n + 255) / 256)), dim3(256), 0, s, A);
    return hipGetLastError();
}
hipError_t launch_terminal_filter(const EmgpuTFilterRun &A, hipStream_t s, const char **name) {
    *name = "k_terminal_filter";
    if (A.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_terminal_filter, dim3((unsigned)((A.n + 255) / 256)), dim3(256), 0, s, A);
    return hipGetLastError();
}

} // namespace emgpu
