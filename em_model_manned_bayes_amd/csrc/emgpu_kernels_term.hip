// emgpu_kernels_term.hip -- k_terminal_propagate: PropagateTrajectory of the correlated terminal
// model (@CorTerminalModel/createEncounter.m:93-265, CreateStartDistribution :268-294,
// CheckTrajectoryConditions :296-329).  One lane = one (encounter, aircraft, direction) track:
// 4 consecutive lanes per encounter.  Every second: point-mass kinematics in f64, discretize the
// continuous state (discretize_bayes.m:14-22), one transition step of the trajectory DBN with the
// "stay" prior (dbn_sample.m with t_max = 2 and every initial variable preset; thresholds gathered
// from the per-model table by binary search, r up to 36), validity re-draws, dediscretize.
// em-core's local_smooth (createEncounter.m:88-89) is not applied: un-vendored dependency.
// Bound: dependent L2 gathers + f64 transcendental math (sincos, atan2); output 24 B per second.
#include <hip/hip_runtime.h>

#include "emgpu_device.h"
#include "emgpu_launch.h"

namespace emgpu {

__device__ __forceinline__ double t_wrapTo360(double lon) {
    const bool positive = lon > 0;
    lon = lon - floor(lon / 360.0) * 360.0;
    return (lon == 0 && positive) ? 360.0 : lon;
}
__device__ __forceinline__ double t_atan2d(double y, double x) { return atan2(y, x) * (180.0 / 3.14159265358979323846); }
__device__ __forceinline__ void t_sincosd(double deg, double &s, double &c) {
    const double r = fmod(deg, 360.0);
    if (r == 0) { s = 0; c = 1; return; }
    if (r == 90 || r == -270) { s = 1; c = 0; return; }
    if (r == 180 || r == -180) { s = 0; c = -1; return; }
    if (r == 270 || r == -90) { s = -1; c = 0; return; }
    const double rad = r * (3.14159265358979323846 / 180.0);
    s = sin(rad); c = cos(rad);
}
__device__ __forceinline__ double t_sign(double x) { return (double)((x > 0) - (x < 0)); }

__device__ __forceinline__ int t_discretize(double x, const double *__restrict__ cut, int n) { // 1-based bin
    if (x >= cut[n - 1]) return n + 1;
    int d = 1;
    for (int q = 0; q < n; q++) { if (x < cut[q]) break; d++; }
    return d;
}

// 1-based bin = 1 + #{t < r-1 : x' >= thr[t]} on a sorted threshold row (binary search)
__device__ __forceinline__ int t_draw(const uint32_t *__restrict__ thr, int rm1, uint32_t x) {
    const uint32_t xp = clamp32(x);
    int lo = 0, hi = rm1; // count in [lo, hi]
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (xp >= thr[mid]) lo = mid + 1; else hi = mid;
    }
    return lo + 1;
}

__global__ void __launch_bounds__(256) k_terminal_propagate(const EmgpuPlan P, const EmgpuTermRun A) {
#pragma clang fp contract(off)
    const int64_t L = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (L >= 4 * A.n) return;
    const int64_t e = L >> 2;
    const int role = (int)(L & 3), ac = role >> 1;
    const double dt_s = (role & 1) ? -1.0 : 1.0;
    const bool is_ownship = ac == 0;
    const uint64_t gidx = A.first_index + (uint64_t)e;
    Rng rng{(uint32_t)gidx, (uint32_t)(gidx >> 32), (uint32_t)role, (uint32_t)A.seed, (uint32_t)(A.seed >> 32)};
    const double *g = A.geo + e * 12 + ac * 6;
    const int intent = (int)g[5];
    const uint32_t *__restrict__ thr = A.thr_base[A.model_of[L]];
    const double minVel = A.dl[ac][0], maxVel = A.dl[ac][1], maxTurn = A.dl[ac][2], maxAlt = A.dl[ac][3], maxVert = A.dl[ac][4];
    // boundaries by variable id 1..6 (identical for every trajectory model, checked on the host)
    const double *bnd[7];
    int nb[7];
#pragma unroll
    for (int v = 1; v <= 6; v++) { bnd[v] = P.bnd + P.i_boff[v - 1]; nb[v] = P.i_nb[v - 1]; }
    int alt_last = 0, spd_first = 0, spd_last = 0;
    for (int q = 0; q < nb[5]; q++) if (bnd[5][q] <= maxAlt) alt_last = q + 1;
    for (int q = 0; q < nb[6]; q++) { if (!(bnd[6][q] >= minVel)) spd_first = q + 1; if (bnd[6][q] <= maxVel) spd_last = q + 1; }
    const double bounds_dist_hi = bnd[2][nb[2] - 1];

    double xy0 = g[0], xy1 = g[1], z_ft = g[2], heading_deg = g[4], t_s = 0, prev_z_rec = 0;
    double sh, chh;
    t_sincosd(heading_deg, sh, chh);
    double v0 = chh * g[3], v1 = sh * g[3];
    int ii = 1, rows = 0;
    const size_t nl = (size_t)4 * (size_t)A.n;
    bool go = true, failed = false;
    while (go) {
        if (rows >= A.cap) { failed = true; break; }
        const double speed = sqrt(v0 * v0 + v1 * v1);
        const double rec_x = xy0, rec_y = xy1;
        xy0 += v0 * dt_s / 6076.1154855643;
        xy1 += v1 * dt_s / 6076.1154855643;
        const double curr_hdg = t_wrapTo360(t_atan2d(v1, v0));
        double rec_z = z_ft;
        if (ii > 1) {
            const double alt_diff = z_ft - prev_z_rec;
            rec_z = prev_z_rec + t_sign(alt_diff) * fmin(maxVert, fabs(alt_diff));
        }
        prev_z_rec = rec_z;
        {
            float *o = A.out + (size_t)rows * nl + (size_t)L;
            const size_t fs = (size_t)A.cap * nl;
            o[0] = (float)t_s; o[fs] = (float)rec_x; o[2 * fs] = (float)rec_y; o[3 * fs] = (float)rec_z;
            o[4 * fs] = (float)curr_hdg; o[5 * fs] = (float)speed;
        }
        rows++;
        // CreateStartDistribution (0-based bins)
        int st[6];
        st[0] = intent - 1;
        st[1] = t_discretize(sqrt(xy0 * xy0 + xy1 * xy1), bnd[2] + 1, nb[2] - 2) - 1;
        st[2] = t_discretize(t_wrapTo360(t_atan2d(xy1, xy0)), bnd[3] + 1, nb[3] - 2) - 1;
        st[3] = t_discretize(heading_deg, bnd[4] + 1, nb[4] - 2) - 1;
        st[4] = t_discretize(z_ft, bnd[5] + 1, nb[5] - 2) - 1;
        st[5] = t_discretize(sqrt(v0 * v0 + v1 * v1), bnd[6] + 1, nb[6] - 2) - 1;
        // CPT column of each dynamic variable (asub2ind.m:13-14 as strides); topological position == variable id
        uint32_t col[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            uint32_t c = 0;
#pragma unroll
            for (int p = 0; p < 6; p++) c += P.d_stride_static[k][p] * (uint32_t)st[p];
#pragma unroll
            for (int q = 0; q < 3; q++) c += P.d_stride_cur[k][q] * (uint32_t)st[P.d_ivar[q]];
            col[k] = c;
        }
        bool resample = true;
        int att = 0;
        while (resample) {
            if (att >= A.max_resample) { failed = true; go = false; break; }
            rng.attempt = (uint32_t)role + 4u * (uint32_t)att;
            att++;
            int newbin[3]; // 1-based
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int rm1 = (int)P.d_r[k] - 1;
                const uint4 w4 = rng.block(11u /* TERM_TRANS */, P.d_tvar[k], (uint32_t)ii >> 2);
                newbin[k] = t_draw(thr + (P.d_off[k] - P.d_off[0]) + (size_t)col[k] * (uint32_t)rm1, rm1, word_of(w4, ii & 3));
            }
            resample = false;
#pragma unroll
            for (int e3 = 0; e3 < 3; e3++) { // events in ascending variable id
                if (resample) break;
                const int k = P.d_emit[e3];
                const int var = (int)P.d_ivar[k] + 1;
                const int d = k == 0 ? newbin[0] : (k == 1 ? newbin[1] : newbin[2]);
                if (d == st[var - 1] + 1) continue;
                // MATLAB: 1:[] and []:1:e are empty, so with no boundary at or below the limit no event of that variable is valid
                const bool ok = var == 4 || (var == 5 && alt_last >= 1 && d >= 1 && d <= alt_last) ||
                                (var == 6 && spd_first >= 1 && d >= spd_first && d <= spd_last);
                if (!ok) { resample = true; break; }
                const uint4 w4 = rng.block(12u /* TERM_DEDISC */, (uint32_t)(var - 1), (uint32_t)ii >> 2);
                const double val = dedisc_f64(P.bnd, P.i_boff[var - 1], d - 1, word_of(w4, ii & 3));
                if (var == 4) heading_deg = val;
                else if (var == 5) z_ft = val;
                else {
                    double s1 = val;
                    if (s1 < minVel) s1 = minVel;
                    if (s1 > maxVel) s1 = maxVel;
                    t_sincosd(heading_deg, sh, chh);
                    v0 = chh * s1; v1 = sh * s1;
                }
            }
        }
        if (failed) break;
        const double turn1 = round((heading_deg - curr_hdg) * 100.0) / 100.0;
        const double delta = fmin(fabs(turn1), maxTurn) * t_sign(turn1);
        t_sincosd(delta, sh, chh);
        const double vx = chh * v0 - sh * v1, vy = sh * v0 + chh * v1;
        v0 = vx; v1 = vy;
        t_s += dt_s; ii++;
        const double d_nm = sqrt(xy0 * xy0 + xy1 * xy1);
        const bool stop = (fabs(t_s) > A.tmax_s) || (d_nm > bounds_dist_hi) || ((intent == 1 || intent == 2) && d_nm <= 0.25) || (is_ownship && xy1 > 0.25);
        go = !stop;
    }
    if (failed) atomicOr(A.status, 1u);
    A.rows[L] = failed ? -rows - 1 : rows;
}

hipError_t launch_terminal_propagate(const EmgpuPlan &P, const EmgpuTermRun &A, hipStream_t s, const char **name) {
    *name = "k_terminal_propagate";
    if (A.n <= 0) return hipSuccess;
    const int64_t blocks = (4 * A.n + 255) / 256;
    hipLaunchKernelGGL(k_terminal_propagate, dim3((unsigned)blocks), dim3(256), 0, s, P, A);
    return hipGetLastError();
}

} // namespace emgpu
