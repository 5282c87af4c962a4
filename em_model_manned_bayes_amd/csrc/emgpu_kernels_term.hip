// emgpu_kernels_term.hip -- k_terminal_propagate: PropagateTrajectory of the correlated terminal
// model (@CorTerminalModel/createEncounter.m:93-265, CreateStartDistribution :268-294,
// CheckTrajectoryConditions :296-329).  One lane = one (encounter, aircraft, direction) track:
// 4 consecutive lanes per encounter.  Every second: point-mass kinematics in f64, discretize the
// continuous state (discretize_bayes.m:14-22), one transition step of the trajectory DBN with the
// "stay" prior (dbn_sample.m with t_max = 2 and every initial variable preset; a column's thresholds gathered
// from the per-model table in one or two independent groups, r up to 36), validity re-draws, dediscretize.
// Cut points live in LDS (a guessed bin walked to the exact one), the Philox blocks of the first attempt serve four steps.
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
// cosd / sind with MATLAB's reduction in degrees: n = round(x/90), x - 90 n in [-45, 45], quadrant m = mod(n, 4)
__device__ __forceinline__ void t_sincosd(double deg, double &s, double &c) {
    const double n = round(deg / 90.0);
    const double x = (3.14159265358979323846 / 180.0) * (deg - n * 90.0);
    const int m = (int)((long long)n & 3ll);
    const double sx = sin(x), cx = cos(x);
    s = (m == 0) ? sx : ((m == 1) ? cx : ((m == 2) ? -sx : -cx));
    c = (m == 0) ? cx : ((m == 1) ? -sx : ((m == 2) ? -cx : sx));
}
__device__ __forceinline__ double t_sign(double x) { return (double)((x > 0) - (x < 0)); }

// discretize_bayes.m:14-22 on cut points held in LDS: 1-based bin = 1 + #{q : x >= cut[q]} for sorted cuts.  The answer is
// guessed from the grid's first point and mean spacing (exact for the 10-degree bearing / heading grids) and then walked to
// the true bin: any sorted grid gives the reference's answer, a uniform one in one or two LDS reads instead of a scan.
struct CutGrid { int off, n; double lo, inv_step; };
__device__ __forceinline__ int t_discretize(double x, const double *__restrict__ s_cut, const CutGrid &gd) {
    const double *cut = s_cut + gd.off;
    double kd = (x - gd.lo) * gd.inv_step;               // candidate number of cut points <= x, minus one
    kd = kd < -1.0 ? -1.0 : (kd > (double)gd.n ? (double)gd.n : kd);
    int k = (int)kd + 1;
    k = k < 0 ? 0 : (k > gd.n ? gd.n : k);
    while (k > 0 && x < cut[k - 1]) k--;
    while (k < gd.n && x >= cut[k]) k++;
    return k + 1;
}

// 1-based bin = 1 + #{t < rm1 : x' >= thr[t]} on a sorted threshold row.  Up to 8 thresholds: loaded together and counted
// (one memory round trip).  More (bearing / heading: 35): every 6th first, then the 6 of the group it falls in (two round
// trips instead of the six of a binary search; the kernel is bound by dependent gathers, not by compares).
// Kept out of line: inlined three times (one per variable) it takes the kernel from 231 to 304 registers and from 48 to 86 ms per
// million encounters.
__device__ __attribute__((noinline)) int t_draw(const uint32_t *__restrict__ thr, int rm1, uint32_t x) {
    const uint32_t xp = clamp32(x);
    if (rm1 <= 8) {
        uint32_t t[8];
#pragma unroll
        for (int q = 0; q < 8; q++) t[q] = thr[q < rm1 ? q : rm1 - 1];
        int b = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) b += (q < rm1 && xp >= t[q]) ? 1 : 0;
        return b + 1;
    }
    const int ngrp = (rm1 + 5) / 6;                         // groups of 6 thresholds; pivot = last threshold of a group
    uint32_t pv[7];
#pragma unroll
    for (int q = 0; q < 7; q++) { const int idx = 6 * q + 5; pv[q] = thr[idx < rm1 ? idx : rm1 - 1]; }
    int g = 0;
#pragma unroll
    for (int q = 0; q < 7; q++) g += (q < ngrp - 1 && xp >= pv[q]) ? 1 : 0;   // full groups entirely at or below x
    if (ngrp > 8) {                                          // beyond 48 thresholds (none of the shipped shapes): plain search
        int lo = 0, hi = rm1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (xp >= thr[mid]) lo = mid + 1; else hi = mid; }
        return lo + 1;
    }
    uint32_t t[6];
#pragma unroll
    for (int q = 0; q < 6; q++) { const int idx = 6 * g + q; t[q] = thr[idx < rm1 ? idx : rm1 - 1]; }
    int b = 6 * g;
#pragma unroll
    for (int q = 0; q < 6; q++) b += (6 * g + q < rm1 && xp >= t[q]) ? 1 : 0;
    return b + 1;
}

#ifndef EMGPU_TERM_WAVES
#define EMGPU_TERM_WAVES 1
#endif
__global__ void __launch_bounds__(256, EMGPU_TERM_WAVES) k_terminal_propagate(const EmgpuPlan P, const EmgpuTermRun A) {
#pragma clang fp contract(off)
    // cut points of variables 2..6 (distance, bearing, heading, altitude, speed): boundaries(2:end-1), identical for every
    // trajectory model (checked on the host)
    __shared__ double s_cut[5 * 64];
    __shared__ CutGrid s_grid[5];
    for (int v = 2; v <= 6; v++) {
        const int nbv = P.i_nb[v - 1], n = nbv - 2;
        for (int q = threadIdx.x; q < n; q += 256) s_cut[(v - 2) * 64 + q] = P.bnd[P.i_boff[v - 1] + 1 + q];
        if (threadIdx.x == 0) {
            const double lo = P.bnd[P.i_boff[v - 1] + 1], hi = P.bnd[P.i_boff[v - 1] + nbv - 2];
            s_grid[v - 2] = CutGrid{(v - 2) * 64, n, lo, (n > 1 && hi > lo) ? (double)(n - 1) / (hi - lo) : 0.0};
        }
    }
    __syncthreads();
    const int64_t L = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (L >= 4 * A.n) return;
    const int64_t e = L >> 2;
    const int role = (int)(L & 3), ac = role >> 1;
    const double dt_s = (role & 1) ? -1.0 : 1.0;
    const bool is_ownship = ac == 0;
    const uint64_t gidx = A.indices ? A.indices[e] : A.first_index + (uint64_t)e;
    Rng rng{(uint32_t)gidx, (uint32_t)(gidx >> 32), (uint32_t)role, (uint32_t)A.seed, (uint32_t)(A.seed >> 32)};
    const double *g = A.geo + e * 12 + ac * 6;
    const int intent = (int)g[5];
    const uint32_t *__restrict__ thr = A.thr_base[A.model_of[L]];
    const double minVel = A.dl[ac][0], maxVel = A.dl[ac][1], maxTurn = A.dl[ac][2], maxAlt = A.dl[ac][3], maxVert = A.dl[ac][4];
    const CutGrid gDist = s_grid[0], gBear = s_grid[1], gHead = s_grid[2], gAlt = s_grid[3], gSpd = s_grid[4];
    int alt_last = 0, spd_first = 0, spd_last = 0;     // discreteValidAlt / discreteValidV as bin ranges (createEncounter.m:118-126)
    {
        const double *bA = P.bnd + P.i_boff[4], *bS = P.bnd + P.i_boff[5];
        for (int q = 0; q < (int)P.i_nb[4]; q++) if (bA[q] <= maxAlt) alt_last = q + 1;
        for (int q = 0; q < (int)P.i_nb[5]; q++) { if (!(bS[q] >= minVel)) spd_first = q + 1; if (bS[q] <= maxVel) spd_last = q + 1; }
    }
    const double bounds_dist_hi = P.bnd[P.i_boff[1] + P.i_nb[1] - 1];

    double xy0 = g[0], xy1 = g[1], z_ft = g[2], heading_deg = g[4], t_s = 0, prev_z_rec = 0;
    double sh, chh;
    t_sincosd(heading_deg, sh, chh);
    double v0 = chh * g[3], v1 = sh * g[3];
    int ii = 1, rows = 0;
    const size_t nl = (size_t)4 * (size_t)A.n;
    bool go = true, failed = false;
    // the TERM_TRANS / TERM_DEDISC blocks of the first attempt serve four consecutive steps (word ii & 3): kept across steps
    uint4 wt[3] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    int wt_blk = -1;
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
        // CreateStartDistribution (0-based bins), createEncounter.m:268-294
        const double d_nm = sqrt(xy0 * xy0 + xy1 * xy1);
        int st[6];
        st[0] = intent - 1;
        st[1] = t_discretize(d_nm, s_cut, gDist) - 1;
        st[2] = t_discretize(t_wrapTo360(t_atan2d(xy1, xy0)), s_cut, gBear) - 1;
        st[3] = t_discretize(heading_deg, s_cut, gHead) - 1;
        st[4] = t_discretize(z_ft, s_cut, gAlt) - 1;
        st[5] = t_discretize(speed, s_cut, gSpd) - 1;          // norm(v_ft_s): the velocity has not changed since `speed`
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
        if ((ii >> 2) != wt_blk) {
            rng.attempt = (uint32_t)role;
#pragma unroll
            for (int k = 0; k < 3; k++) wt[k] = rng.block(11u /* TERM_TRANS */, P.d_tvar[k], (uint32_t)ii >> 2);
            wt_blk = ii >> 2;
        }
        bool resample = true;
        int att = 0;
        while (resample) {
            if (att >= A.max_resample) { failed = true; go = false; break; }
            rng.attempt = (uint32_t)role + 4u * (uint32_t)att;
            int newbin[3]; // 1-based
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int rm1 = (int)P.d_r[k] - 1;
                const uint4 w4 = att == 0 ? wt[k] : rng.block(11u /* TERM_TRANS */, P.d_tvar[k], (uint32_t)ii >> 2);
                newbin[k] = t_draw(thr + (P.d_off[k] - P.d_off[0]) + (size_t)col[k] * (uint32_t)rm1, rm1, word_of(w4, ii & 3));
            }
            att++;
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
        if (delta != 0.0) {                                  // rotationmatrix(0) is the identity
            t_sincosd(delta, sh, chh);
            const double vx = chh * v0 - sh * v1, vy = sh * v0 + chh * v1;
            v0 = vx; v1 = vy;
        }
        t_s += dt_s; ii++;
        const bool stop = (fabs(t_s) > A.tmax_s) || (d_nm > bounds_dist_hi) || ((intent == 1 || intent == 2) && d_nm <= 0.25) || (is_ownship && xy1 > 0.25);
        go = !stop;
    }
    if (failed && !A.quiet) atomicOr(A.status, 1u);
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
