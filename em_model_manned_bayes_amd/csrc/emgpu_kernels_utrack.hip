// emgpu_kernels_utrack.hip -- UncorEncounterModel.track (@UncorEncounterModel/UncorEncounterModel.m:419-471, coordSys 'NEU'):
// the sampled controls integrated into a 10 Hz track and the three rejection tests of :462-470, one lane per trajectory,
// reading the sampler's device output where it lies (init_val rows + the time-blocked dense trace).
//
// The reference integrates with em-core's run_dynamics_fast mex and differentiates the altitude with em-core's
// computeVerticalRate; em-core is not vendored by the reference ("dynamics unpinned", HISTORY.md section 10).  The point-mass
// model used instead is built from the quantities the reference hands to run_dynamics_fast (ic = [v n e h psi theta phi a],
// :443; dyn = [v_low v_high dh_min dh_max qmax rmax], :414; control rows [t hdot psidot a], :291-297), in f64 without
// contraction.  dt = 0.1 s, g = 32.2 ft/s^2 (the constant of :440); per step, with the control row active at t:
//     a      = 0 when it would drive v beyond [v_low, v_high]
//     hd     = min(max(hdot, dh_min), dh_max)
//     theta += clamp((asin(clamp(hd / v, -1, 1)) - theta) / dt, -qmax, qmax) * dt     pitch follows the commanded climb
//     phi   += clamp((atan(v * psidot / g) - phi) / dt, -rmax, rmax) * dt             bank follows the commanded turn
//     n += v cos(theta) cos(psi) dt;  e += v cos(theta) sin(psi) dt;  h += v sin(theta) dt
//     psi += g tan(phi) / v * dt;  v = min(max(v + a dt, v_low), v_high)
// results.time = 0 : 0.1 : T; computeVerticalRate(up(is_sec), time(is_sec)) := forward difference of the 1 Hz altitudes.
// Limits: @UncorEncounterModel/getDynamicLimits.m:1-130 evaluated on the host for every (G, A, altitude-layer range,
// speed-bin range) the expression can see, looked up here by the track's own minima and maxima.
// Bound: f64 issue (asin, atan and three sin/cos pairs on Horner sums per 0.1 s step); 64 B per recorded step.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "emgpu_device.h"
#include "emgpu_launch.h"
#include "emgpu_plan.h"

namespace emgpu {

// sin and cos of an angle of moderate size (|x| < 1e5; pitch, bank, heading in radians): k = rint(x * 2/pi), the remainder
// x - k * pi/2 taken in two pieces of pi/2 (the first has 33 significant bits: k * hi is exact), Horner sums on the remainder
// (sincos_small, emgpu_device.h), the quadrant from k.  About 45 instructions for both
// where the library's sin() + cos() take 180; results within 2 ulp of them (the tracks are compared at 1e-9).
template <bool SK = false>   // SK: coefficients as scalar operands (sk64): for the variant that evaluates these once per recorded row only
__device__ __forceinline__ void ut_sincos(double x, double &s, double &c) {
    if (!(fabs(x) < 1.0e5)) { s = sin(x); c = cos(x); return; }
    const double k = rint(x * 0.63661977236758134308);                 // 2/pi
    double r = fma(-k, 1.57079632673412561417, x);                     // pi/2, first 33 bits
    r = fma(-k, 6.07710050650619224932e-11, r);                        // pi/2 - the above
    double sr, cr;
    sincos_small<SK>(r, sr, cr);   // (scalar-operand coefficients in the literal step -- three waves instead of two -- were measured: +4 %, it is issue-bound)
    const int q = (int)((long long)k & 3ll);
    s = (q == 0) ? sr : ((q == 1) ? cr : ((q == 2) ? -sr : -cr));
    c = (q == 0) ? cr : ((q == 1) ? -sr : ((q == 2) ? -cr : sr));
}

// 1 / d to about one ulp: v_rcp_f64 and two Newton steps (5 instructions; the IEEE-exact division the compiler emits for `/`
// takes ~35).  The dynamics divide six times per 0.1 s step as written; here they share ONE reciprocal, 1 / (v cos(phi)).
__device__ __forceinline__ double ut_rcp(double d) {
    double x = __builtin_amdgcn_rcp(d);
    double e = fma(-d, x, 1.0);
    x = fma(x, e, x);
    e = fma(-d, x, 1.0);
    return fma(x, e, x);
}
// atan on fdlibm's breakpoints (s_atan.c: 7/16, 11/16, 19/16, 39/16) and kernel polynomial, the reduced argument's division through
// ut_rcp, selects instead of branches; within 2 ulp of the library's (checked against numpy in tests/test_host.py's restatement).
template <bool SK = false>
__device__ __forceinline__ double ut_atan(double x) {
    auto K = [](double v) { return SK ? sk64(v) : v; };
    const double ax = fabs(x);
    double num = ax, den = 1.0, c = 0.0;
    if (ax >= 0.4375) { num = fma(2.0, ax, -1.0); den = 2.0 + ax; c = 4.63647609000806093515e-01; }      // atan(0.5)
    if (ax >= 0.6875) { num = ax - 1.0; den = ax + 1.0; c = 7.85398163397448278999e-01; }                 // atan(1)
    if (ax >= 1.1875) { num = ax - 1.5; den = fma(1.5, ax, 1.0); c = 9.82793723247329054082e-01; }        // atan(1.5)
    if (ax >= 2.4375) { num = -1.0; den = ax; c = 1.57079632679489655800e+00; }                           // atan(inf)
    const double t = num * ut_rcp(den);
    const double z = t * t, w = z * z;
    double s1 = fma(w, K(1.62858201153657823623e-02), K(4.97687799461593236017e-02));
    s1 = fma(w, s1, K(6.66107313738753120669e-02));
    s1 = fma(w, s1, K(9.09088713343650656196e-02));
    s1 = fma(w, s1, K(1.42857142725034663711e-01));
    s1 = fma(w, s1, K(3.33333333333329318027e-01));
    s1 = z * s1;
    double s2 = fma(w, K(-3.65315727442169155270e-02), K(-5.83357013379057348645e-02));
    s2 = fma(w, s2, K(-7.69187620504482999495e-02));
    s2 = fma(w, s2, K(-1.11111104054623557880e-01));
    s2 = fma(w, s2, K(-1.99999999998764832476e-01));
    s2 = w * s2;
    const double r = c + (t - t * (s1 + s2));
    return x < 0 ? -r : r;
}
// asin for |x| < 0.5 (a climb angle below 30 degrees): fdlibm's rational kernel (e_asin.c), its division through ut_rcp
template <bool SK = false>
__device__ __forceinline__ double ut_asin_small(double x) {
    auto K = [](double v) { return SK ? sk64(v) : v; };
    const double t = x * x;
    double p = fma(t, K(3.47933107596021167570e-05), K(7.91534994289814532176e-04));
    p = fma(t, p, K(-4.00555345006794114027e-02));
    p = fma(t, p, K(2.01212532134862925881e-01));
    p = fma(t, p, K(-3.25565818622400915405e-01));
    p = fma(t, p, K(1.66666666666666657415e-01));
    p = t * p;
    double q = fma(t, K(7.70381505559019352791e-02), K(-6.88283971605453293030e-01));
    q = fma(t, q, K(2.02094576023350569471e+00));
    q = fma(t, q, K(-2.40339491173441421878e+00));
    q = fma(t, q, K(1.0));
    return fma(x, p * ut_rcp(q), x);
}

__device__ __forceinline__ int ut_discretize(double x, const double *cut, int n) { // discretize_bayes.m:14-22
    if (x >= cut[n - 1]) return n + 1;
    int d = n + 1;
    for (int i = n - 1; i >= 0; i--) d = (x < cut[i]) ? i + 1 : d;
    return d;
}

// FASTBANK: the bank-rate limit r_max can never bind (|atan(..) - phi| / dt <= 10 pi < r_max; the reference passes 1e6, :414), so the bank
// angle IS its command after every step, phi = atan(v psidot / g), and the step simplifies algebraically:
//   * g tan(phi) / v dt = psidot dt: the heading advances by the commanded rate -- no atan, no sin/cos of phi, no division by cos(phi);
//     within a second psidot is constant, so (cos psi, sin psi) are ROTATED by the second's (cos, sin)(psidot dt) instead of evaluated
//   * the pitch either reaches its command, theta = asin(sn) with sn = hd / v -- then sin(theta) = sn and cos(theta) = sqrt(1 - sn^2) -- or
//     turns towards it by q_max dt: a rotation of (cos theta, sin theta) by that fixed angle; which of the two, |asin(sn) - theta| <=
//     q_max dt, is read off sin(asin(sn) - theta) = sn cos(theta) - sqrt(1 - sn^2) sin(theta) together with the sign of
//     cos(asin(sn) - theta) (round 4: the sine alone is also small 180 degrees away): no asin
//   * phi and theta themselves are only needed in the recorded rows (atan / asin once per recorded step).
// One reciprocal, one square root and ~60 other f64 operations per 0.1 s step where the literal form takes three sin/cos pairs, an
// atan and an asin; every quantity differs from the literal form's by rounding (1e-16 per step, accumulated 1e-13: the tracks are
// compared with the oracle's literal form at 1e-9).  The literal form below stays for other r_max.
template <bool FASTBANK>
__global__ void __launch_bounds__(256) k_uncor_track(const EmgpuUTrackRun A) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n) return;
    const size_t ld = (size_t)A.ld;
    const int64_t slot = A.slot ? A.slot[i] : i;
    const double kPi180 = 3.14159265358979323846 / 180.0, dt = 0.1, g = 32.2;
    // :434-446
    const double h0 = (double)A.iL[i], v0 = (double)A.iV[i] * 1.68780972222222;
    const double dv0 = (double)A.iDV[i] * 1.68780972222222, dh0 = (double)A.iDH[i] / 60.0, dpsi0 = (double)A.iDPsi[i] * kPi180;
    double v = v0, n = 0.0, e = 0.0, h = h0, psi = 0.0, theta = asin(dh0 / v0), phi = atan(v0 * dpsi0 / 32.2);
    (void)dv0; // ic(8): the model takes its acceleration from the control rows, whose first row repeats it
    double up_min = h, up_max = h, v_min = v, v_max = v, vr_max = 0.0, h_sec = h;
    double *out = A.tracks ? A.tracks + (size_t)slot * (size_t)A.S * 8 : nullptr;
    if (out) { out[0] = 0.0; out[1] = n; out[2] = e; out[3] = h; out[4] = v; out[5] = phi; out[6] = theta; out[7] = psi; }
    const float4 *dv4 = reinterpret_cast<const float4 *>(A.dyn_val);
    int step = 0, since_rec = 0, rec = 0;
    if constexpr (FASTBANK) {
        double st, ct, cp = 1.0, sp = 0.0, cdl, sdl;   // (sin, cos) theta, (cos, sin) psi, (cos, sin)(q_max dt)
        ut_sincos<true>(theta, st, ct);
        sincos_small<false>(A.dyn[4] * dt, sdl, cdl);
        for (int c4 = 0; c4 * 4 < A.T; c4++) {
            const float4 qDH = dv4[((size_t)c4 * A.nd + A.sDH) * ld + (size_t)i];
            const float4 qDP = dv4[((size_t)c4 * A.nd + A.sDPsi) * ld + (size_t)i];
            const float4 qDV = dv4[((size_t)c4 * A.nd + A.sDV) * ld + (size_t)i];
            const float fDH[4] = {qDH.x, qDH.y, qDH.z, qDH.w}, fDP[4] = {qDP.x, qDP.y, qDP.z, qDP.w}, fDV[4] = {qDV.x, qDV.y, qDV.z, qDV.w};
#pragma unroll 1
            for (int w = 0; w < 4 && 4 * c4 + w < A.T; w++) {
                const double hdot = (double)fDH[w] / 60.0, psidot = (double)fDP[w] * kPi180, acmd = (double)fDV[w] * 1.68780972222222;
                const double hd = hdot < A.dyn[2] ? A.dyn[2] : (hdot > A.dyn[3] ? A.dyn[3] : hdot);
                const double dpsi = psidot * dt;   // the heading step of this second: g tan(atan(v psidot / g)) / v dt
                double cd, sd;
                ut_sincos<true>(dpsi, sd, cd);
#pragma unroll 1
                for (int s = 0; s < 10; s++) {
                    double a = acmd;
                    if ((v >= A.dyn[1] && a > 0) || (v <= A.dyn[0] && a < 0)) a = 0;
                    const double v_in = v;
                    double sn = hd * ut_rcp(v); sn = sn < -1 ? -1 : (sn > 1 ? 1 : sn);
                    const double cn = sqrt(fma(-sn, sn, 1.0));
                    const double dd = sn * ct - cn * st;                       // sin(asin(sn) - theta)
                    // |sin(x)| <= sin(q_max dt) says |x| <= q_max dt only while cos(x) > 0: a command clamped to +-90 degrees (|hdot| >= v:
                    // a slow rotorcraft) that changes sign is up to 180 degrees away from the pitch, where sin(x) is small again
                    if (fabs(dd) <= sdl && cn * ct + sn * st > 0.0) { st = sn; ct = cn; }   // the pitch reaches its command
                    else {                                                     // ... or turns towards it at q_max (theta and asin(sn) are
                        // both in [-90, 90] degrees, where sin is monotone: the command is above the pitch iff sn > sin(theta))
                        const double sg = sn < st ? -sdl : sdl, s2 = st * cdl + ct * sg, c2 = ct * cdl - st * sg;
                        st = s2; ct = c2;
                    }
                    n = n + v * ct * cp * dt;
                    e = e + v * ct * sp * dt;
                    h = h + v * st * dt;
                    psi = psi + dpsi;
                    { const double c2 = cp * cd - sp * sd, s2 = sp * cd + cp * sd; cp = c2; sp = s2; }
                    v = v + a * dt; v = v < A.dyn[0] ? A.dyn[0] : (v > A.dyn[1] ? A.dyn[1] : v);
                    step++;
                    if (++since_rec == A.stride) {   // every stride-th step is kept: the two angles of the row are evaluated here only
                        since_rec = 0; rec++;
                        if (out) {
                            double *o = out + (size_t)rec * 8;
                            theta = fabs(st) < 0.5 ? ut_asin_small<true>(st) : asin(st);
                            phi = ut_atan<true>(v_in * psidot * (1.0 / 32.2));
                            o[0] = (double)step / 10.0; o[1] = n; o[2] = e; o[3] = h; o[4] = v; o[5] = phi; o[6] = theta; o[7] = psi;
                        }
                    }
                    up_min = h < up_min ? h : up_min; up_max = h > up_max ? h : up_max;
                    v_min = v < v_min ? v : v_min; v_max = v > v_max ? v : v_max;
                }
                const double vr = fabs(h - h_sec);
                vr_max = vr > vr_max ? vr : vr_max;
                h_sec = h;
            }
        }
    } else
    for (int c4 = 0; c4 * 4 < A.T; c4++) {
        const float4 qDH = dv4[((size_t)c4 * A.nd + A.sDH) * ld + (size_t)i];
        const float4 qDP = dv4[((size_t)c4 * A.nd + A.sDPsi) * ld + (size_t)i];
        const float4 qDV = dv4[((size_t)c4 * A.nd + A.sDV) * ld + (size_t)i];
        const float fDH[4] = {qDH.x, qDH.y, qDH.z, qDH.w}, fDP[4] = {qDP.x, qDP.y, qDP.z, qDP.w}, fDV[4] = {qDV.x, qDV.y, qDV.z, qDV.w};
#pragma unroll 1
        for (int w = 0; w < 4 && 4 * c4 + w < A.T; w++) {
            // the control row active during this second (events2controls.m:16-27; units :295-297)
            const double hdot = (double)fDH[w] / 60.0, psidot = (double)fDP[w] * kPi180, acmd = (double)fDV[w] * 1.68780972222222;
            const double hd = hdot < A.dyn[2] ? A.dyn[2] : (hdot > A.dyn[3] ? A.dyn[3] : hdot);
#pragma unroll 1
            for (int s = 0; s < 10; s++) {
                double a = acmd;
                if ((v >= A.dyn[1] && a > 0) || (v <= A.dyn[0] && a < 0)) a = 0;
                // the step as the header states it, with its six divisions folded into one reciprocal, 1 / (v cos(phi)): 1 / v is its
                // product with cos(phi), tan(phi) / v its product with sin(phi); `/ dt` is `* 10`, `/ g` a multiplication by 1 / g.
                // Every rewrite moves a result by an ulp or two -- the tracks are compared with the oracle's at 1e-9 (same as before).
                double r = (ut_atan(v * psidot * (1.0 / 32.2)) - phi) * 10.0; r = r < -A.dyn[5] ? -A.dyn[5] : (r > A.dyn[5] ? A.dyn[5] : r);
                phi = phi + r * dt;
                double ct, st, cp, sp, cb, sb;
                ut_sincos(phi, sb, cb);
                const double rcv = ut_rcp(v * cb), inv_v = cb * rcv;
                double sn = hd * inv_v; sn = sn < -1 ? -1 : (sn > 1 ? 1 : sn);
                double as;   // chosen per lane by the lane's own value: a result never depends on which trajectories share its wave
                if (fabs(sn) < 0.5) as = ut_asin_small(sn);
                else as = asin(sn);                                                            // steeper than 30 degrees: the library's
                double q = (as - theta) * 10.0; q = q < -A.dyn[4] ? -A.dyn[4] : (q > A.dyn[4] ? A.dyn[4] : q);
                theta = theta + q * dt;
                ut_sincos(theta, st, ct);
                ut_sincos(psi, sp, cp);
                n = n + v * ct * cp * dt;
                e = e + v * ct * sp * dt;
                h = h + v * st * dt;
                psi = psi + g * (sb * rcv) * dt;
                v = v + a * dt; v = v < A.dyn[0] ? A.dyn[0] : (v > A.dyn[1] ? A.dyn[1] : v);
                step++;
                if (++since_rec == A.stride) {   // every stride-th step is kept
                    since_rec = 0; rec++;
                    if (out) {
                        double *o = out + (size_t)rec * 8;
                        o[0] = (double)step / 10.0; o[1] = n; o[2] = e; o[3] = h; o[4] = v; o[5] = phi; o[6] = theta; o[7] = psi;
                    }
                }
                up_min = h < up_min ? h : up_min; up_max = h > up_max ? h : up_max;
                v_min = v < v_min ? v : v_min; v_max = v > v_max ? v : v_max;
            }
            const double vr = fabs(h - h_sec);
            vr_max = vr > vr_max ? vr : vr_max;
            h_sec = h;
        }
    }
    // getDynamicLimits.m through the host-built table
    const double *L3 = A.lim;
    if (A.ordered) {
        const int dG = (int)A.iG[i], dA = (int)A.iA[i];
        int l0, l1, b0, b1;
        if (A.discL) l0 = l1 = (int)A.iL[i];
        else { l0 = ut_discretize(up_min, A.cutL, A.ncL); l1 = ut_discretize(up_max, A.cutL, A.ncL); }
        if (A.discV) b0 = b1 = (int)A.iV[i];
        else { b0 = ut_discretize(v_min * 0.592484, A.cutV, A.ncV); b1 = ut_discretize(v_max * 0.592484, A.cutV, A.ncV); }
        const size_t idx = (((((size_t)(dG - 1) * A.rA + (size_t)(dA - 1)) * A.rL + (size_t)(l0 - 1)) * A.rL + (size_t)(l1 - 1)) * A.rV + (size_t)(b0 - 1)) * A.rV + (size_t)(b1 - 1);
        L3 = A.lim + idx * 3;
    }
    const double minVel = L3[0], maxVel = L3[1], maxVR = L3[2];
    if (A.limits) { A.limits[(size_t)slot * 3] = minVel; A.limits[(size_t)slot * 3 + 1] = maxVel; A.limits[(size_t)slot * 3 + 2] = maxVR; }
    const bool viol = (up_min < A.min_alt || up_max > A.max_alt) || (v_min < minVel || v_max > maxVel) || (vr_max > maxVR);   // :462-470
    A.accepted[i] = viol ? 0 : 1;
    if (A.attempts) {
        if (!viol) A.attempts[slot] = A.attempt_no;
        else if (A.last_round) A.attempts[slot] = -1;
    }
}

// Rejected lanes of a round -> the next round's index lists, IN ORDER (lane i before lane i' > i: the launch order of a later round
// does not depend on which workgroup finished first): per-workgroup counts (ballot + population count), one exclusive scan over
// the workgroups, then every rejected lane writes at its workgroup's offset + its rank among the rejected lanes before it.
// scratch: count[0] = total, count[1 + b] = rejected lanes of workgroup b, then their exclusive scan in place.
__global__ void __launch_bounds__(256) k_count_rejected(int64_t n, const uint8_t *accepted, uint32_t *count) {
    __shared__ uint32_t s_w[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool rej = i < n && !accepted[i];
    const unsigned long long bal = __ballot(rej);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = (uint32_t)__popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) count[1 + blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ void __launch_bounds__(1024) k_scan_rejected(uint32_t nb, uint32_t *count) {
    __shared__ uint32_t s_part[1024];
    const uint32_t t = threadIdx.x, per = (nb + 1023u) / 1024u, lo = t * per, hi = min(lo + per, nb);
    uint32_t sum = 0;
    for (uint32_t b = lo; b < hi; b++) sum += count[1 + b];
    s_part[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {   // inclusive scan of the thread sums
        const uint32_t v = t >= d ? s_part[t - d] : 0u;
        __syncthreads();
        s_part[t] += v;
        __syncthreads();
    }
    uint32_t run = s_part[t] - sum;
    for (uint32_t b = lo; b < hi; b++) { const uint32_t c = count[1 + b]; count[1 + b] = run; run += c; }
    if (t == 1023u) count[0] = s_part[1023];
}
__global__ void __launch_bounds__(256) k_scatter_rejected(int64_t n, uint64_t first_index, const uint8_t *accepted, const uint64_t *gidx_in, const int64_t *slot_in,
                                                          uint64_t *gidx_out, int64_t *slot_out, const uint32_t *count) {
    __shared__ uint32_t s_w[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool rej = i < n && !accepted[i];
    const unsigned long long bal = __ballot(rej);
    const uint32_t w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) s_w[w] = (uint32_t)__popcll(bal);
    __syncthreads();
    if (!rej) return;
    uint32_t k = count[1 + blockIdx.x] + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
    for (uint32_t q = 0; q < w; q++) k += s_w[q];
    gidx_out[k] = gidx_in ? gidx_in[i] : first_index + (uint64_t)i;   // round 0 covers the contiguous range
    slot_out[k] = slot_in ? slot_in[i] : i;
}

hipError_t launch_uncor_track(const EmgpuUTrackRun &A, hipStream_t s, const char **name, int force_literal) {
    static const bool literal = getenv("EMGPU_DEBUG_UTRACK_LITERAL") != nullptr;   // tests: the literal step on the shipped limits
    const bool fast = A.dyn[5] >= 32.0 && !literal && !force_literal;   // |atan - phi| / dt <= 10 pi: a bank-rate limit above that never binds
    *name = fast ? "k_uncor_track<fastbank>" : "k_uncor_track";
    if (A.n <= 0) return hipSuccess;
    if (fast) hipLaunchKernelGGL(k_uncor_track<true>, dim3((unsigned)((A.n + 255) / 256)), dim3(256), 0, s, A);
    else hipLaunchKernelGGL(k_uncor_track<false>, dim3((unsigned)((A.n + 255) / 256)), dim3(256), 0, s, A);
    return hipGetLastError();
}

size_t compact_scratch_words(int64_t n) { return 2 + (size_t)((n + 255) / 256); }

hipError_t launch_compact_rejected(int64_t n, uint64_t first_index, const uint8_t *accepted, const uint64_t *gidx_in, const int64_t *slot_in,
                                   uint64_t *gidx_out, int64_t *slot_out, uint32_t *count /* compact_scratch_words(n) words; [0] receives the total */, hipStream_t s) {
    if (n <= 0) return hipMemsetAsync(count, 0, sizeof(uint32_t), s);
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_count_rejected, dim3(nb), dim3(256), 0, s, n, accepted, count);
    hipLaunchKernelGGL(k_scan_rejected, dim3(1), dim3(1024), 0, s, (uint32_t)nb, count);
    hipLaunchKernelGGL(k_scatter_rejected, dim3(nb), dim3(256), 0, s, n, first_index, accepted, gidx_in, slot_in, gidx_out, slot_out, count);
    return hipGetLastError();
}

} // namespace emgpu
