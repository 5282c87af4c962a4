// emgpu_kernels_term.hip -- k_terminal_propagate: PropagateTrajectory of the correlated terminal
// model (@CorTerminalModel/createEncounter.m:93-265, CreateStartDistribution :268-294,
// CheckTrajectoryConditions :296-329).  One lane = one (encounter, aircraft, direction) track:
// 4 consecutive lanes per encounter.  Every second: point-mass kinematics in f64, discretize the
// continuous state (discretize_bayes.m:14-22), one transition step of the trajectory DBN with the
// "stay" prior (dbn_sample.m with t_max = 2 and every initial variable preset; a column's thresholds gathered
// from the per-model table in one or two independent groups, r up to 36), validity re-draws, dediscretize.
// Structure: ONE loop over attempts (a re-drawing lane does not hold its wave back), boundaries in LDS,
// the recorded rows in a per-lane LDS ring that is flushed row by row with coalesced stores, the velocity's direction carried
// as an angle, sin/cos of the reduced angle as Horner sums, an instance for the terminal model's row shapes.
// em-core's local_smooth (createEncounter.m:88-89) is not applied: un-vendored dependency.
// Round 3: ONE Philox call per attempt serves the transition draws of all three dynamic variables (slot map: block = the step,
// word = the variable's row of the temporal map; the same for the dediscretize draws, made only when the lane has an event);
// the bearing bin comes from an f32 guess of the angle walked to the exact bin with f64 cross products against the cut directions
// (no atan2); distance is compared squared and the speed is carried (no square roots in the loop).
// Bound: vector instruction issue at two waves per SIMD (256 registers of f64 state) + dependent gathers; output 24 B per second.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "emgpu_device.h"
#include "emgpu_launch.h"

namespace emgpu {

// A constant held in a SCALAR register pair at its use: the f64 constants of the step loop are loop invariants, and left to the compiler
// they are materialised once and then occupy ~60 vector registers through the whole loop of a kernel that is short of them.
__device__ __forceinline__ double t_k(double c) { asm volatile("" : "+s"(c)); return c; }
// the carried direction angle brought into [0, 360): what wrapTo360(atan2d(v)) gives for the same direction
__device__ __forceinline__ double t_mod360(double lon) { return lon - floor(lon * t_k(1.0 / 360.0)) * t_k(360.0); }
__device__ __forceinline__ double t_atan2d(double y, double x) { return atan2(y, x) * (180.0 / 3.14159265358979323846); }
// wrapTo360(atan2d(y, x)): the angle is in [-180, 180], where lon - floor(lon / 360) * 360 is lon + 360 below zero and lon from
// zero up (the same roundings: floor is -1 or 0), and the "== 0 && positive" rule never fires
__device__ __forceinline__ double t_wrap_atan2d(double y, double x) {
    const double a = t_atan2d(y, x);
    return a < 0.0 ? a + 360.0 : a + 0.0;
}
// cosd / sind: MATLAB's reduction in degrees (n = round(x/90), x - 90 n in [-45, 45], quadrant mod(n, 4)) + the Horner sums of
// sincos_small (emgpu_device.h: same coefficients, same order of operations => the same bits), coefficients as scalar operands
__device__ __forceinline__ void t_sincosd(double deg, double &s, double &c) {
    const double n = round(deg * t_k(1.0 / 90.0));
    const double x = t_k(3.14159265358979323846 / 180.0) * (deg - n * 90.0);
    const int m = (int)((long long)n & 3ll);
    const double z = x * x;
    double ps = t_k(-1.0 / 121645100408832000.0);
    ps = fma(ps, z, t_k(1.0 / 355687428096000.0));
    ps = fma(ps, z, t_k(-1.0 / 1307674368000.0));
    ps = fma(ps, z, t_k(1.0 / 6227020800.0));
    ps = fma(ps, z, t_k(-1.0 / 39916800.0));
    ps = fma(ps, z, t_k(1.0 / 362880.0));
    ps = fma(ps, z, t_k(-1.0 / 5040.0));
    ps = fma(ps, z, t_k(1.0 / 120.0));
    ps = fma(ps, z, t_k(-1.0 / 6.0));
    const double sx = fma(x * z, ps, x);
    double pc = t_k(-1.0 / 6402373705728000.0);
    pc = fma(pc, z, t_k(1.0 / 20922789888000.0));
    pc = fma(pc, z, t_k(-1.0 / 87178291200.0));
    pc = fma(pc, z, t_k(1.0 / 479001600.0));
    pc = fma(pc, z, t_k(-1.0 / 3628800.0));
    pc = fma(pc, z, t_k(1.0 / 40320.0));
    pc = fma(pc, z, t_k(-1.0 / 720.0));
    pc = fma(pc, z, t_k(1.0 / 24.0));
    const double cx = (1.0 - 0.5 * z) + (z * z) * pc;
    s = (m == 0) ? sx : ((m == 1) ? cx : ((m == 2) ? -sx : -cx));
    c = (m == 0) ? cx : ((m == 1) ? -sx : ((m == 2) ? -cx : sx));
}
// dediscretize.m:33-39 on the two boundaries of 1-based bin d (LDS), f64 without contraction
__device__ __forceinline__ double t_dedisc(const double *__restrict__ bnd, int d, uint32_t x) {
#pragma clang fp contract(off)
    const double a = bnd[d - 1], b = bnd[d];
    const double dd = b - a;
    const double mm = dd * uniform32(x);
    return a + mm;
}
__device__ __forceinline__ double t_sign(double x) { return (double)((x > 0) - (x < 0)); }

// discretize_bayes.m:14-22 on boundaries held in LDS: 1-based bin = 1 + #{q : x >= cut[q]} for the sorted cut points
// cut = boundaries(2:end-1).  The answer is guessed from the grid's first point and mean spacing (exact for the 10-degree bearing /
// heading grids) and then walked to the true bin: any sorted grid gives the reference's answer, a uniform one in one or two LDS
// reads instead of a scan.
#ifndef EMGPU_TERM_RING
#define EMGPU_TERM_RING 6
#endif
constexpr int kRing = EMGPU_TERM_RING;       // rows a lane may run ahead of the slowest lane of its wave
constexpr int kBndStride = 68; // boundaries per variable in LDS (the host checks i_nb <= 66)
struct CutGrid { int off, n; double lo, inv_step; };
__device__ __forceinline__ int t_discretize(double x, const double *__restrict__ s_bnd, const CutGrid &gd) {
    const double *cut = s_bnd + gd.off + 1;
    double kd = (x - gd.lo) * gd.inv_step;               // candidate number of cut points <= x, minus one
    kd = kd < -1.0 ? -1.0 : (kd > (double)gd.n ? (double)gd.n : kd);
    int k = (int)kd + 1;
    k = k < 0 ? 0 : (k > gd.n ? gd.n : k);
    while (k > 0 && x < cut[k - 1]) k--;
    while (k < gd.n && x >= cut[k]) k++;
    return k + 1;
}

// The bearing bin of createEncounter.m:277-279, discretize_bayes(wrapTo360(atan2d(y, x)), cut) = 1 + #{q : angle >= cut[q]}, without the
// atan2: the half-plane is exact from the sign of y (atan2d < 0 <=> y < 0, its wrap adds 360), a 20-instruction f32 estimate of the
// angle inside it guesses the count, and the guess is walked to the exact count by testing the neighbouring cut directions with f64
// cross products: angle >= cut  <=>  cos(cut) y - sin(cut) x >= 0 for a cut within 180 degrees of the angle (the walk only ever
// looks at the guess's neighbours).  s_dir[q] = (cosd, sind)(cut[q]) with MATLAB's exact zeros at the multiples of 90.
__device__ __forceinline__ int t_bearing_bin(double x, double y, const double *__restrict__ cut, const double2 *__restrict__ s_dir, int n, double lo, double inv_step) {
    const float fx = fabsf((float)x), fy = fabsf((float)y);
    const float mx = fmaxf(fx, fy), mn = fminf(fx, fy);
    const float t = mx > 0.f ? mn * __builtin_amdgcn_rcpf(mx) : 0.f, t2 = t * t;
    float a = t * (0.99997726f + t2 * (-0.33262347f + t2 * (0.19354346f + t2 * (-0.11643287f + t2 * (0.05265332f - t2 * 0.01172120f)))));   // atan(t), 1e-5 rad
    a *= 57.29578f;
    a = fy > fx ? 90.f - a : a;                       // first-quadrant angle of (|x|, |y|)
    a = x < 0 ? 180.f - a : a;                        // upper half-plane angle of (x, |y|)
    a = y < 0 ? 360.f - a : a;                        // exact half: y < 0 <=> the angle is in (180, 360]
    double kd = ((double)a - lo) * inv_step;          // candidate number of cut points <= angle, minus one
    kd = kd < -1.0 ? -1.0 : (kd > (double)n ? (double)n : kd);
    int k = (int)kd + 1;
    k = k < 0 ? 0 : (k > n ? n : k);
    if (x == 0.0 && y == 0.0) {                       // atan2d(0, 0) = 0: only cut points at or below 0 count
        k = 0;
        while (k < n && 0.0 >= cut[k]) k++;
        return k + 1;
    }
    auto ge = [&](int q) { const double2 d = s_dir[q]; return d.x * y - d.y * x >= 0.0; };
    while (k > 0 && !ge(k - 1)) k--;
    while (k < n && ge(k)) k++;
    return k + 1;
}

// The same count on a grid of at most 8 cut points, held in LDS padded with +inf to 8: eight broadcast reads issued together and
// eight compares, no walk (distance, altitude and speed have 4 to 6 cut points)
__device__ __forceinline__ int t_discretize8(double x, const double *__restrict__ cut8) {
    int b = 1;
#pragma unroll
    for (int q = 0; q < 8; q++) b += (x >= cut8[q]) ? 1 : 0;
    return b;
}

// The transition draws of the three dynamic variables of one attempt (select_random.m:17-20 on precompiled thresholds):
// 1-based bin = 1 + #{t < rm1 : x' >= thr[t]} on a sorted threshold row.  Up to 8 thresholds are loaded together and counted;
// more (bearing / heading: 35) take the row's pivots first (every 6th threshold, kept as a row of their own), then the 6 of the
// group the draw falls in.  The first groups of all three
// variables are requested before any is used: one memory round trip for the lot, a second one only for the long rows.
// Indices past a row's end are read (the table carries 64 words of slack) and masked, not clamped.
typedef const uint32_t __attribute__((address_space(1))) *gptr_t;   // a global-memory pointer (the compiler cannot tell from a loaded one)
typedef uint32_t uint4u_t __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t uint2u_t __attribute__((ext_vector_type(2), aligned(4)));
struct Draw3 { int bin[3]; };
// c8: the compact form of a long row (EmgpuPlan::d_c8off; null: the variable has none): six distinct thresholds + the byte map decide
// the draw from ONE 32-byte gather; a row with more than six (flag byte) sends its lane down the pivot path
__device__ __forceinline__ Draw3 t_draw3(const gptr_t (&row)[3], const gptr_t (&piv)[3], const gptr_t (&c8)[3], const bool (&has_c8)[3], const int (&rm1)[3], const uint32_t (&x)[3]) {
    uint32_t first[3][8];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (has_c8[k]) {                                      // wave-uniform
            const uint4u_t a = *(const uint4u_t __attribute__((address_space(1))) *)c8[k], b = *(const uint4u_t __attribute__((address_space(1))) *)(c8[k] + 4);
            first[k][0] = a.x; first[k][1] = a.y; first[k][2] = a.z; first[k][3] = a.w;
            first[k][4] = b.x; first[k][5] = b.y; first[k][6] = b.z; first[k][7] = b.w;
        } else if (rm1[k] <= 8) {                             // wave-uniform
            // eight consecutive words from a 4-byte aligned address: two 16-byte loads
            const uint4u_t a = *(const uint4u_t __attribute__((address_space(1))) *)row[k];
            uint4u_t b = {0u, 0u, 0u, 0u};
            if (rm1[k] > 6) b = *(const uint4u_t __attribute__((address_space(1))) *)(row[k] + 4);   // (wave-uniform; masked below either way)
            else if (rm1[k] > 4) { const uint2u_t b2 = *(const uint2u_t __attribute__((address_space(1))) *)(row[k] + 4); b.x = b2.x; b.y = b2.y; }
            first[k][0] = a.x; first[k][1] = a.y; first[k][2] = a.z; first[k][3] = a.w;
            first[k][4] = b.x; first[k][5] = b.y; first[k][6] = b.z; first[k][7] = b.w;
        } else {
            // the row's pivots (EmgpuPlan::d_pivoff): every 6th threshold of its full groups but the last, "never" elsewhere
            const uint4u_t a = *(const uint4u_t __attribute__((address_space(1))) *)piv[k], b = *(const uint4u_t __attribute__((address_space(1))) *)(piv[k] + 4);
            first[k][0] = a.x; first[k][1] = a.y; first[k][2] = a.z; first[k][3] = a.w;
            first[k][4] = b.x; first[k][5] = b.y; first[k][6] = b.z; first[k][7] = b.w;
        }
    }
    Draw3 out;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const uint32_t xp = clamp32(x[k]);
        if (rm1[k] <= 8 && has_c8[k]) {                       // a 5- or 6-threshold row in its aligned compact form: six always fit
            int nf = 0;
#pragma unroll
            for (int q = 0; q < 6; q++) nf += (xp >= first[k][q]) ? 1 : 0;
            out.bin[k] = (int)((nf < 4 ? first[k][6] >> (8 * nf) : first[k][7] >> (8 * (nf - 4))) & 0xFFu);
        } else if (rm1[k] <= 8) {
            int b = 0;
#pragma unroll
            for (int q = 0; q < 8; q++) b += (q < rm1[k] && xp >= first[k][q]) ? 1 : 0;
            out.bin[k] = b + 1;
        } else if (rm1[k] <= 48) {
            bool pivots = true;
            if (has_c8[k]) {
                int nf = 0;
#pragma unroll
                for (int q = 0; q < 6; q++) nf += (xp >= first[k][q]) ? 1 : 0;
                out.bin[k] = (int)((nf < 4 ? first[k][6] >> (8 * nf) : first[k][7] >> (8 * (nf - 4))) & 0xFFu);
                pivots = (first[k][7] >> 24) != 0u;            // a row with more than six distinct thresholds (none in sparse tables)
                if (pivots) {
                    const uint4u_t a = *(const uint4u_t __attribute__((address_space(1))) *)piv[k], b = *(const uint4u_t __attribute__((address_space(1))) *)(piv[k] + 4);
                    first[k][0] = a.x; first[k][1] = a.y; first[k][2] = a.z; first[k][3] = a.w;
                    first[k][4] = b.x; first[k][5] = b.y; first[k][6] = b.z; first[k][7] = b.w;
                }
            }
            if (pivots) {
            int g = 0;                                        // groups of 6 thresholds; pivot = last threshold of a group
#pragma unroll
            for (int q = 0; q < 7; q++) g += (xp >= first[k][q]) ? 1 : 0;   // full groups entirely at or below x
            uint32_t t[6];
            {
                const gptr_t gp = row[k] + 6 * g;
                const uint4u_t a = *(const uint4u_t __attribute__((address_space(1))) *)gp;
                const uint2u_t b = *(const uint2u_t __attribute__((address_space(1))) *)(gp + 4);
                t[0] = a.x; t[1] = a.y; t[2] = a.z; t[3] = a.w; t[4] = b.x; t[5] = b.y;
            }
            int b = 6 * g;
#pragma unroll
            for (int q = 0; q < 6; q++) b += (6 * g + q < rm1[k] && xp >= t[q]) ? 1 : 0;
            out.bin[k] = b + 1;
            }
        } else {                                              // beyond 48 thresholds (none of the shipped shapes): plain search
            int lo = 0, hi = rm1[k];
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (xp >= row[k][mid]) lo = mid + 1; else hi = mid; }
            out.bin[k] = lo + 1;
        }
    }
    return out;
}

#ifndef EMGPU_TERM_WAVES
#define EMGPU_TERM_WAVES 4
#endif
// RM1_k: thresholds per row of dynamic variable k as a compile-time constant (0: read from the plan).  The instance built for the
// terminal model's shape (36 headings, 7 altitude and 5 speed bins) folds every "is this index inside the row" test; left to run
// time those wave-uniform masks are hoisted out of the loop, spill, and come back through v_readlane every iteration.
template <int RM1_0, int RM1_1, int RM1_2>
__global__ void __launch_bounds__(256, EMGPU_TERM_WAVES) k_terminal_propagate(const EmgpuPlan P, const EmgpuTermRun A) {
#pragma clang fp contract(off)
    // boundaries of variables 2..6 (distance, bearing, heading, altitude, speed), identical for every trajectory model (checked
    // on the host): the cut points of discretize_bayes are boundaries(2:end-1), dediscretize reads a bin's two edges
    __shared__ double s_bnd[5 * kBndStride];
    __shared__ CutGrid s_grid[5];
    __shared__ double s_cut8[5][8];   // the cut points of a grid with at most 8 of them, padded with +inf
    __shared__ float s_ring[4][kRing][5][64];   // per wave: the last kRing recorded rows of every lane (x y z heading speed; the time is the row number), field-major (conflict-free)
    __shared__ double2 s_dir[kBndStride];       // (cosd, sind) of the bearing variable's cut points
    __shared__ double s_cut8sq[8];              // squares of the distance variable's cut points (when it has at most 8)
    for (int q = threadIdx.x; q < (int)P.i_nb[2] - 2; q += 256) {
        double sd, cd;
        sincosd_small(P.bnd[P.i_boff[2] + 1 + q], sd, cd);
        s_dir[q] = make_double2(cd, sd);
    }
    if (threadIdx.x < 8) {
        const double c = ((int)threadIdx.x < (int)P.i_nb[1] - 2) ? P.bnd[P.i_boff[1] + 1 + threadIdx.x] : __builtin_inf();
        s_cut8sq[threadIdx.x] = c * c;   // (cut points of a distance are >= 0: d >= c <=> d^2 >= c^2)
    }
    for (int v = 2; v <= 6; v++) {
        const int nbv = P.i_nb[v - 1], n = nbv - 2;
        for (int q = threadIdx.x; q < nbv; q += 256) s_bnd[(v - 2) * kBndStride + q] = P.bnd[P.i_boff[v - 1] + q];
        if (threadIdx.x < 8) s_cut8[v - 2][threadIdx.x] = ((int)threadIdx.x < n) ? P.bnd[P.i_boff[v - 1] + 1 + threadIdx.x] : __builtin_inf();
        if (threadIdx.x == 0) {
            const double lo = P.bnd[P.i_boff[v - 1] + 1], hi = P.bnd[P.i_boff[v - 1] + nbv - 2];
            s_grid[v - 2] = CutGrid{(v - 2) * kBndStride, n, lo, (n > 1 && hi > lo) ? (double)(n - 1) / (hi - lo) : 0.0};
        }
    }
    __syncthreads();
    const int64_t L = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (L >= 4 * A.n) return;
    const int64_t e = L >> 2;
    const int role = (int)(L & 3), ac = role >> 1, lane = (int)(threadIdx.x & 63);
    const double dt_s = (role & 1) ? -1.0 : 1.0;
    const bool is_ownship = ac == 0;
    const uint64_t gidx = A.indices ? A.indices[e] : A.first_index + (uint64_t)e;
    Rng rng{(uint32_t)gidx, (uint32_t)(gidx >> 32), (uint32_t)role, (uint32_t)A.seed, (uint32_t)(A.seed >> 32)};
    const double *g = A.geo + e * 12 + ac * 6;
    const int intent = (int)g[5];
    const gptr_t thr = (gptr_t)A.thr_base[A.model_of[L]];
    // the aircraft's limits are picked from the kernel arguments where they are used (kept per lane they cost ten registers)
#define T_LIM(q) (ac ? A.dl[1][q] : A.dl[0][q])
    const double maxAlt = T_LIM(3);
    // the grids are wave-uniform: pinned in scalar registers (read back from LDS they would sit in 30 vector registers)
    auto sgrid = [&](int q) {
        const CutGrid gq = s_grid[q];
        auto u32 = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
        auto f64 = [&](double d) { const uint64_t b = (uint64_t)__double_as_longlong(d); return __longlong_as_double((long long)(((uint64_t)u32((uint32_t)(b >> 32)) << 32) | u32((uint32_t)b))); };
        return CutGrid{(int)u32((uint32_t)gq.off), (int)u32((uint32_t)gq.n), f64(gq.lo), f64(gq.inv_step)};
    };
    const CutGrid gDist = sgrid(0), gBear = sgrid(1), gHead = sgrid(2), gAlt = sgrid(3), gSpd = sgrid(4);
    int alt_last = 0, spd_first = 0, spd_last = 0;     // discreteValidAlt / discreteValidV as bin ranges (createEncounter.m:118-126)
    {
        const double *bA = s_bnd + 3 * kBndStride, *bS = s_bnd + 4 * kBndStride;
        for (int q = 0; q < (int)P.i_nb[4]; q++) if (bA[q] <= maxAlt) alt_last = q + 1;
        for (int q = 0; q < (int)P.i_nb[5]; q++) { if (!(bS[q] >= T_LIM(0))) spd_first = q + 1; if (bS[q] <= T_LIM(1)) spd_last = q + 1; }
    }
    const double bounds_dist_hi = s_bnd[P.i_nb[1] - 1];
    const int rm1[3] = {RM1_0 ? RM1_0 : (int)P.d_r[0] - 1, RM1_1 ? RM1_1 : (int)P.d_r[1] - 1, RM1_2 ? RM1_2 : (int)P.d_r[2] - 1};
    // which dynamic variable is heading / altitude / speed (the host checks that all three are there)
    // (the instance built for the shipped shape also knows the order: heading, altitude, speed with temporal-map rows 0, 1, 2 --
    // launch_terminal_propagate checks it; the selects below then fold)
    constexpr bool kShipped = RM1_0 == 35 && RM1_1 == 6 && RM1_2 == 4;
    const int kh = kShipped ? 0 : (P.d_ivar[0] == 3 ? 0 : (P.d_ivar[1] == 3 ? 1 : 2)), ka = kShipped ? 1 : (P.d_ivar[0] == 4 ? 0 : (P.d_ivar[1] == 4 ? 1 : 2)),
              ks = kShipped ? 2 : (P.d_ivar[0] == 5 ? 0 : (P.d_ivar[1] == 5 ? 1 : 2));
    const int drow[3] = {kShipped ? 0 : (int)P.d_row[0], kShipped ? 1 : (int)P.d_row[1], kShipped ? 2 : (int)P.d_row[2]};
    // the part of a column index that never changes along a track: the intent (variable 1)

    double xy0 = g[0], xy1 = g[1], z_ft = g[2], heading_deg = g[4], prev_z_rec = 0;
    double sh, chh;
    t_sincosd(heading_deg, sh, chh);
    double v0 = chh * g[3], v1 = sh * g[3];
    double speed = g[3];   // norm(v_ft_s), carried: the velocity is only ever speed * (cosd, sind) rotated (its norm to 1e-16)
    // The direction of the velocity, carried as an angle: the velocity is only ever set to speed * (cosd, sind)(heading) and rotated
    // by the step's turn, so atan2d(v) is this angle up to rounding (1e-14 degrees) -- the reference's per-step atan2d
    // (createEncounter.m:163) costs a hundred instructions here.  (v = 0 would give atan2d = 0: speeds are clamped to minVel > 0.)
    double vang = heading_deg;
    int ii = 1, rows = 0;
    const size_t nl = (size_t)4 * (size_t)A.n;
    bool done = false, failed = false;
    // ONE loop whose body is one attempt of the lane's current step: a lane whose draw produced an invalid event (createEncounter.m:
    // 218-262 re-draws the step) comes round again with att + 1 while its neighbours start their next step, instead of the whole
    // wave idling through an inner re-draw loop of the few.  The lanes of a wave therefore drift apart in their row numbers, and a
    // row written straight to the [6][cap][4n] output would be 64 scattered 4-byte stores; each lane keeps its last kRing rows in
    // LDS instead, and row r leaves for memory -- one 256-byte store per field for the wave -- once every running lane is past it.
    // A lane more than kRing rows ahead of the slowest waits (the slowest lane sets the wave's run time either way).
    // asub2ind.m:13-14 over the step's start state: the strides of a transition node's current-bin parents (heading, altitude, speed
    // in the model's order) folded into the strides of the same variables as initial-state parents, once (wave-uniform, scalar) --
    // per step the column is then six multiply-adds per node instead of nine plus three 6-way selects
    uint32_t cstr[3][6];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int p = 0; p < 6; p++) {
            uint32_t sv = P.d_stride_static[k][p];
#pragma unroll
            for (int q = 0; q < 3; q++) sv += ((int)P.d_ivar[q] == p) ? P.d_stride_cur[k][q] : 0u;
            cstr[k][p] = sv;
        }
    int att = 0, st[6] = {0, 0, 0, 0, 0, 0};
    uint32_t colk[3] = {0u, 0u, 0u};   // the step's CPT columns; the row addresses are formed at the draw (nine 64-bit pointers kept per lane cost 18 registers)
    // (compile_plan builds the compact rows for exactly the rows of 9 to 48 thresholds: a compile-time fact in the shipped-shape instance)
    const bool has_c8[3] = {rm1[0] > 8 && rm1[0] <= 48, rm1[1] > 8 && rm1[1] <= 48, rm1[2] > 8 && rm1[2] <= 48};   // wave-uniform
    double curr_hdg = 0;
    const bool dist8 = kShipped || gDist.n <= 8;   // wave-uniform: the distance grid is compared squared (the shipped-shape instance is only launched on such a grid)
    int flushed = 0; // wave-uniform: rows [0, flushed) of every lane are in memory
    while (__ballot(!done) != 0ull) {
        if (!done && (att != 0 || rows - flushed < kRing)) do {
            if (att == 0) {
                // ---- the step begins: record the state, move, discretize (createEncounter.m:156-200)
                if (rows >= A.cap) { failed = true; done = true; break; }
                float *rec = &s_ring[threadIdx.x >> 6][rows % kRing][0][lane];
                rec[0 * 64] = (float)xy0; rec[1 * 64] = (float)xy1; rec[4 * 64] = (float)speed;
                xy0 += (v0 * dt_s) * t_k(1.0 / 6076.1154855643);
                xy1 += (v1 * dt_s) * t_k(1.0 / 6076.1154855643);
                curr_hdg = (speed > 0.0) ? t_mod360(vang) : 0.0;
                double rec_z = z_ft;
                if (ii > 1) {
                    const double alt_diff = z_ft - prev_z_rec;
                    rec_z = prev_z_rec + t_sign(alt_diff) * fmin(T_LIM(4), fabs(alt_diff));
                }
                prev_z_rec = rec_z;
                rec[2 * 64] = (float)rec_z; rec[3 * 64] = (float)curr_hdg;
                rows++;
                // CreateStartDistribution (0-based bins), createEncounter.m:268-294
                const double d2_nm = xy0 * xy0 + xy1 * xy1;
                st[0] = intent - 1;
                st[1] = (dist8 ? t_discretize8(d2_nm, s_cut8sq) : t_discretize(sqrt(d2_nm), s_bnd, gDist)) - 1;     // wave-uniform choices
                st[2] = t_bearing_bin(xy0, xy1, s_bnd + gBear.off + 1, s_dir, gBear.n, gBear.lo, gBear.inv_step) - 1;
                st[3] = t_discretize(heading_deg, s_bnd, gHead) - 1;
                st[4] = ((kShipped || gAlt.n <= 8) ? t_discretize8(z_ft, s_cut8[3]) : t_discretize(z_ft, s_bnd, gAlt)) - 1;   // (7 and 5 bins: at most 8 cut points, checked at launch)
                st[5] = ((kShipped || gSpd.n <= 8) ? t_discretize8(speed, s_cut8[4]) : t_discretize(speed, s_bnd, gSpd)) - 1;    // norm(v_ft_s): the velocity has not changed since `speed`
                // CPT column of each dynamic variable (asub2ind.m:13-14 as strides); topological position == variable id
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    uint32_t c = 0u;
#pragma unroll
                    for (int p = 0; p < 6; p++)   // st[0]: the intent, which never changes
                        c = kShipped ? __umul24(cstr[k][p], (uint32_t)st[p]) + c   // one v_mad_u32_u24 (launch_terminal_propagate checks the strides fit 24 bits)
                                     : cstr[k][p] * (uint32_t)st[p] + c;
                    colk[k] = c;
                }
            }
            // ---- one attempt at the step's transition draw (attempt number in the Philox key)
            if (att >= A.max_resample) { failed = true; done = true; break; }
            rng.attempt = (uint32_t)role + 4u * (uint32_t)att;
            uint32_t xw[3];
            {   // block = the step, word = the variable's row of the temporal map: one Philox call for the three draws
                const uint4 tw = rng.block(11u /* TERM_TRANS */, 0u, (uint32_t)ii);
#pragma unroll
                for (int k = 0; k < 3; k++) xw[k] = word_of(tw, drow[k]);
            }
            gptr_t row[3], piv[3], c8[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                row[k] = thr + (P.d_off[k] - P.d_off[0]) + (size_t)colk[k] * (uint32_t)rm1[k];
                piv[k] = (rm1[k] > 8 && rm1[k] <= 48) ? thr + (P.d_pivoff[k] - P.d_off[0]) + (size_t)colk[k] * 8u : thr;   // wave-uniform
                c8[k] = has_c8[k] ? thr + (P.d_c8off[k] - P.d_off[0]) + (size_t)colk[k] * 8u : thr;
            }
            const Draw3 nb = t_draw3(row, piv, c8, has_c8, rm1, xw);
            // events in ascending variable id (createEncounter.m:218-262): heading (4), altitude (5), speed (6); an invalid altitude
            // or speed bin makes the step be drawn again -- the events applied before it stay applied, as in the reference's loop
            bool resample = false;
            const int dH = kh == 0 ? nb.bin[0] : (kh == 1 ? nb.bin[1] : nb.bin[2]), dA = ka == 0 ? nb.bin[0] : (ka == 1 ? nb.bin[1] : nb.bin[2]),
                      dS = ks == 0 ? nb.bin[0] : (ks == 1 ? nb.bin[1] : nb.bin[2]);
            uint4 dw = make_uint4(0u, 0u, 0u, 0u);   // the step's dediscretize draws (one Philox call, made by the lanes that have an event)
            if (dH != st[3] + 1 || dA != st[4] + 1 || dS != st[5] + 1) dw = rng.block(12u /* TERM_DEDISC */, 0u, (uint32_t)ii);
            if (dH != st[3] + 1) heading_deg = t_dedisc(s_bnd + 2 * kBndStride, dH, word_of(dw, kh == 0 ? drow[0] : (kh == 1 ? drow[1] : drow[2])));
            if (dA != st[4] + 1) {
                // MATLAB: 1:[] is empty, so with no boundary at or below the limit no altitude event is valid
                if (alt_last >= 1 && dA >= 1 && dA <= alt_last) z_ft = t_dedisc(s_bnd + 3 * kBndStride, dA, word_of(dw, ka == 0 ? drow[0] : (ka == 1 ? drow[1] : drow[2])));
                else resample = true;
            }
            if (!resample && dS != st[5] + 1) {
                if (spd_first >= 1 && dS >= spd_first && dS <= spd_last) {
                    double s1 = t_dedisc(s_bnd + 4 * kBndStride, dS, word_of(dw, ks == 0 ? drow[0] : (ks == 1 ? drow[1] : drow[2])));
                    const double minVel = T_LIM(0), maxVel = T_LIM(1);
                    if (s1 < minVel) s1 = minVel;
                    if (s1 > maxVel) s1 = maxVel;
                    t_sincosd(heading_deg, sh, chh);
                    v0 = chh * s1; v1 = sh * s1;
                    vang = heading_deg;
                    speed = s1;
                } else resample = true;
            }
            if (resample) { att++; break; }
            att = 0;
            // ---- the step ends: turn towards the new heading, advance the clock, stop conditions
            const double turn1 = round((heading_deg - curr_hdg) * t_k(100.0)) * t_k(0.01);
            const double delta = fmin(fabs(turn1), T_LIM(2)) * t_sign(turn1);
            if (delta != 0.0) {                                  // rotationmatrix(0) is the identity
                t_sincosd(delta, sh, chh);
                const double vx = chh * v0 - sh * v1, vy = sh * v0 + chh * v1;
                v0 = vx; v1 = vy;
                vang += delta;
            }
            ii++;
            const double d2_nm = xy0 * xy0 + xy1 * xy1;   // (the position has not moved since the step began: recomputed, not carried)
            done = ((double)(ii - 1) > A.tmax_s) || (d2_nm > bounds_dist_hi * bounds_dist_hi) || ((intent == 1 || intent == 2) && d2_nm <= 0.0625) || (is_ownship && xy1 > 0.25);
        } while (false);
        // ---- rows that every running lane has produced leave for memory
        for (;;) {
            if (__ballot(!done && rows <= flushed) != 0ull) break;   // a running lane has not produced this row yet
            if (__ballot(rows > flushed) == 0ull) break;             // nobody holds it
            if (rows > flushed) {
                const float *rec = &s_ring[threadIdx.x >> 6][flushed % kRing][0][lane];
                float *o = A.out + (size_t)flushed * nl + (size_t)L;
                const size_t fs = (size_t)A.cap * nl;
                o[0] = (float)(dt_s * (double)flushed);   // t_s = +-row: whole seconds, exact
#pragma unroll
                for (int f = 1; f < 6; f++) o[f * fs] = rec[(f - 1) * 64];
            }
            flushed++;
        }
    }
    if (failed && !A.quiet) atomicOr(A.status, 1u);
    A.rows[L] = failed ? -rows - 1 : rows;
}

hipError_t launch_terminal_propagate(const EmgpuPlan &P, const EmgpuTermRun &A, hipStream_t s, const char **name) {
    *name = "k_terminal_propagate";
    if (A.n <= 0) return hipSuccess;
    const int64_t blocks = (4 * A.n + 255) / 256;
    static const bool generic_only = getenv("EMGPU_DEBUG_TERM_GENERIC") != nullptr;   // tests: the run-time-shape instance on the shipped shape
    bool shipped_order = P.d_ivar[0] == 3 && P.d_ivar[1] == 4 && P.d_ivar[2] == 5 && P.d_row[0] == 0 && P.d_row[1] == 1 && P.d_row[2] == 2;
    if ((int)P.i_nb[1] - 2 > 8 || (int)P.i_nb[4] - 2 > 8 || (int)P.i_nb[5] - 2 > 8) shipped_order = false;   // distance, altitude and speed grids compared against eight padded cut points
    for (int k = 0; k < 3; k++)       // the folded strides of the instance's 24-bit multiply-adds
        for (int p = 0; p < 6; p++) {
            uint64_t sv = P.d_stride_static[k][p];
            for (int q = 0; q < 3; q++) sv += ((int)P.d_ivar[q] == p) ? P.d_stride_cur[k][q] : 0u;
            if (sv >= (1u << 24)) shipped_order = false;
        }
    if (!generic_only && shipped_order && P.d_r[0] == 36 && P.d_r[1] == 7 && P.d_r[2] == 5) {
        *name = "k_terminal_propagate<35,6,4>";
        hipLaunchKernelGGL((k_terminal_propagate<35, 6, 4>), dim3((unsigned)blocks), dim3(256), 0, s, P, A);
    } else {
        hipLaunchKernelGGL((k_terminal_propagate<0, 0, 0>), dim3((unsigned)blocks), dim3(256), 0, s, P, A);
    }
    return hipGetLastError();
}

} // namespace emgpu
