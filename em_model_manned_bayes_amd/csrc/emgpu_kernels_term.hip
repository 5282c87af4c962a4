// emgpu_kernels_term.hip -- k_terminal_propagate: PropagateTrajectory of the correlated terminal
// model (@CorTerminalModel/createEncounter.m:93-265, CreateStartDistribution :268-294,
// CheckTrajectoryConditions :296-329).  One lane = one (encounter, aircraft, direction) track:
// 4 consecutive lanes per encounter.  Every second: point-mass kinematics in f64, discretize the
// continuous state (discretize_bayes.m:14-22), one transition step of the trajectory DBN with the
// "stay" prior (dbn_sample.m with t_max = 2 and every initial variable preset; a column's thresholds gathered
// from the per-model table in one or two independent groups, r up to 36), validity re-draws, dediscretize.
// Structure (HISTORY.md section 7): ONE loop over attempts (a re-drawing lane does not hold its wave back), boundaries in LDS,
// persistent workgroups whose lanes take the next track from a queue when theirs ends (round 4), the recorded rows staged per lane in
// LDS and written by the wave as contiguous pieces of a TRACK-MAJOR output (round 4), the velocity's direction carried
// as an angle, sin/cos of the reduced angle as Horner sums, an instance for the terminal model's row shapes.
// em-core's local_smooth (createEncounter.m:88-89) is a separate, flagged pass (k_terminal_smooth): un-vendored dependency.
// Round 3: ONE Philox call per attempt serves the transition draws of all three dynamic variables (slot map: block = the step,
// word = the variable's row of the temporal map; the same for the dediscretize draws, made only when the lane has an event);
// the bearing bin comes from an f32 guess of the angle walked to the exact bin with f64 cross products against the cut directions
// (no atan2); distance is compared squared and the speed is carried (no square roots in the loop).
// Bound: vector instruction issue + dependent gathers; output 20 B per track-second (x y z heading speed; t_s is the row number).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <type_traits>

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

// The same bin from the bin of the second before.  The bearing of an aircraft moves by a fraction of a degree per second, so last step's
// count k0 is almost always still right: two cross products confirm it.  The tests compare angles cyclically, so the walk stays inside the
// angle's exact half -- [0, 180) holds the counts 0..kA (kA = cut points below 180), [180, 360) the counts kB..n (kB = cut points at or
// below 180) -- where every cut point is within 180 degrees of the angle; when the half has changed the walk starts from the end of the
// new half the position is next to (x >= 0: the 0 / 360 end).  atan2d(0, x < 0) = 180 belongs to the upper half-interval.
__device__ __forceinline__ int t_bearing_walk(double x, double y, const double *__restrict__ cut, const double2 *__restrict__ s_dir, int n, int kA, int kB, int k0) {
    if (x == 0.0 && y == 0.0) {                       // atan2d(0, 0) = 0: only cut points at or below 0 count
        int k = 0;
        while (k < n && 0.0 >= cut[k]) k++;
        return k + 1;
    }
    const bool low = y > 0.0 || (y == 0.0 && x > 0.0);   // the angle is in [0, 180)
    const int lo = low ? 0 : kB, hi = low ? kA : n;
    int k = k0;
    if (k < lo || k > hi) k = (x >= 0.0) == low ? lo : hi;
    auto ge = [&](int q) { const double2 d = s_dir[q]; return d.x * y - d.y * x >= 0.0; };
    while (k > lo && !ge(k - 1)) k--;
    while (k < hi && ge(k)) k++;
    return k + 1;
}

// dediscretize drew v from bin d (1-based) of a variable with nb boundaries: v lies in [bnd[d-1], bnd[d]), which discretize_bayes maps back to
// d -- unless a rounding put it on the upper edge (the first bin has no lower cut point, the last no upper one)
__device__ __forceinline__ bool t_in_bin(const double *__restrict__ bnd, int nb, int d, double v) {
    return (d == 1 || v >= bnd[d - 1]) && (d == nb - 1 || v < bnd[d]);
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
typedef float float4u_t __attribute__((ext_vector_type(4), aligned(4)));   // a 20-byte row grid: pieces start on 4-byte boundaries
typedef float float2u_t __attribute__((ext_vector_type(2), aligned(4)));
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
#ifndef EMGPU_TERM_ROWS
#define EMGPU_TERM_ROWS 6
#endif
constexpr int kRows = EMGPU_TERM_ROWS;        // rows a lane collects in LDS before the wave writes them out as ONE contiguous piece of its track
constexpr int kLaneStride = 5 * kRows + 1;    // dwords of a lane's staging area (odd: the lanes' rows fall on different banks)
// the flush copies a piece with one dword per lane of a half-wave, and packs its source offset (dwords into the wave's staging area) into
// 11 bits and its length (dwords) into 5 bits of the address's high dword (bits 48-63: device addresses stay below 2^48)
static_assert(64 * kLaneStride <= 2048 && 5 * kRows < 32, "EMGPU_TERM_ROWS: a piece no longer fits a half-wave / the flush descriptor (11 + 5 bits)");
#ifndef EMGPU_TERM_REFILL
#define EMGPU_TERM_REFILL 8
#endif
constexpr int kRefillMin = EMGPU_TERM_REFILL; // idle lanes a wave collects before it spends the (divergent) track set-up on them
constexpr uint32_t kChunk = 128;              // tracks a wave takes from the launch's queue at a time (>= 64)
// (Round 5 built a wave-level EVENT QUEUE into this loop -- lanes with an event park until N have collected -- measured it at three
// thresholds and dropped it: profiles/r05_terminal_event_queue.txt.  That variant, the ablation builds of HISTORY.md section 11.1 and the
// path counters of tools/term_counters.py live in tools/patches/term_lab.patch, which tools/build_term_variant.sh applies to a copy.)

// RM1_k: thresholds per row of dynamic variable k as a compile-time constant (0: read from the plan).  The instance built for the
// terminal model's shape (36 headings, 7 altitude and 5 speed bins) folds every "is this index inside the row" test; left to run
// time those wave-uniform masks are hoisted out of the loop, spill, and come back through v_readlane every iteration.
//
// Round 4: LANE REFILL + TRACK-MAJOR OUTPUT.  Tracks end anywhere between 2 and tmax_s + 2 rows, so a wave that owns 64 fixed tracks
// idles a fifth of its lane-iterations behind its longest one.  Now the workgroups are persistent: a wave takes tracks from a queue (one
// atomic per 128 tracks), and a lane whose track has ended starts the next one (set-up batched: kRefillMin idle lanes at a time).  The
// lanes of a wave are then at unrelated rows of unrelated tracks, which rules out the row-synchronous planes of rounds 1-3
// ([6][cap][4n]: a row of 64 neighbours = one store); the output is TRACK-MAJOR instead -- the joined, time-ordered track of an aircraft
// (createEncounter.m:74-84: [fwd, bck(2:end)] sorted by t_s) is one contiguous [W][5] block, row C + t for second t (forward lanes write
// upwards from C, backward lanes downwards; t_s itself is the row number and is not stored).  A lane stages kRows rows in LDS; the wave
// then writes them as one contiguous 20 kRows-byte piece (lanes = consecutive dwords), so stores stay coalesced although tracks are not.
template <int RM1_0, int RM1_1, int RM1_2>
__global__ void __launch_bounds__(256, EMGPU_TERM_WAVES) k_terminal_propagate(const EmgpuPlan P, const EmgpuTermRun A) {
#pragma clang fp contract(off)
    // boundaries of variables 2..6 (distance, bearing, heading, altitude, speed), identical for every trajectory model (checked
    // on the host): the cut points of discretize_bayes are boundaries(2:end-1), dediscretize reads a bin's two edges
    __shared__ double s_bnd[5 * kBndStride];
    __shared__ CutGrid s_grid[5];
    __shared__ double s_cut8[5][8];   // the cut points of a grid with at most 8 of them, padded with +inf
    __shared__ float s_stage[4 * 64 * kLaneStride + 32];   // per wave and lane: up to kRows recorded rows (x y z heading speed) waiting to be written (+ slack: the flush reads whole 16-byte parts)
    __shared__ uint2 s_desc[4][64];                        // per wave: the pieces of one flush, in lane order
    __shared__ double2 s_dir[kBndStride];       // (cosd, sind) of the bearing variable's cut points
    __shared__ double s_cut8sq[8];              // squares of the distance variable's cut points (when it has at most 8)
    // Wave-uniform values the loop needs now and then.  Held in scalar registers across the loop they do not fit (the kernel had 97 scalar
    // spills, each reload a v_readlane in the loop: 95 per iteration); read from here where they are used (volatile: not hoisted back).
    struct Uniforms {
        double dl[2][5];                 // per aircraft: minVel maxVel maxTurnRate maxAltitude maxVertRate
        double tmax_s, dist_hi2;         // CheckTrajectoryConditions (createEncounter.m:296-329)
        double grid[2][2];               // bearing, heading: first cut point, 1 / spacing (the guess of the discretize walk)
        int grid_n[2];
        int alt_last[2], spd_first[2], spd_last[2];   // discreteValidAlt / discreteValidV as bin ranges (createEncounter.m:118-126), per aircraft
        int cap, max_resample, quiet, pad;
        int bear_kA, bear_kB;            // bearing cut points below 180 degrees / at or below 180
        const double *geo; const int32_t *model_of; const uint32_t *const *thr_base; const uint64_t *indices; uint64_t first_index;
        int32_t *rows; uint32_t *status; uint32_t *queue;
    };
    __shared__ Uniforms s_u;
#define U(field) (*(const volatile std::remove_reference_t<decltype(s_u.field)> __attribute__((address_space(3))) *)&s_u.field)
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
    if (threadIdx.x < 10) s_u.dl[threadIdx.x / 5][threadIdx.x % 5] = A.dl[threadIdx.x / 5][threadIdx.x % 5];
    if (threadIdx.x == 64) {
        s_u.tmax_s = A.tmax_s; s_u.cap = A.cap; s_u.max_resample = A.max_resample; s_u.quiet = A.quiet;
        s_u.geo = A.geo; s_u.model_of = A.model_of; s_u.thr_base = A.thr_base; s_u.indices = A.indices; s_u.first_index = A.first_index;
        s_u.rows = A.rows; s_u.status = A.status; s_u.queue = A.queue;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        // discreteValidAlt / discreteValidV as bin ranges (createEncounter.m:118-126)
        const int a = threadIdx.x;
        const double *bA = s_bnd + 3 * kBndStride, *bS = s_bnd + 4 * kBndStride;
        int al = 0, sf = 0, sl = 0;
        for (int q = 0; q < (int)P.i_nb[4]; q++) if (bA[q] <= A.dl[a][3]) al = q + 1;
        for (int q = 0; q < (int)P.i_nb[5]; q++) { if (!(bS[q] >= A.dl[a][0])) sf = q + 1; if (bS[q] <= A.dl[a][1]) sl = q + 1; }
        s_u.alt_last[a] = al; s_u.spd_first[a] = sf; s_u.spd_last[a] = sl;
        const CutGrid gq = s_grid[1 + a];
        s_u.grid[a][0] = gq.lo; s_u.grid[a][1] = gq.inv_step; s_u.grid_n[a] = gq.n;
    }
    if (threadIdx.x == 2) {
        const double hi = s_bnd[P.i_nb[1] - 1];
        s_u.dist_hi2 = hi * hi;
        int ka = 0, kb = 0;
        for (int q = 0; q < (int)P.i_nb[2] - 2; q++) { const double c = s_bnd[kBndStride + 1 + q]; ka += c < 180.0; kb += c <= 180.0; }
        s_u.bear_kA = ka; s_u.bear_kB = kb;
    }
    __syncthreads();
    const int lane = (int)(threadIdx.x & 63);
    float *const stage = s_stage + (threadIdx.x >> 6) * (64 * kLaneStride);
    uint2 *const desc = s_desc[threadIdx.x >> 6];
    float *const mine = stage + lane * kLaneStride;
    const uint32_t total = (uint32_t)(4 * A.n);
    // the aircraft's limits are picked from the kernel arguments where they are used (kept per lane they cost ten registers)
#define T_LIM(q) (*(const volatile double __attribute__((address_space(3))) *)&s_u.dl[ac][q])   // the limits an event needs (rare)
#define T_LIMS(q) (ac ? A.dl[1][q] : A.dl[0][q])               // the two every step needs: from the kernel arguments (scalar)
    // the grids are wave-uniform: pinned in scalar registers (read back from LDS they would sit in 30 vector registers)
    auto sgrid = [&](int q) {
        const CutGrid gq = s_grid[q];
        auto u32 = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
        auto f64 = [&](double d) { const uint64_t b = (uint64_t)__double_as_longlong(d); return __longlong_as_double((long long)(((uint64_t)u32((uint32_t)(b >> 32)) << 32) | u32((uint32_t)b))); };
        return CutGrid{(int)u32((uint32_t)gq.off), (int)u32((uint32_t)gq.n), f64(gq.lo), f64(gq.inv_step)};
    };
    const CutGrid gDist = sgrid(0);   // (only the run-time-shape instance reads it)
    const int bear_n = __builtin_amdgcn_readfirstlane(s_grid[1].n);
    const double dist_hi2 = U(dist_hi2);
    const int bear_kA = __builtin_amdgcn_readfirstlane(U(bear_kA)), bear_kB = __builtin_amdgcn_readfirstlane(U(bear_kB));
    const int rm1[3] = {RM1_0 ? RM1_0 : (int)P.d_r[0] - 1, RM1_1 ? RM1_1 : (int)P.d_r[1] - 1, RM1_2 ? RM1_2 : (int)P.d_r[2] - 1};
    // which dynamic variable is heading / altitude / speed (the host checks that all three are there)
    // (the instance built for the shipped shape also knows the order: heading, altitude, speed with temporal-map rows 0, 1, 2 --
    // launch_terminal_propagate checks it; the selects below then fold)
    constexpr bool kShipped = RM1_0 == 35 && RM1_1 == 6 && RM1_2 == 4;
    const int kh = kShipped ? 0 : (P.d_ivar[0] == 3 ? 0 : (P.d_ivar[1] == 3 ? 1 : 2)), ka = kShipped ? 1 : (P.d_ivar[0] == 4 ? 0 : (P.d_ivar[1] == 4 ? 1 : 2)),
              ks = kShipped ? 2 : (P.d_ivar[0] == 5 ? 0 : (P.d_ivar[1] == 5 ? 1 : 2));
    const int drow[3] = {kShipped ? 0 : (int)P.d_row[0], kShipped ? 1 : (int)P.d_row[1], kShipped ? 2 : (int)P.d_row[2]};
    // asub2ind.m:13-14 over the step's start state: the strides of a transition node's current-bin parents (heading, altitude, speed
    // in the model's order) folded into the strides of the same variables as initial-state parents, once (wave-uniform, scalar) --
    // per step the column is then six multiply-adds per node instead of nine plus three 6-way selects
    // (18 wave-uniform strides: in LDS, read as five 16-byte loads next to the step's other LDS reads -- in scalar registers they and the
    // Philox key schedule were what spilled)
    __shared__ uint32_t s_cstr[20];
    if (threadIdx.x < 18) {
        const int k = threadIdx.x / 6, p = threadIdx.x % 6;
        uint32_t sv = P.d_stride_static[k][p];
        for (int q = 0; q < 3; q++) sv += ((int)P.d_ivar[q] == p) ? P.d_stride_cur[k][q] : 0u;
        s_cstr[threadIdx.x] = sv;
    }
    __syncthreads();
    // (compile_plan builds the compact rows for exactly the rows of 9 to 48 thresholds: a compile-time fact in the shipped-shape instance)
    const bool has_c8[3] = {rm1[0] > 8 && rm1[0] <= 48, rm1[1] > 8 && rm1[1] <= 48, rm1[2] > 8 && rm1[2] <= 48};   // wave-uniform
    const bool dist8 = kShipped || gDist.n <= 8;   // wave-uniform: the distance grid is compared squared (the shipped-shape instance is only launched on such a grid)
    const int C = EMGPU_TERMINAL_T0_ROW(A.cap);    // row of t = 0 in an aircraft's block of W = 2 C rows
    const size_t Wrows = (size_t)(2 * C);

    // ---- the lane's track
    bool active = false, failed = false;
    uint32_t L = 0;                 // track = 4 * encounter + role, role = 2 * (aircraft - 1) + (backward)
    int ac = 0, intent = 0;
    double dt_s = 1.0;
    Rng rng{0u, 0u, 0u, (uint32_t)A.seed, (uint32_t)(A.seed >> 32)};
    bool vdirty = false;            // the velocity's components are due from (speed, vang)
    bool fresh = false;             // a new track: its bins come from the full discretize, once
    // The bins of the three variables only an event changes (heading, altitude, speed: 0-based, one byte each), as the NEXT step will
    // see them: an event writes the bin it drew -- the dediscretized value lies in it (checked; else the full discretize) -- instead of
    // every step discretizing three values that have not changed since.  st[3..5] stay the step's own (createEncounter.m:187: `start`
    // and heading_discrete are fixed while a step is re-drawn).
    uint32_t pend = 0u;
    gptr_t thr = nullptr;
    double xy0 = 0, xy1 = 0, z_ft = 0, heading_deg = 0, prev_z_rec = 0, v0 = 0, v1 = 0, speed = 0, vang = 0, curr_hdg = 0;
    double sh, chh;
    int ii = 1, rows = 0, cnt = 0;  // rows: recorded so far; cnt: of them staged in LDS, not yet written
    int att = 0, st[6] = {0, 0, 0, 0, 0, 0};
    uint32_t colk[3] = {0u, 0u, 0u};   // the step's CPT columns; the row addresses are formed at the draw (nine 64-bit pointers kept per lane cost 18 registers)
    // ---- the wave's share of the queue (wave-uniform)
    uint32_t q_next = 0, q_end = 0;
    bool exhausted = false;
    for (;;) {
        // ---- idle lanes take the next tracks
        const uint64_t idle = __ballot(!active);
        const int nidle = __popcll(idle);
        if (!exhausted && nidle >= kRefillMin) {
            const uint32_t avail = q_end - q_next;
            uint32_t nb = 0;
            const bool grab = avail < (uint32_t)nidle;

            if (grab) {
                uint32_t b = 0;
                if (lane == 0) b = atomicAdd(U(queue), kChunk);
                nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
            }
            if (!active) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                const uint32_t t = rank < avail ? q_next + rank : nb + (rank - avail);
                if (t < total) {
                    // the track's start state (createEncounter.m:41-49, :96)
                    // the queue hands out the tracks role by role (all forward ownship tracks, then the backward ones, ...): the lanes in
                    // flight then gather from the tables of two or three trajectory models (one role, every intent) instead of all ten --
                    // measured: table fetches from HBM 14 -> 4 GB per 2 M encounters, -6 % run time
                    const uint32_t nn = (uint32_t)A.n;
                    const int role = (int)(t >= nn) + (int)(t >= 2u * nn) + (int)(t >= 3u * nn);
                    const uint32_t e = t - (uint32_t)role * nn;
                    L = 4u * e + (uint32_t)role;
                    ac = role >> 1;
                    dt_s = (role & 1) ? -1.0 : 1.0;
                    const uint64_t *ind = U(indices);
                    const uint64_t gidx = ind ? ind[e] : U(first_index) + (uint64_t)e;
                    rng.c0 = (uint32_t)gidx; rng.c1 = (uint32_t)(gidx >> 32);
                    const double *g = U(geo) + (size_t)e * 12 + ac * 6;
                    intent = (int)g[5];
                    thr = (gptr_t)U(thr_base)[U(model_of)[L]];
                    xy0 = g[0]; xy1 = g[1]; z_ft = g[2]; heading_deg = g[4]; prev_z_rec = 0;
                    speed = g[3];   // norm(v_ft_s), carried: the velocity is only ever speed * (cosd, sind) of a direction (its norm to 1e-16)
                    // The velocity is (speed, direction): it is only ever set to speed * (cosd, sind)(heading) (:96, :238-247) and rotated by the
                    // step's turn (:262), so it IS speed * (cosd, sind)(vang) with vang = the heading it was last set to + the turns since, up to
                    // rounding (1e-14 degrees) -- and atan2d(v) (createEncounter.m:176) is vang.  Its components are only read by the next step's
                    // move: ONE place computes them (a new track, a speed event and a turn all mark them due) instead of a sincosd at each.
                    // (v = 0 would give atan2d = 0: speeds are clamped to minVel > 0.)
                    vang = heading_deg; vdirty = true; fresh = true;
                    ii = 1; rows = 0; cnt = 0; att = 0; failed = false;
                    active = true;

                }
            }
            if (grab) { q_next = nb + ((uint32_t)nidle - avail); q_end = nb + kChunk; }
            else q_next += (uint32_t)nidle;
            exhausted = q_next >= total;
        }
        if (__ballot(active) == 0ull) break;   // (the queue is exhausted: an all-idle wave always tries to refill)

        // ---- ONE attempt of the lane's current step: a lane whose draw produced an invalid event (createEncounter.m:218-262 re-draws the
        // step) comes round again with att + 1 while its neighbours start their next step
        bool done = false;
        const bool bck = dt_s < 0.0;
        if (active) do {
            if (att == 0) {
                // ---- the step begins: record the state, move, discretize (createEncounter.m:156-200)
                if (rows >= A.cap) { failed = true; done = true; break; }
                if (vdirty) {
                    t_sincosd(vang, sh, chh);
                    v0 = chh * speed; v1 = sh * speed;
                    vdirty = false;
                }
                // forward lanes fill their staging area upwards, backward lanes downwards: either way it holds ascending rows of the
                // joined track; the backward track's row 0 is the forward track's (bck(1, 2:end), createEncounter.m:77) and is not kept
                float *rec = mine + 5 * (bck ? kRows - 1 - cnt : cnt);
                rec[0] = (float)xy0; rec[1] = (float)xy1; rec[4] = (float)speed;
                xy0 += (v0 * dt_s) * t_k(1.0 / 6076.1154855643);
                xy1 += (v1 * dt_s) * t_k(1.0 / 6076.1154855643);
                curr_hdg = (speed > 0.0) ? t_mod360(vang) : 0.0;
                double rec_z = z_ft;
                if (ii > 1) {
                    const double alt_diff = z_ft - prev_z_rec;
                    rec_z = prev_z_rec + t_sign(alt_diff) * fmin(T_LIMS(4), fabs(alt_diff));
                }
                prev_z_rec = rec_z;
                rec[2] = (float)rec_z; rec[3] = (float)curr_hdg;
                cnt += (bck && rows == 0) ? 0 : 1;
                rows++;
                // CreateStartDistribution (0-based bins), createEncounter.m:268-294
                const double d2_nm = xy0 * xy0 + xy1 * xy1;
                st[0] = intent - 1;
                st[1] = (dist8 ? t_discretize8(d2_nm, s_cut8sq) : t_discretize(sqrt(d2_nm), s_bnd, gDist)) - 1;     // wave-uniform choices
                int kb0 = st[2];
                if (fresh) {   // a track's first step: the full discretize (once); from then on events keep `pend` and the bearing is walked
                    const CutGrid gB = s_grid[1];
                    kb0 = t_bearing_bin(xy0, xy1, s_bnd + gB.off + 1, s_dir, gB.n, gB.lo, gB.inv_step) - 1;
                    pend = (uint32_t)(t_discretize(heading_deg, s_bnd, s_grid[2]) - 1) | ((uint32_t)(t_discretize(z_ft, s_bnd, s_grid[3]) - 1) << 8) |
                           ((uint32_t)(t_discretize(speed, s_bnd, s_grid[4]) - 1) << 16);
                    fresh = false;
                }
                st[2] = t_bearing_walk(xy0, xy1, s_bnd + kBndStride + 1, s_dir, bear_n, bear_kA, bear_kB, kb0) - 1;
                st[3] = (int)(pend & 0xFFu); st[4] = (int)((pend >> 8) & 0xFFu); st[5] = (int)(pend >> 16);   // (speed: norm(v_ft_s) is `speed`)
                // CPT column of each dynamic variable (asub2ind.m:13-14 as strides); topological position == variable id
                uint32_t cstr[3][6];
                {
                    int z = 0;
                    asm volatile("" : "+v"(z));   // (keeps these loads inside the loop: hoisted they would be 18 registers for its whole length)
#pragma unroll
                    for (int q = 0; q < 18; q++) cstr[q / 6][q % 6] = s_cstr[z + q];
                }
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
            rng.attempt = (L & 3u) + 4u * (uint32_t)att;
            uint32_t xw[3], spare;
            {   // block = the step, word = the variable's row of the temporal map: one Philox call for the three draws
                // (the round keys are scalar adds at the call, not 14 registers kept for the loop)
                { uint32_t k0 = (uint32_t)A.seed, k1 = (uint32_t)(A.seed >> 32); asm volatile("" : "+s"(k0), "+s"(k1)); rng.k0 = k0; rng.k1 = k1; }
                const uint4 tw = rng.block(11u /* TERM_TRANS */, 0u, (uint32_t)ii);
#pragma unroll
                for (int k = 0; k < 3; k++) xw[k] = word_of(tw, drow[k]);
                spare = tw.w;   // (the temporal map has three rows: words 0-2)
            }
            gptr_t row[3], piv[3], c8[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                row[k] = thr + (P.d_off[k] - P.d_off[0]) + (size_t)colk[k] * (uint32_t)rm1[k];
                piv[k] = (rm1[k] > 8 && rm1[k] <= 48) ? thr + (P.d_pivoff[k] - P.d_off[0]) + (size_t)colk[k] * 8u : thr;   // wave-uniform
                c8[k] = has_c8[k] ? thr + (P.d_c8off[k] - P.d_off[0]) + (size_t)colk[k] * 8u : thr;
            }
            const Draw3 nb3 = t_draw3(row, piv, c8, has_c8, rm1, xw);
            // events in ascending variable id (createEncounter.m:218-262): heading (4), altitude (5), speed (6); an invalid altitude
            // or speed bin makes the step be drawn again -- the events applied before it stay applied, as in the reference's loop
            bool resample = false;
            const int dH = kh == 0 ? nb3.bin[0] : (kh == 1 ? nb3.bin[1] : nb3.bin[2]), dA = ka == 0 ? nb3.bin[0] : (ka == 1 ? nb3.bin[1] : nb3.bin[2]),
                      dS = ks == 0 ? nb3.bin[0] : (ks == 1 ? nb3.bin[1] : nb3.bin[2]);
#include "emgpu_term_events.h"
            if (resample) { att++; break; }
#include "emgpu_term_endstep.h"
        } while (false);
        if (done) {
            if (failed && !U(quiet)) atomicOr(U(status), 1u);
            U(rows)[L] = failed ? -rows - 1 : rows;
            active = false;
        }
        // ---- staged rows leave for memory: a lane whose staging area is full, or whose track has just ended, hands its rows to the wave,
        // which writes them as one contiguous piece (lane j = dword j of the piece)
        const bool fl = cnt == kRows || (done && cnt > 0);
        const uint64_t fm = __ballot(fl);
        if (fm != 0ull) {
            // Round 5: FOUR PIECES PER TRIP.  Round 4 walked the flagged lanes one by one (find the lane, read its descriptor into scalar
            // registers, 30 lanes copy its piece a dword each: 9.2 trips per wave-iteration, each a scalar dependency chain through an LDS
            // round trip -- measured: 23 % of the kernel's time, none of it the stores themselves).  Now the flagged lanes put their
            // descriptors into a small table in lane order and the wave copies four pieces per trip: each half-wave serves two, lane i of
            // the half = dword i of the piece, both pieces' reads in flight together.  The stores stay what the memory system takes for free:
            // 4 bytes per lane, a piece's lanes contiguous (measured on the way: sixteen pieces per pass as 16-byte parts, four lanes per
            // piece -- non-temporal 20.2 ms per 2 M encounters, plain stores 16.8, the pass without its stores 14.3; round 4's loop 17.5).
            if (fl) {
                // rows [rows - cnt, rows) of this direction: the piece starts at the joined track's row C + (rows - cnt) (forward) or
                // C - (rows - 1) (backward)
                const size_t r0 = (size_t)(bck ? C - (rows - 1) : C + (rows - cnt));
                const uint64_t dst = (uint64_t)(A.traj + ((size_t)(L >> 1) * Wrows + r0) * 5);
                // the piece's address; bits 48-63: its source in the staging area (11 bits: dword) and its length (5 bits: dwords)
                // (measured and dropped: 16-byte descriptors with the address carried from piece to piece -- fewer instructions, +4 % time)
                const uint32_t dhi = (uint32_t)(dst >> 32) | ((uint32_t)(lane * kLaneStride + (bck ? 5 * (kRows - cnt) : 0)) << 16) | ((uint32_t)(5 * cnt) << 27);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
                desc[rank] = make_uint2((uint32_t)dst, dhi);
                cnt = 0;
            }
            __builtin_amdgcn_wave_barrier();
            const int np = __popcll(fm);
            const int half = lane >> 5, li = lane & 31;
            typedef float __attribute__((address_space(1))) gf_t;   // (global, not flat: the address is assembled from integers)
#ifndef EMGPU_TERM_FLUSH_SLOTS
#define EMGPU_TERM_FLUSH_SLOTS 4
#endif
            constexpr int kSlots = EMGPU_TERM_FLUSH_SLOTS;   // pieces per half-wave and trip
            for (int base = 0; base < np; base += 2 * kSlots) {

                uint2 d[kSlots];
                float v[kSlots];
                bool w[kSlots];
#pragma unroll
                for (int s_ = 0; s_ < kSlots; s_++) {
                    const int pp = base + 2 * s_ + half;
                    d[s_] = make_uint2(0u, 0u);   // (length 0: nothing to copy)
                    if (pp < np) d[s_] = desc[pp];
                }
#pragma unroll
                for (int s_ = 0; s_ < kSlots; s_++) {
                    w[s_] = li < (int)(d[s_].y >> 27);
                    v[s_] = 0.f;
                    if (w[s_]) v[s_] = stage[((d[s_].y >> 16) & 0x7FFu) + li];
                }
#pragma unroll
                for (int s_ = 0; s_ < kSlots; s_++) {
                    gf_t *q = (gf_t *)(((uint64_t)(d[s_].y & 0xFFFFu) << 32) | d[s_].x) + li;
                    // nontemporal: a piece is a part of a line that nobody reads back here; written through L2 as ordinary stores the 260 000
                    // half-written lines of the lanes in flight crowd the trajectory tables out of it (round 4, measured: table fetches 13 -> 30 GB
                    // per 2 M encounters, +3 % run time)
                    if (w[s_]) __builtin_nontemporal_store(v[s_], q);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// measuring builds only: read (and clear) the path counters of the launches so far; returns 0 in a normal build
int terminal_debug_counters(unsigned long long *out, int n) {
    (void)out; (void)n;
    return 0;
}

hipError_t launch_terminal_propagate(const EmgpuPlan &P, const EmgpuTermRun &A, hipStream_t s, const char **name) {
    *name = "k_terminal_propagate";
    if (A.n <= 0) return hipSuccess;
    if (4 * A.n >= (int64_t)1 << 31) return hipErrorInvalidValue;   // the queue counts tracks in 32 bits
    static const bool generic_only = getenv("EMGPU_DEBUG_TERM_GENERIC") != nullptr;   // tests: the run-time-shape instance on the shipped shape
    bool shipped_order = P.d_ivar[0] == 3 && P.d_ivar[1] == 4 && P.d_ivar[2] == 5 && P.d_row[0] == 0 && P.d_row[1] == 1 && P.d_row[2] == 2;
    if ((int)P.i_nb[1] - 2 > 8 || (int)P.i_nb[4] - 2 > 8 || (int)P.i_nb[5] - 2 > 8) shipped_order = false;   // distance, altitude and speed grids compared against eight padded cut points
    for (int k = 0; k < 3; k++)       // the folded strides of the instance's 24-bit multiply-adds
        for (int p = 0; p < 6; p++) {
            uint64_t sv = P.d_stride_static[k][p];
            for (int q = 0; q < 3; q++) sv += ((int)P.d_ivar[q] == p) ? P.d_stride_cur[k][q] : 0u;
            if (sv >= (1u << 24)) shipped_order = false;
        }
    const bool shipped = !generic_only && shipped_order && P.d_r[0] == 36 && P.d_r[1] == 7 && P.d_r[2] == 5;
    // persistent workgroups: as many as the device holds at once (a wave takes its tracks from the queue), fewer for a small batch
    int dev = 0, cus = 0, per_cu = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e == hipSuccess)
        e = shipped ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_terminal_propagate<35, 6, 4>, 256, 0)
                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_terminal_propagate<0, 0, 0>, 256, 0);
    if (e != hipSuccess) return e;
    if (per_cu < 1) per_cu = 1;
    const int64_t need = (4 * A.n + 255) / 256;
    const int64_t blocks = need < (int64_t)cus * per_cu ? need : (int64_t)cus * per_cu;
    e = hipMemsetAsync(A.queue, 0, sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    if (shipped) {
        *name = "k_terminal_propagate<35,6,4>";
        hipLaunchKernelGGL((k_terminal_propagate<35, 6, 4>), dim3((unsigned)blocks), dim3(256), 0, s, P, A);
    } else {
        hipLaunchKernelGGL((k_terminal_propagate<0, 0, 0>), dim3((unsigned)blocks), dim3(256), 0, s, P, A);
    }
    return hipGetLastError();
}

} // namespace emgpu
