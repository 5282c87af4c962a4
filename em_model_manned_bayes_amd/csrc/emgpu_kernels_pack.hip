// emgpu_kernels_pack.hip -- event lists [n][cap] -> one packed run of rows, for the host path (emgpu_sample_dbn_host): what crosses PCIe
// is sum(min(ev_count, cap)) rows instead of n x cap (out_events{i} of UncorEncounterModel.m:283 has ~43 rows at T = 240, the capacity
// the class layer asks for is 256).  Three launches: per-workgroup row counts (one wave = 64 lists), one exclusive scan over the
// workgroups, then every wave copies its 64 lists, list after list with the lanes across a list's rows -- reads and writes are both
// contiguous 8-byte rows.  Bound: HBM, and small beside the sampler itself (a list is ~350 B of a trajectory's 3.6 KB).
#include <hip/hip_runtime.h>

#include "emgpu_launch.h"

namespace emgpu {

// scratch: bsum[0] = total rows (u64 in words 0-1), bsum[2 + b] = rows of workgroup b (256 lists), then their exclusive scan in place
__global__ void __launch_bounds__(256) k_pack_count(int64_t n, uint32_t cap, const uint32_t *cnt, uint32_t *bsum) {
    __shared__ uint32_t s_w[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t c = i < n ? min(cnt[i], cap) : 0u;
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) bsum[2 + blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ void __launch_bounds__(1024) k_pack_scan(uint32_t nb, uint32_t *bsum) {
    __shared__ uint64_t s_part[1024];
    const uint32_t t = threadIdx.x, per = (nb + 1023u) / 1024u, lo = min(t * per, nb), hi = min(lo + per, nb);
    uint64_t sum = 0;
    for (uint32_t b = lo; b < hi; b++) sum += bsum[2 + b];
    s_part[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        const uint64_t v = t >= d ? s_part[t - d] : 0ull;
        __syncthreads();
        s_part[t] += v;
        __syncthreads();
    }
    uint64_t run = s_part[t] - sum;
    for (uint32_t b = lo; b < hi; b++) { const uint32_t c = bsum[2 + b]; bsum[2 + b] = (uint32_t)run; run += c; }   // (a chunk holds < 2^32 rows: checked by the launcher)
    if (t == 1023u) *reinterpret_cast<uint64_t *>(bsum) = s_part[1023];
}

__global__ void __launch_bounds__(256) k_pack_rows(int64_t n, uint32_t cap, const uint32_t *cnt, const uint64_t *ev, const uint32_t *bsum, uint64_t *packed) {
    __shared__ uint32_t s_w[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t c = i < n ? min(cnt[i], cap) : 0u;
    uint32_t inc = c;                                       // inclusive scan over the wave's 64 lists
    for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(inc, d, 64); if (lane >= (uint32_t)d) inc += v; }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    uint32_t base = bsum[2 + blockIdx.x];
    for (uint32_t q = 0; q < w; q++) base += s_w[q];
    const uint32_t off = base + inc - c;                    // first packed row of this lane's list
    const int64_t i0 = (int64_t)blockIdx.x * 256 + w * 64;  // the wave's first list
    for (int l = 0; l < 64; l++) {                          // list after list, lanes across rows
        const uint32_t cl = __shfl(c, l, 64), ol = __shfl(off, l, 64);
        const uint64_t *src = ev + (size_t)(i0 + l) * cap;
        for (uint32_t r = lane; r < cl; r += 64) packed[(size_t)ol + r] = src[r];
    }
}

size_t pack_scratch_words(int64_t n) { return 4 + (size_t)((n + 255) / 256); }

hipError_t launch_pack_events(int64_t n, uint32_t cap, const uint32_t *cnt, const uint64_t *ev, uint32_t *scratch, uint64_t *packed, hipStream_t s) {
    if (n <= 0) return hipMemsetAsync(scratch, 0, 2 * sizeof(uint32_t), s);
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_pack_count, dim3(nb), dim3(256), 0, s, n, cap, cnt, scratch);
    hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(1024), 0, s, (uint32_t)nb, scratch);
    hipLaunchKernelGGL(k_pack_rows, dim3(nb), dim3(256), 0, s, n, cap, cnt, ev, scratch, packed);
    return hipGetLastError();
}

} // namespace emgpu
