// tools/ubench/vmm_repro.hip -- the HIP virtual-memory calls the trace pool uses (csrc/emgpu_host.cpp), alone: K blocks of A GiB are built (address range on a
// 1 GiB boundary, 1 GiB chunks), released, then one block of B GiB is built.  Prints every step (a crash names its call).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); printf("%s -> %d\n", #x, (int)e_); fflush(stdout); if (e_ != hipSuccess) exit(1); } while (0)
struct Blk { void *va; size_t total; std::vector<hipMemGenericAllocationHandle_t> h; };
static Blk build(size_t gib) {
    const size_t chunk = (size_t)1 << 30, total = gib << 30;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    Blk b; b.total = total; b.va = nullptr;
    CK(hipMemAddressReserve(&b.va, total, chunk, nullptr, 0));
    printf("  va %p\n", b.va);
    for (size_t o = 0; o < total; o += chunk) {
        hipMemGenericAllocationHandle_t hnd;
        if (hipMemCreate(&hnd, chunk, &prop, 0) != hipSuccess) { printf("hipMemCreate failed at %zu\n", o); exit(1); }
        b.h.push_back(hnd);
        if (hipMemMap((char *)b.va + o, chunk, 0, hnd, 0) != hipSuccess) { printf("hipMemMap failed at %zu\n", o); exit(1); }
    }
    printf("  mapped %zu chunks\n", b.h.size()); fflush(stdout);
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(b.va, total, &acc, 1));
    CK(hipMemset(b.va, 1, total)); CK(hipDeviceSynchronize());
    if (getenv("VMM_COPY")) {   // a device -> pinned copy out of the block on a stream of its own (what the host path does with its chunk buffers)
        static void *hp = nullptr; static hipStream_t st = nullptr;
        const size_t nb = (size_t)atoi(getenv("VMM_COPY")) << 20;   // MiB; more than 1 024 spans two physical chunks
        if (!hp) { CK(hipHostMalloc(&hp, (size_t)2 << 30, 0)); CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); }
        CK(hipMemcpyAsync(hp, b.va, nb, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
    }
    return b;
}
static void release(Blk &b) {
    const size_t chunk = (size_t)1 << 30;
    for (size_t i = 0; i < b.h.size(); i++) { hipError_t e1 = hipMemUnmap((char *)b.va + i * chunk, chunk), e2 = hipMemRelease(b.h[i]); if (e1 || e2) printf("unmap %d release %d\n", e1, e2); }
    if (!getenv("VMM_KEEP_VA")) CK(hipMemAddressFree(b.va, b.total));
}
int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 4; const size_t A = argc > 2 ? atoi(argv[2]) : 2, B = argc > 3 ? atoi(argv[3]) : 37;
    std::vector<Blk> bl;
    for (int k = 0; k < K; k++) bl.push_back(build(A));
    for (auto &b : bl) release(b);
    printf("released\n"); fflush(stdout);
    Blk big = build(B);
    release(big);
    printf("done\n");
    return 0;
}
