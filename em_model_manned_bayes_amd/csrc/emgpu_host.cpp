// emgpu_host.cpp -- the parts of the C ABI that own memory on behalf of the caller (round 6):
//   * the trace pool: emgpu_trace_alloc / _out / _report / _free -- device memory for the sampler's outputs whose PLACEMENT has been
//     measured with the caller's own launch (profiles/r05_placement_probe.txt: the same launch writes one 36 GB allocation in 6.0 ms
//     and another in 7.1 ms);
//   * the pinned pool: emgpu_host_alloc / _free;
//   * emgpu_sample_dbn_host as a pipeline: chunk k's kernel | chunk k-1's copy over PCIe | chunk k-2's copy into the caller's arrays.
// Reference semantics: the loop over samples of UncorEncounterModel.m:244-300 and the host arrays it returns (:283-300).
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <system_error>
#include <thread>

#include "emgpu_internal.hpp"

struct emgpu_trace {
    emgpu_ctx::TraceBlock blk;
    emgpu_sample_out out{};
    emgpu_trace_report_t rep{};
};

namespace {
using Clock = std::chrono::steady_clock;
double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }
size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ------------------------------------------------------------------------------------------------ device blocks
// How a block of the trace pool is obtained (round 6, tools/placement_probe5.py, profiles/r06_placement_probe.txt).  The same launch writes a
// 36 GB trace in 5.9, 6.6 or 7.0 ms depending on the allocation.  Blocks that hipMalloc hands out are mostly of the 6.6 ms kind, now and then
// of the others; ONE ADDRESS RANGE BACKED BY SEPARATELY CREATED 1 GiB PHYSICAL CHUNKS (hipMemAddressReserve + hipMemCreate + hipMemMap) is of the
// 5.9 ms kind four to six times out of six, of the 7.0 ms kind the rest (chunks of 256 MiB - 2 GiB alike, 4 GiB chunks like hipMalloc; where the
// range starts -- on a 1 GiB boundary or 2 MiB off one -- makes no difference: measured both ways).  Why is not known; the allocator does not need
// to know: blocks of 1 GiB and more are built that way (falling back to hipMalloc where the virtual-memory calls fail), smaller ones come from
// hipMalloc, and emgpu_trace_alloc MEASURES its candidates -- candidate 0 a plain hipMalloc block, so that the report shows what a caller's own
// allocation would have got.
// EMGPU_TRACE_ALLOC (read once; experiments) = "plain": hipMalloc only; "contiguous": hipExtMallocWithFlags(hipDeviceMallocContiguous);
// "vmm:<chunk MiB>": another chunk size.
struct VmmBlock { size_t bytes = 0, chunk = 0; std::vector<hipMemGenericAllocationHandle_t> handles; };
std::mutex g_vmm_mu;
std::map<void *, VmmBlock> g_vmm;
struct AllocMode { int mode; size_t chunk; };
const AllocMode &alloc_mode_once() {
    static const AllocMode am = [] {   // (a function-local static: initialised once, also when several host threads come here together)
        AllocMode a{3, (size_t)1 << 30};   // automatic: a range over 1 GiB chunks for blocks of 1 GiB and more, hipMalloc below (and as the fallback)
        const char *e = getenv("EMGPU_TRACE_ALLOC");
        if (e && !strncmp(e, "plain", 5)) a.mode = 0;
        if (e && !strncmp(e, "contiguous", 10)) a.mode = 1;
        if (e && !strncmp(e, "vmm", 3)) {
            a.mode = 2;
            if (e[3] == ':' && atol(e + 4) > 0) a.chunk = (size_t)atol(e + 4) << 20;
        }
        return a;
    }();
    return am;
}
int alloc_mode(size_t *chunk) {
    const AllocMode &a = alloc_mode_once();
    if (chunk) *chunk = a.chunk;
    return a.mode;
}
bool vmm_block(size_t bytes, void **p) {
    size_t chunk = 0;
    (void)alloc_mode(&chunk);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || !gran) { (void)hipGetLastError(); return false; }
    chunk = round_up(chunk, gran);
    const size_t total = round_up(bytes, chunk);
    void *va = nullptr;   // (hipMemAddressReserve returns 2 MiB-aligned ranges whatever alignment it is asked for: tools/ubench/vmm_repro.hip)
    if (hipMemAddressReserve(&va, total, 0, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
    VmmBlock B;
    B.bytes = total; B.chunk = chunk;
    bool ok = true;
    for (size_t o = 0; o < total && ok; o += chunk) {
        hipMemGenericAllocationHandle_t hnd;
        if (hipMemCreate(&hnd, chunk, &prop, 0) != hipSuccess) { ok = false; break; }
        B.handles.push_back(hnd);
        if (hipMemMap((char *)va + o, chunk, 0, hnd, 0) != hipSuccess) { ok = false; break; }
    }
    if (ok) {
        hipMemAccessDesc acc;
        memset(&acc, 0, sizeof acc);
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        ok = hipMemSetAccess(va, total, &acc, 1) == hipSuccess;
    }
    if (!ok) {
        (void)hipGetLastError();
        for (size_t i = 0; i < B.handles.size(); i++) { (void)hipMemUnmap((char *)va + i * chunk, chunk); (void)hipMemRelease(B.handles[i]); }
        (void)hipMemAddressFree(va, total);
        (void)hipGetLastError();
        return false;
    }
    std::lock_guard<std::mutex> lk(g_vmm_mu);
    g_vmm[va] = std::move(B);
    *p = va;
    return true;
}
bool device_block(size_t bytes, void **p, bool plain = false) {   // an allocation that reports failure instead of throwing (a candidate too many is not an error)
    *p = nullptr;
    const int mode = plain ? 0 : alloc_mode(nullptr);
    if (mode == 2) return vmm_block(bytes, p);
    if (mode == 3 && bytes >= ((size_t)1 << 30) && vmm_block(bytes, p)) return true;
    const hipError_t e = mode == 1 ? hipExtMallocWithFlags(p, bytes, hipDeviceMallocContiguous) : hipMalloc(p, bytes);
    if (e == hipSuccess) return true;
    (void)hipGetLastError();
    *p = nullptr;
    return false;
}
void device_release(void *p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_vmm_mu);
        auto it = g_vmm.find(p);
        if (it != g_vmm.end()) {
            VmmBlock &B = it->second;
            for (size_t i = 0; i < B.handles.size(); i++) { (void)hipMemUnmap((char *)p + i * B.chunk, B.chunk); (void)hipMemRelease(B.handles[i]); }
            // The physical chunks go back; the ADDRESS RANGE does not (unless EMGPU_VMM_FREE_VA is set).  A HIP runtime (the 7.0 build PyTorch wheels
            // bundle) crashes in hipMemMap -- VirtualGPU::submitVirtualMap -- when a new range overlaps one whose block had been the source of
            // hipMemcpyAsync calls before it was released (tools/copy_placement_probe.py; the 7.2 system runtime does not).  A reservation costs
            // address space only (47 bits of it: a thousand 36 GB traces are 36 TiB), so ranges are simply never handed back for re-use.
            static const bool free_va = getenv("EMGPU_VMM_FREE_VA") != nullptr;
            if (free_va) (void)hipMemAddressFree(p, B.bytes);
            g_vmm.erase(it);
            return;
        }
    }
    (void)hipFree(p);
}
void pool_release(emgpu_ctx *ctx) {
    for (auto &b : ctx->trace_pool) device_release(b.p);
    ctx->trace_pool.clear();
}
// a free block of the pool that fits (and is not more than a quarter too large), or a fresh allocation; {nullptr} when neither exists
emgpu_ctx::TraceBlock pool_take(emgpu_ctx *ctx, size_t bytes, bool *from_pool, bool plain = false) {
    int best = -1;
    for (int i = 0; i < (int)ctx->trace_pool.size(); i++) {
        const auto &b = ctx->trace_pool[(size_t)i];
        if (b.bytes >= bytes && b.bytes <= bytes + bytes / 4 + (1u << 20) && (best < 0 || b.bytes < ctx->trace_pool[(size_t)best].bytes)) best = i;
    }
    if (from_pool) *from_pool = best >= 0;
    if (best >= 0) {
        emgpu_ctx::TraceBlock b = ctx->trace_pool[(size_t)best];
        ctx->trace_pool.erase(ctx->trace_pool.begin() + best);
        return b;
    }
    emgpu_ctx::TraceBlock b;
    if (!device_block(bytes, &b.p, plain)) {   // out of memory: give the pool's idle blocks back and try once more
        HIP_OK(hipStreamSynchronize(ctx->stream));
        pool_release(ctx);
        if (!device_block(bytes, &b.p, plain)) return b;
    }
    b.bytes = bytes;
    return b;
}

// ------------------------------------------------------------------------------------------------ trace layout
struct TraceLayout {
    size_t o_ib = 0, o_iv = 0, o_db = 0, o_dv = 0, o_ec = 0, o_ev = 0, o_at = 0, bytes = 0;
    int64_t ld = 0;
};
TraceLayout trace_layout(const Model &m, const emgpu_sample_params *p, uint32_t want) {
    constexpr size_t kA = 2u << 20;   // every array of a trace starts on a 2 MiB boundary
    TraceLayout L;
    L.ld = (int64_t)round_up((size_t)std::max<int64_t>(p->n, 1), 1024);
    const size_t ld = (size_t)L.ld, ni = (size_t)m.n_initial, nd = (size_t)m.n_dyn(), G4 = ((size_t)p->sample_time + 3) / 4;
    size_t o = 0;
    auto put = [&](size_t bytes) { const size_t at = o; o = round_up(o + std::max<size_t>(bytes, 1), kA); return at; };
    if (want & EMGPU_TRACE_DENSE) { L.o_dv = put(G4 * nd * ld * 16); L.o_db = put(G4 * nd * ld * 4); }
    if (want & EMGPU_TRACE_INIT) { L.o_iv = put(ni * ld * 4); L.o_ib = put(ni * ld); }
    if (want & EMGPU_TRACE_EVENTS) { L.o_ev = put(ld * (size_t)p->event_cap * 8); L.o_ec = put(ld * 4); }
    if (want & EMGPU_TRACE_ATTEMPTS) L.o_at = put(ld * 4);
    L.bytes = std::max<size_t>(o, kA);
    return L;
}
void trace_bind(const TraceLayout &L, uint32_t want, void *base, emgpu_sample_out *o) {
    char *b = (char *)base;
    memset(o, 0, sizeof *o);
    if (want & EMGPU_TRACE_DENSE) { o->dyn_val = (float *)(b + L.o_dv); o->dyn_bin = (uint32_t *)(b + L.o_db); }
    if (want & EMGPU_TRACE_INIT) { o->init_val = (float *)(b + L.o_iv); o->init_bin = (uint8_t *)(b + L.o_ib); }
    if (want & EMGPU_TRACE_EVENTS) { o->events = (emgpu_event *)(b + L.o_ev); o->ev_count = (uint32_t *)(b + L.o_ec); }
    if (want & EMGPU_TRACE_ATTEMPTS) o->attempts = (int32_t *)(b + L.o_at);
    o->ld = L.ld;
    o->col_offset = 0;
}

struct Events {   // a few HIP events, destroyed on every path out
    std::vector<hipEvent_t> e;
    explicit Events(int n) : e((size_t)n, nullptr) { for (auto &x : e) HIP_OK(hipEventCreate(&x)); }
    ~Events() { for (auto x : e) if (x) (void)hipEventDestroy(x); }
    hipEvent_t operator[](int i) const { return e[(size_t)i]; }
};

// `timed` launches of the caller's call into `o` after `warm` untimed ones: ms per launch (HIP events on the ctx stream)
float time_launches(emgpu_ctx *ctx, const emgpu_model *m, const emgpu_sample_params *p, const emgpu_sample_out *o, int warm, int timed, const Events &ev) {
    auto launch = [&]() {
        const int rc = emgpu_sample_dbn_device(ctx, m, p, o);
        if (rc != EMGPU_OK) throw Error(rc, g_err);
    };
    for (int i = 0; i < warm; i++) launch();
    HIP_OK(hipEventRecord(ev[0], ctx->stream));
    for (int i = 0; i < timed; i++) launch();
    HIP_OK(hipEventRecord(ev[1], ctx->stream));
    HIP_OK(hipEventSynchronize(ev[1]));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, ev[0], ev[1]));
    return ms / (float)timed;
}

bool is_pinned(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

int host_threads() {
    static const int n = [] {
        const char *e = getenv("EMGPU_HOST_THREADS");
        int v = e ? atoi(e) : 0;
        if (v < 1) v = (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
        return std::min(v, 64);
    }();
    return n;
}
template <typename F>
void run_parallel(int T, F fn) {   // fn(t) for t = 0..T-1, fn(0) on the calling thread; fn does not throw (memcpy loops)
    std::vector<std::thread> th;
    int started = 1;
    try {
        for (int t = 1; t < T; t++) { th.emplace_back(fn, t); started = t + 1; }
    } catch (const std::system_error &) {}   // the host will not give another thread: the parts not started run here (a joinable thread must not be destroyed)
    fn(0);
    for (int t = started; t < T; t++) fn(t);
    for (auto &x : th) x.join();
}
} // namespace

void ctx_release_host_side(emgpu_ctx *ctx, bool everything) {
    pool_release(ctx);
    if (everything) { for (void *p : ctx->device_blocks) device_release(p); ctx->device_blocks.clear(); }
    for (auto &b : ctx->chunk_buf) { device_release(b.p); b = emgpu_ctx::TraceBlock(); }
    for (auto &s : ctx->h_stage) { if (s) (void)hipHostFree(s); s = nullptr; }
    ctx->h_stage_cap = 0;
    for (auto it = ctx->host_pool.begin(); it != ctx->host_pool.end();) {
        if (!it->in_use || everything) { (void)hipHostFree(it->p); it = ctx->host_pool.erase(it); }
        else ++it;
    }
    if (everything) {
        if (ctx->h_total) (void)hipHostFree(ctx->h_total);
        ctx->h_total = nullptr;
        if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
        ctx->copy_stream = nullptr;
    }
}

extern "C" {

// ================================================================================================ the trace pool
int emgpu_trace_alloc(emgpu_ctx *ctx, const emgpu_model *h, const emgpu_sample_params *p, uint32_t want, int32_t candidates, emgpu_trace **out) {
    EMGPU_TRY
    if (!ctx || !h || !p || !out) return fail(EMGPU_ERR_ARG, "null argument");
    if (p->n < 0 || p->sample_time < 1) return fail(EMGPU_ERR_ARG, "n < 0 or sample_time < 1");
    if (!(want & (EMGPU_TRACE_INIT | EMGPU_TRACE_DENSE | EMGPU_TRACE_EVENTS | EMGPU_TRACE_ATTEMPTS)) || (want & ~15u)) return fail(EMGPU_ERR_ARG, "want: a combination of EMGPU_TRACE_*");
    if ((want & EMGPU_TRACE_EVENTS) && p->event_cap < 1) return fail(EMGPU_ERR_ARG, "EMGPU_TRACE_EVENTS needs event_cap >= 1");
    if (candidates < 0 || candidates > 8) return fail(EMGPU_ERR_ARG, "candidates outside 0..8");
    CTX_LOCK(ctx);
    HIP_OK(hipSetDevice(ctx->device));
    const TraceLayout L = trace_layout(h->m, p, want);
    std::unique_ptr<emgpu_trace> t(new emgpu_trace());
    t->rep.bytes = (int64_t)L.bytes;
    t->rep.ld = L.ld;
    const bool automatic = candidates == 0;
    int target = automatic ? (L.bytes < ((size_t)1 << 30) ? 1 : 6) : candidates;   // (no early stop: a candidate costs a quarter of a second, and two
    if (p->n == 0) target = 1;                                                      //  medium ones that agree say nothing about a fast one further on)

    bool from_pool = false;
    std::vector<emgpu_ctx::TraceBlock> cands;
    // candidate 0 of a probe is what hipMalloc hands a caller (the report's first_allocation_ms); the others are the library's own kind
    cands.push_back(pool_take(ctx, L.bytes, &from_pool, /*plain=*/target > 1));
    if (!cands[0].p) return fail(EMGPU_ERR_HIP, "emgpu_trace_alloc: out of device memory (" + std::to_string(L.bytes) + " bytes)");
    auto give_up = [&]() { for (auto &c : cands) device_release(c.p); cands.clear(); };
    try {
        if (from_pool && (cands[0].probed || target == 1)) {   // placed by an earlier call (or the caller does not want a probe): take it as it is
            t->rep.candidates = 1;
            t->rep.reused = 1;
            t->rep.kept_ms = cands[0].ms;
        } else if (target == 1) {
            t->rep.candidates = 1;
        } else {
            auto room_for_one_more = [&]() {
                size_t fr = 0, tot = 0;
                if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return false; }
                return fr >= L.bytes + ((size_t)4 << 30);
            };
            auto one_more = [&]() {
                if (!room_for_one_more()) return false;
                emgpu_ctx::TraceBlock b;
                if (!device_block(L.bytes, &b.p)) return false;   // (another process took the memory meanwhile)
                b.bytes = L.bytes;
                cands.push_back(b);
                return true;
            };
            while ((int)cands.size() < target && one_more()) {}
            if (cands.size() == 1) {
                t->rep.candidates = 1;   // no memory for a second candidate
            } else {
                Events ev(2);
                std::vector<emgpu_sample_out> outs(cands.size());
                for (size_t i = 0; i < cands.size(); i++) trace_bind(L, want, cands[i].p, &outs[i]);
                // the allocations above left the device idle and its clocks fell: load it first
                const auto t0 = Clock::now();
                while (ms_since(t0) < 500.0) (void)time_launches(ctx, h, p, &outs.back(), 0, 4, ev);
                std::vector<float> ms(cands.size(), 1e30f);
                for (int round = 0; round < 2; round++)   // a b c a b c: what is left of a ramp does not favour the last one
                    for (size_t i = 0; i < cands.size(); i++) ms[i] = std::min(ms[i], time_launches(ctx, h, p, &outs[i], 2, 5, ev));
                const size_t kept = (size_t)(std::min_element(ms.begin(), ms.end()) - ms.begin());
                t->rep.candidates = (int32_t)cands.size();
                t->rep.kept = (int32_t)kept;
                for (size_t i = 0; i < cands.size() && i < 8; i++) t->rep.ms[i] = ms[i];
                t->rep.first_allocation_ms = ms[0];
                t->rep.kept_ms = ms[kept];
                // the probe's launches may have left deferred per-trajectory bits (a rejection cap ...): the caller's own call will raise them again
                const int rc = emgpu_ctx_sync(ctx);
                if (rc == EMGPU_ERR_HIP) throw Error(rc, g_err);
                for (size_t i = 0; i < cands.size(); i++)
                    if (i != kept) device_release(cands[i].p);
                emgpu_ctx::TraceBlock k = cands[kept];
                k.probed = true;
                k.ms = ms[kept];
                cands.assign(1, k);
            }
        }
    } catch (...) {
        (void)hipStreamSynchronize(ctx->stream);
        give_up();
        throw;
    }
    t->blk = cands[0];
    trace_bind(L, want, t->blk.p, &t->out);
    *out = t.release();
    return EMGPU_OK;
    EMGPU_CATCH
}

int emgpu_trace_out(const emgpu_trace *t, emgpu_sample_out *out) {
    if (!t || !out) return fail(EMGPU_ERR_ARG, "null argument");
    *out = t->out;
    return EMGPU_OK;
}

int emgpu_trace_report(const emgpu_trace *t, emgpu_trace_report_t *out) {
    if (!t || !out) return fail(EMGPU_ERR_ARG, "null argument");
    *out = t->rep;
    return EMGPU_OK;
}

int emgpu_trace_free(emgpu_ctx *ctx, emgpu_trace *t) {
    EMGPU_TRY
    if (!t) return EMGPU_OK;
    if (!ctx) return fail(EMGPU_ERR_ARG, "null ctx");
    CTX_LOCK(ctx);
    HIP_OK(hipSetDevice(ctx->device));
    HIP_OK(hipStreamSynchronize(ctx->stream));   // nothing in flight may still write the block when somebody else takes it
    ctx->trace_pool.push_back(t->blk);
    delete t;
    return EMGPU_OK;
    EMGPU_CATCH
}

// Plain device memory from the same allocator as the traces (no probe): for outputs that are not a DBN trace -- the joined tracks of
// emgpu_sample_terminal_device, a consumer's own buffers.
int emgpu_device_alloc(emgpu_ctx *ctx, uint64_t bytes, void **out) {
    EMGPU_TRY
    if (!ctx || !out) return fail(EMGPU_ERR_ARG, "null argument");
    CTX_LOCK(ctx);
    HIP_OK(hipSetDevice(ctx->device));
    void *p = nullptr;
    const size_t need = std::max<size_t>((size_t)bytes, 256);
    if (!device_block(need, &p)) {
        HIP_OK(hipStreamSynchronize(ctx->stream));
        pool_release(ctx);
        if (!device_block(need, &p)) return fail(EMGPU_ERR_HIP, "emgpu_device_alloc: out of device memory (" + std::to_string(need) + " bytes)");
    }
    ctx->device_blocks.insert(p);
    *out = p;
    return EMGPU_OK;
    EMGPU_CATCH
}

int emgpu_device_free(emgpu_ctx *ctx, void *p) {
    EMGPU_TRY
    if (!p) return EMGPU_OK;
    if (!ctx) return fail(EMGPU_ERR_ARG, "null ctx");
    CTX_LOCK(ctx);
    if (!ctx->device_blocks.erase(p)) return fail(EMGPU_ERR_ARG, "emgpu_device_free: not a block of this ctx");
    HIP_OK(hipSetDevice(ctx->device));
    HIP_OK(hipStreamSynchronize(ctx->stream));
    device_release(p);
    return EMGPU_OK;
    EMGPU_CATCH
}

// ================================================================================================ the pinned pool
int emgpu_host_alloc(emgpu_ctx *ctx, uint64_t bytes, void **out) {
    EMGPU_TRY
    if (!ctx || !out) return fail(EMGPU_ERR_ARG, "null argument");
    CTX_LOCK(ctx);
    HIP_OK(hipSetDevice(ctx->device));
    const size_t need = std::max<size_t>((size_t)bytes, 64);   // (portable: emgpu_sample_dbn_multi_host hands one caller array to the contexts of several devices)
    emgpu_ctx::HostBlock *best = nullptr;
    for (auto &b : ctx->host_pool)
        if (!b.in_use && b.bytes >= need && b.bytes <= need + need / 2 + (1u << 20) && (!best || b.bytes < best->bytes)) best = &b;
    if (best) {
        best->in_use = true;
        *out = best->p;
        return EMGPU_OK;
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, need, hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        for (auto it = ctx->host_pool.begin(); it != ctx->host_pool.end();)   // the pool's idle blocks first, then once more
            if (!it->in_use) { (void)hipHostFree(it->p); it = ctx->host_pool.erase(it); } else ++it;
        HIP_OK(hipHostMalloc(&p, need, hipHostMallocPortable));
    }
    ctx->host_pool.push_back({p, need, true});
    *out = p;
    return EMGPU_OK;
    EMGPU_CATCH
}

int emgpu_host_free(emgpu_ctx *ctx, void *p) {
    if (!p) return EMGPU_OK;
    if (!ctx) return fail(EMGPU_ERR_ARG, "null ctx");
    CTX_LOCK(ctx);
    for (auto &b : ctx->host_pool)
        if (b.p == p) {
            if (!b.in_use) return fail(EMGPU_ERR_ARG, "emgpu_host_free: block freed twice");
            b.in_use = false;
            // the pool keeps at most 16 GiB of idle pinned memory (callers that wander through many sizes would pin the host's RAM away)
            size_t idle = 0;
            for (const auto &q : ctx->host_pool) idle += q.in_use ? 0 : q.bytes;
            if (idle > ((size_t)16 << 30)) {
                (void)hipSetDevice(ctx->device);
                for (auto it = ctx->host_pool.begin(); it != ctx->host_pool.end();)
                    if (!it->in_use && it->p != p) { (void)hipHostFree(it->p); it = ctx->host_pool.erase(it); } else ++it;
            }
            return EMGPU_OK;
        }
    return fail(EMGPU_ERR_ARG, "emgpu_host_free: not a block of this ctx");
}

int emgpu_host_stats(const emgpu_ctx *ctx, emgpu_host_stats_t *out) {
    if (!ctx || !out) return fail(EMGPU_ERR_ARG, "null argument");
    *out = ctx->host_stats;
    return EMGPU_OK;
}

// ================================================================================================ the host path
int emgpu_sample_dbn_host(emgpu_ctx *ctx, const emgpu_model *h, const emgpu_sample_params *p, const emgpu_sample_out *out) {
    EMGPU_TRY
    if (!ctx || !h || !p || !out) return fail(EMGPU_ERR_ARG, "null argument");
    CTX_LOCK(ctx);
    HIP_OK(hipSetDevice(ctx->device));
    const auto t_call = Clock::now();
    const Model &m = h->m;
    const size_t n = (size_t)(p->n > 0 ? p->n : 0), ni = (size_t)m.n_initial, nd = (size_t)m.n_dyn();
    const size_t G4 = ((size_t)(p->sample_time > 0 ? p->sample_time : 0) + 3) / 4;
    // the host arrays may be dimensioned for a larger batch (ld) of which this call fills columns [off, off + n)
    const size_t ld = out->ld ? (size_t)out->ld : n, off = (size_t)out->col_offset;
    if (out->ld < 0 || out->col_offset < 0 || off + n > ld) return fail(EMGPU_ERR_ARG, "col_offset + n exceeds ld");
    if ((out->ev_count != nullptr) != (out->events != nullptr)) return fail(EMGPU_ERR_ARG, "ev_count and events go together");
    if (out->events && p->event_cap < 1) return fail(EMGPU_ERR_ARG, "event_cap must be >= 1");
    const size_t cap = out->events ? (size_t)p->event_cap : 0;
    emgpu_host_stats_t st{};
    if (n == 0) {   // nothing to draw: the arguments are still checked like any call's
        emgpu_sample_out d{};
        const int rc0 = emgpu_sample_dbn_device(ctx, h, p, &d);
        ctx->host_stats = st;
        return rc0 == EMGPU_OK ? emgpu_ctx_sync(ctx) : rc0;
    }

    // ---- one chunk on the device: [small arrays | large arrays | event lists | packed rows | pack scratch]; what is staged is a prefix
    struct Arr { void *dst; size_t rows, elem, dev_off; bool offset_by_col; };
    std::vector<Arr> small, large;
    const bool direct = (out->init_bin || out->init_val || out->dyn_bin || out->dyn_val) &&
                        (!out->init_bin || is_pinned(out->init_bin)) && (!out->init_val || is_pinned(out->init_val)) &&
                        (!out->dyn_bin || is_pinned(out->dyn_bin)) && (!out->dyn_val || is_pinned(out->dyn_val));
    size_t bpt = 0;   // device bytes per trajectory
    bpt += (out->ev_count ? 4 : 0) + (out->attempts ? 4 : 0) + (out->log_weight ? 8 : 0);
    bpt += (out->init_bin ? ni : 0) + (out->init_val ? 4 * ni : 0);
    bpt += (out->dyn_bin ? 4 * G4 * nd : 0) + (out->dyn_val ? 16 * G4 * nd : 0);
    bpt += 16 * cap;   // the lists and their packed copy
    // pinned outputs: one pitched copy per array (56.5 GB/s of the 57 GB/s a plain pinned copy reaches; row-by-row linear copies: 46 -- tools/host_path_probe.py)
    static const bool direct_rows = [] { const char *e = getenv("EMGPU_HOST_DIRECT"); return e && !strcmp(e, "rows"); }();
    size_t target = (size_t)(direct ? 1024 : 256) << 20;   // pinned outputs: larger pieces (the copy engine writes row by row into the caller's pitch)
    if (const char *e = getenv("EMGPU_HOST_CHUNK_MB")) { const long v = atol(e); if (v > 0) target = (size_t)v << 20; }
    size_t C = std::max<size_t>(1024, target / std::max<size_t>(bpt, 1) / 1024 * 1024);
    if (cap) C = std::min(C, std::max<size_t>(1024, ((size_t)0xFFFF0000u / cap) / 1024 * 1024));   // a chunk's packed rows are counted in 32 bits
    if (C >= n) C = n;
    else {   // chunks of equal size: the last one is not a sliver (and a staged copy moves whole chunk buffers)
        const size_t k = (n + C - 1) / C;
        C = std::min(C, round_up((n + k - 1) / k, 1024));
    }
    const size_t Cp = round_up(C, 256), nchunks = (n + C - 1) / C;
    size_t o = 0;
    auto put = [&](size_t bytes) { const size_t at = o; o = round_up(o + bytes, 256); return at; };
    if (out->ev_count) small.push_back({out->ev_count, 1, 4, put(Cp * 4), true});
    if (out->attempts) small.push_back({out->attempts, 1, 4, put(Cp * 4), true});
    if (out->log_weight) small.push_back({out->log_weight, 1, 8, put(Cp * 8), false});
    const size_t small_bytes = o;
    if (out->init_bin) large.push_back({out->init_bin, ni, 1, put(ni * Cp), true});
    if (out->init_val) large.push_back({out->init_val, ni, 4, put(ni * Cp * 4), true});
    if (out->dyn_bin) large.push_back({out->dyn_bin, G4 * nd, 4, put(G4 * nd * Cp * 4), true});
    if (out->dyn_val) large.push_back({out->dyn_val, G4 * nd, 16, put(G4 * nd * Cp * 16), true});
    const size_t stage_prefix = direct ? small_bytes : o;
    const size_t o_ev = cap ? put(Cp * cap * 8) : 0, o_packed = cap ? put(Cp * cap * 8) : 0;
    const size_t o_scratch = cap ? put(emgpu::pack_scratch_words((int64_t)Cp) * 4) : 0;
    const size_t dev_bytes = std::max<size_t>(o, 256);
    const size_t stage_bytes = std::max<size_t>(stage_prefix + (cap ? C * cap * 8 : 0), 256);

    // the chunk buffers are blocks of the trace pool's allocator (nothing is probed: these launches are microseconds beside their copies)
    for (size_t q = 0; q < (nchunks == 1 ? 1u : 2u); q++) {
        emgpu_ctx::TraceBlock &b = ctx->chunk_buf[q];
        if (b.bytes >= dev_bytes) continue;
        HIP_OK(hipStreamSynchronize(ctx->stream));
        if (b.p) { device_release(b.p); b = emgpu_ctx::TraceBlock(); }
        // (plain hipMalloc blocks: these buffers are the SOURCE of copies, which is all their placement could matter for -- measured: it does not)
        b = pool_take(ctx, dev_bytes + dev_bytes / 8, nullptr, /*plain=*/true);   // (some headroom: batch sizes that wobble do not reallocate)
        if (!b.p) return fail(EMGPU_ERR_HIP, "emgpu_sample_dbn_host: out of device memory");
    }
    if (ctx->h_stage_cap < stage_bytes) {
        const size_t want_cap = stage_bytes + stage_bytes / 8;
        for (auto &s : ctx->h_stage) { if (s) HIP_OK(hipHostFree(s)); s = nullptr; }
        ctx->h_stage_cap = 0;
        for (size_t b = 0; b < (nchunks == 1 ? 1u : 2u); b++) HIP_OK(hipHostMalloc(&ctx->h_stage[b], want_cap, hipHostMallocDefault));
        ctx->h_stage_cap = want_cap;
    }
    if (nchunks > 1 && !ctx->h_stage[1]) HIP_OK(hipHostMalloc(&ctx->h_stage[1], ctx->h_stage_cap, hipHostMallocDefault));
    if (!ctx->copy_stream) HIP_OK(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    if (!ctx->h_total) HIP_OK(hipHostMalloc((void **)&ctx->h_total, 2 * sizeof(uint64_t), hipHostMallocDefault));

    emgpu_sample_params pd = *p;
    if (p->start) {     // the start grid and the index list are caller (host) memory here: uploaded once, every chunk reads its rows
        int32_t *ds = (int32_t *)ctx_scratch(ctx, 0, n * ni * sizeof(int32_t));
        HIP_OK(hipMemcpyAsync(ds, p->start, n * ni * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        pd.start = ds;
    }
    if (p->indices) {
        uint64_t *di = (uint64_t *)ctx_scratch(ctx, 1, n * sizeof(uint64_t));
        HIP_OK(hipMemcpyAsync(di, p->indices, n * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
        pd.indices = di;
    }
    if (p->start || p->indices) HIP_OK(hipStreamSynchronize(ctx->stream));

    Events ev(8);   // per buffer b: 4b + {kernel start, kernel end, copy start, copy end}
    const int T = host_threads();
    int rc = EMGPU_OK;
    size_t packed_rows[2] = {0, 0};
    st.chunks = (int32_t)nchunks; st.chunk_n = (int32_t)C; st.threads = T; st.direct = direct ? 1 : 0;

    auto launch_chunk = [&](size_t k) {
        const int b = (int)(k & 1);
        const size_t k0 = k * C, c = std::min(C, n - k0);
        char *dev = (char *)ctx->chunk_buf[b].p;
        emgpu_sample_params q = pd;
        q.n = (int64_t)c;
        q.first_index = p->first_index + (uint64_t)k0;
        if (pd.indices) q.indices = pd.indices + k0;
        if (pd.start) q.start = pd.start + k0 * ni;
        emgpu_sample_out d{};
        d.ld = (int64_t)Cp;
        for (const Arr &a : small) {
            if (a.dst == out->ev_count) d.ev_count = (uint32_t *)(dev + a.dev_off);
            else if (a.dst == out->attempts) d.attempts = (int32_t *)(dev + a.dev_off);
            else d.log_weight = (double *)(dev + a.dev_off);
        }
        for (const Arr &a : large) {
            if (a.dst == out->init_bin) d.init_bin = (uint8_t *)(dev + a.dev_off);
            else if (a.dst == out->init_val) d.init_val = (float *)(dev + a.dev_off);
            else if (a.dst == out->dyn_bin) d.dyn_bin = (uint32_t *)(dev + a.dev_off);
            else d.dyn_val = (float *)(dev + a.dev_off);
        }
        if (cap) d.events = (emgpu_event *)(dev + o_ev);
        HIP_OK(hipEventRecord(ev[4 * b], ctx->stream));
        const int r = emgpu_sample_dbn_device(ctx, h, &q, &d);
        if (r != EMGPU_OK) throw Error(r, g_err);
        if (cap) {
            hipError_t e = emgpu::launch_pack_events((int64_t)c, (uint32_t)cap, d.ev_count, (const uint64_t *)(dev + o_ev), (uint32_t *)(dev + o_scratch),
                                                     (uint64_t *)(dev + o_packed), ctx->stream);
            if (e != hipSuccess) throw Error(EMGPU_ERR_HIP, std::string("pack launch: ") + hipGetErrorString(e));
            HIP_OK(hipMemcpyAsync(&ctx->h_total[b], dev + o_scratch, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
        }
        HIP_OK(hipEventRecord(ev[4 * b + 1], ctx->stream));
        if (cap) {   // how many rows cross PCIe is known only now (the launch stream holds nothing but this chunk)
            HIP_OK(hipEventSynchronize(ev[4 * b + 1]));
            packed_rows[b] = (size_t)ctx->h_total[b];
        }
        // ---- the copy: behind the kernel, on the copy stream
        HIP_OK(hipStreamWaitEvent(ctx->copy_stream, ev[4 * b + 1], 0));
        HIP_OK(hipEventRecord(ev[4 * b + 2], ctx->copy_stream));
        char *stg = (char *)ctx->h_stage[b];
        if (stage_prefix) HIP_OK(hipMemcpyAsync(stg, dev, stage_prefix, hipMemcpyDeviceToHost, ctx->copy_stream));
        st.bytes_d2h += (int64_t)stage_prefix;
        if (cap && packed_rows[b]) {
            HIP_OK(hipMemcpyAsync(stg + stage_prefix, dev + o_packed, packed_rows[b] * 8, hipMemcpyDeviceToHost, ctx->copy_stream));
            st.bytes_d2h += (int64_t)packed_rows[b] * 8;
            st.event_rows += (int64_t)packed_rows[b];
        }
        if (direct)
            for (const Arr &a : large) {
                char *dst = (char *)a.dst + (off + k0) * a.elem;
                if (ld == c && Cp == c) HIP_OK(hipMemcpyAsync(dst, dev + a.dev_off, a.rows * c * a.elem, hipMemcpyDeviceToHost, ctx->copy_stream));
                else if (direct_rows)   // row by row: linear copies (the copy engine's fastest form), 1-16 MB each
                    for (size_t r = 0; r < a.rows; r++)
                        HIP_OK(hipMemcpyAsync(dst + r * ld * a.elem, dev + a.dev_off + r * Cp * a.elem, c * a.elem, hipMemcpyDeviceToHost, ctx->copy_stream));
                else HIP_OK(hipMemcpy2DAsync(dst, ld * a.elem, dev + a.dev_off, Cp * a.elem, c * a.elem, a.rows, hipMemcpyDeviceToHost, ctx->copy_stream));
                st.bytes_d2h += (int64_t)(a.rows * c * a.elem);
            }
        HIP_OK(hipEventRecord(ev[4 * b + 3], ctx->copy_stream));
    };

    auto drain_chunk = [&](size_t k) {
        const int b = (int)(k & 1);
        const size_t k0 = k * C, c = std::min(C, n - k0);
        HIP_OK(hipEventSynchronize(ev[4 * b + 3]));
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, ev[4 * b], ev[4 * b + 1])); st.kernel_ms += ms;
        HIP_OK(hipEventElapsedTime(&ms, ev[4 * b + 2], ev[4 * b + 3])); st.d2h_ms += ms;
        const auto t0 = Clock::now();
        const char *stg = (const char *)ctx->h_stage[b];
        struct Job { char *dst; const char *src; size_t bytes; };
        std::vector<Job> jobs;
        for (const Arr &a : small) jobs.push_back({(char *)a.dst + ((a.offset_by_col ? off : 0) + k0) * a.elem, stg + a.dev_off, c * a.elem});
        if (!direct)
            for (const Arr &a : large)
                for (size_t r = 0; r < a.rows; r++) jobs.push_back({(char *)a.dst + (r * ld + off + k0) * a.elem, stg + a.dev_off + r * Cp * a.elem, c * a.elem});
        std::vector<uint32_t> offs;
        const uint32_t *cnt = nullptr;
        if (cap) {   // the lists' first packed rows: a prefix sum over the chunk's counts (what the device did, redone on 4 c bytes)
            cnt = (const uint32_t *)(stg + small[0].dev_off);
            offs.resize(c + 1);
            uint32_t run = 0;
            for (size_t i = 0; i < c; i++) { offs[i] = run; run += std::min<uint32_t>(cnt[i], (uint32_t)cap); }
            offs[c] = run;
            if ((size_t)run != packed_rows[b]) throw Error(EMGPU_ERR_HIP, "emgpu_sample_dbn_host: packed event rows disagree with the counts");
        }
        size_t moved = cap ? packed_rows[b] * 8 : 0;
        for (const Job &j : jobs) moved += j.bytes;
        const int TT = (int)std::min<size_t>((size_t)T, std::max<size_t>(1, moved >> 20));   // a thread per MiB, at most T
        run_parallel(TT, [&](int t) {
            for (size_t j = (size_t)t; j < jobs.size(); j += (size_t)TT) memcpy(jobs[j].dst, jobs[j].src, jobs[j].bytes);
            if (cap) {
                const uint64_t *packed = (const uint64_t *)(stg + stage_prefix);
                emgpu_event *hev = out->events + (off + k0) * cap;
                const size_t i0 = c * (size_t)t / (size_t)TT, i1 = c * ((size_t)t + 1) / (size_t)TT;
                for (size_t i = i0; i < i1; i++) memcpy(hev + i * cap, packed + offs[i], (size_t)(offs[i + 1] - offs[i]) * 8);
            }
        });
        st.scatter_ms += ms_since(t0);
    };

    try {
        for (size_t k = 0; k <= nchunks; k++) {
            if (k < nchunks) launch_chunk(k);
            if (k > 0) drain_chunk(k - 1);
        }
        rc = emgpu_ctx_sync(ctx);   // deferred per-trajectory errors of every chunk (rejection cap, event cap, presets)
    } catch (...) {
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
        throw;
    }
    st.total_ms = ms_since(t_call);
    ctx->host_stats = st;
    return rc;
    EMGPU_CATCH
}

} // extern "C"
