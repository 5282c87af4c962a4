// emgpu_capi.cpp -- the C ABI declared in include/emgpu.h.
// No CPU fallback anywhere: without a HIP device emgpu_ctx_create fails with EMGPU_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/emgpu.h"
#include "emgpu_launch.h"
#include "emgpu_model.hpp"

#include "emgpu_internal.hpp"

namespace emgpu_detail {
std::string &last_error() {
    thread_local std::string e;
    return e;
}
int fail(int code, const std::string &msg) {
    last_error() = msg;
    return code;
}
} // namespace emgpu_detail

template <typename T, typename V>
static int64_t copy_out(const V &v, T *out, int64_t cap) {
    if (out) {
        if ((int64_t)v.size() > cap) return fail(EMGPU_ERR_ARG, "output buffer too small");
        for (size_t i = 0; i < v.size(); i++) out[i] = (T)v[i];
    }
    return (int64_t)v.size();
}

extern "C" {

const char *emgpu_last_error(void) { return g_err.c_str(); }
#ifndef EMGPU_SRC_HASH
#define EMGPU_SRC_HASH "unhashed"
#endif
#define EMGPU_STR2(x) #x
#define EMGPU_STR(x) EMGPU_STR2(x)
const char *emgpu_version(void) { return "emgpu 0.4 (gfx950) philox4x32-" EMGPU_STR(EMGPU_PHILOX_ROUNDS) " src:" EMGPU_SRC_HASH; }
int32_t emgpu_philox_rounds(void) { return EMGPU_PHILOX_ROUNDS; }
int32_t emgpu_slot_map_revision(void) { return 2; }

int emgpu_model_load_txt(const char *path, const int32_t *idx_zero_boundaries, int32_t n_idx,
                         int32_t is_overwrite_zero_boundaries, emgpu_model **out) {
    EMGPU_TRY
    if (!path || !out) return fail(EMGPU_ERR_ARG, "null argument");
    std::unique_ptr<Model> m(emgpu::load_txt(path, idx_zero_boundaries, n_idx, is_overwrite_zero_boundaries != 0));
    *out = new emgpu_model{std::move(*m)};
    return EMGPU_OK;
    EMGPU_CATCH
}

/* em_read.m:47-107 once, then the binary cache */
int emgpu_model_save_bin(const emgpu_model *m, const char *path) {
    EMGPU_TRY
    if (!m || !path) return fail(EMGPU_ERR_ARG, "null argument");
    emgpu::save_bin(m->m, path, EMGPU_SRC_HASH);
    return EMGPU_OK;
    EMGPU_CATCH
}
int emgpu_model_load_bin(const char *path, emgpu_model **out) {
    EMGPU_TRY
    if (!path || !out) return fail(EMGPU_ERR_ARG, "null argument");
    std::unique_ptr<Model> m(emgpu::load_bin(path, EMGPU_SRC_HASH));
    emgpu_model *h = new emgpu_model{std::move(*m)};
    *out = h;
    return EMGPU_OK;
    EMGPU_CATCH
}

static std::vector<std::string> split_nl(const char *s) {
    std::vector<std::string> out;
    if (!s) return out;
    std::string cur;
    for (const char *p = s; *p; p++) {
        if (*p == '\n') { out.push_back(cur); cur.clear(); }
        else cur.push_back(*p);
    }
    out.push_back(cur);
    return out;
}

int emgpu_model_from_arrays(const emgpu_model_desc *d, emgpu_model **out) {
    EMGPU_TRY
    if (!d || !out || d->n_initial <= 0 || !d->G_initial || !d->r_initial || !d->N_initial) return fail(EMGPU_ERR_ARG, "null/empty argument");
    if (d->n_initial > EMGPU_MAX_NI) return fail(EMGPU_ERR_UNSUPPORTED, "more initial variables than EMGPU_MAX_NI");
    for (int i = 0; i < d->n_initial; i++)
        if (d->r_initial[i] < 1 || d->r_initial[i] > EMGPU_MAX_R) return fail(EMGPU_ERR_ARG, "r_initial entry outside 1..EMGPU_MAX_R");
    if (d->n_transition > 0) {
        if (d->n_transition < d->n_initial || !d->r_transition) return fail(EMGPU_ERR_ARG, "transition arrays missing");
        for (int i = 0; i < d->n_transition; i++)
            if (d->r_transition[i] < 1 || d->r_transition[i] > EMGPU_MAX_R) return fail(EMGPU_ERR_ARG, "r_transition entry outside 1..EMGPU_MAX_R");
        if (d->n_dyn < 0 || d->n_dyn > d->n_transition - d->n_initial || (d->n_dyn > 0 && !d->temporal_map))
            return fail(EMGPU_ERR_ARG, "n_dyn must be within 0..n_transition-n_initial (with a temporal_map)");
        for (int k = 0; k < d->n_dyn; k++)
            if (d->temporal_map[2 * k] < 1 || d->temporal_map[2 * k] > d->n_initial || d->temporal_map[2 * k + 1] <= d->n_initial ||
                d->temporal_map[2 * k + 1] > d->n_transition)
                return fail(EMGPU_ERR_ARG, "temporal_map row outside (1..n_initial, n_initial+1..n_transition)");
    }
    if (d->boundaries && d->bnd_len)
        for (int i = 0; i < d->n_initial; i++)
            if (d->bnd_len[i] != 0 && d->bnd_len[i] != d->r_initial[i] + 1)
                return fail(EMGPU_ERR_ARG, "bnd_len[i] must be 0 ('*') or r_initial[i] + 1");
    if (d->zero_bins)
        for (int i = 0; i < d->n_initial; i++)
            if (d->zero_bins[i] < 0 || d->zero_bins[i] > d->r_initial[i]) return fail(EMGPU_ERR_ARG, "zero_bins entry outside 0..r");
    std::unique_ptr<emgpu_model> h(new emgpu_model());
    Model &m = h->m;
    const int ni = d->n_initial, nt = d->n_transition;
    m.n_initial = ni;
    m.n_transition = nt;
    m.G_initial.assign(d->G_initial, d->G_initial + (size_t)ni * ni);
    m.r_initial.assign(d->r_initial, d->r_initial + ni);
    m.labels_initial = split_nl(d->labels_initial);
    m.labels_transition = split_nl(d->labels_transition);
    auto q_of = [](const std::vector<uint8_t> &G, int n, const std::vector<int> &r, int c) {
        int64_t q = 1;
        for (int p = 0; p < n; p++) if (G[(size_t)p * n + c]) q *= r[p];
        return q;
    };
    m.N_initial.resize(ni);
    int64_t idx = 0;
    for (int i = 0; i < ni; i++) {
        int64_t cnt = q_of(m.G_initial, ni, m.r_initial, i) * m.r_initial[i];
        if (idx + cnt > d->n_N_initial) return fail(EMGPU_ERR_ARG, "N_initial too short");
        m.N_initial[i].assign(d->N_initial + idx, d->N_initial + idx + cnt);
        idx += cnt;
    }
    if (idx != d->n_N_initial) return fail(EMGPU_ERR_ARG, "N_initial length mismatch");
    if (nt > 0) {
        if (!d->G_transition || !d->r_transition || !d->N_transition || nt < ni) return fail(EMGPU_ERR_ARG, "transition arrays missing");
        m.G_transition.assign(d->G_transition, d->G_transition + (size_t)nt * nt);
        m.r_transition.assign(d->r_transition, d->r_transition + nt);
        m.N_transition.resize(nt);
        idx = 0;
        for (int i = ni; i < nt; i++) {
            int64_t cnt = q_of(m.G_transition, nt, m.r_transition, i) * m.r_transition[i];
            if (idx + cnt > d->n_N_transition) return fail(EMGPU_ERR_ARG, "N_transition too short");
            m.N_transition[i].assign(d->N_transition + idx, d->N_transition + idx + cnt);
            idx += cnt;
        }
        if (idx != d->n_N_transition) return fail(EMGPU_ERR_ARG, "N_transition length mismatch");
        if (d->temporal_map)
            for (int k = 0; k < d->n_dyn; k++) m.temporal_map.push_back({d->temporal_map[2 * k], d->temporal_map[2 * k + 1]});
    }
    m.boundaries.assign(ni, {});
    if (d->boundaries && d->bnd_len) {
        int64_t o = 0;
        for (int i = 0; i < ni; i++) {
            m.boundaries[i].assign(d->boundaries + o, d->boundaries + o + d->bnd_len[i]);
            o += d->bnd_len[i];
        }
    }
    if (d->zero_bins) m.zero_bins.assign(d->zero_bins, d->zero_bins + ni);
    if (d->resample_rates) m.resample_rates.assign(d->resample_rates, d->resample_rates + ni);
    m.finalize();
    *out = h.release();
    return EMGPU_OK;
    EMGPU_CATCH
}

void emgpu_model_free(emgpu_model *m) { delete m; }

int emgpu_model_info(const emgpu_model *h, emgpu_model_info_t *out) {
    if (!h || !out) return fail(EMGPU_ERR_ARG, "null argument");
    const Model &m = h->m;
    memset(out, 0, sizeof *out);
    out->n_initial = m.n_initial;
    out->n_transition = m.n_transition;
    out->n_dyn = m.n_dyn();
    out->is_dynvar_depend = (m.n_transition > 0 && m.n_dyn() > 0) ? (int)m.is_dynvar_depend() : 0;
    for (auto &v : m.N_initial) out->n_N_initial += (int64_t)v.size();
    for (auto &v : m.N_transition) out->n_N_transition += (int64_t)v.size();
    for (int r : m.r_initial) out->max_r = r > out->max_r ? r : out->max_r;
    for (int r : m.r_transition) out->max_r = r > out->max_r ? r : out->max_r;
    for (double r : m.resample_rates) out->n_resample_active += r > 0.0;
    return EMGPU_OK;
}

int64_t emgpu_model_get_i32(const emgpu_model *h, int32_t field, int32_t *out, int64_t cap) {
    if (!h) return fail(EMGPU_ERR_ARG, "null model");
    const Model &m = h->m;
    switch (field) {
    case EMGPU_F_R_INITIAL: return copy_out(m.r_initial, out, cap);
    case EMGPU_F_R_TRANSITION: return copy_out(m.r_transition, out, cap);
    case EMGPU_F_ORDER_INITIAL: return copy_out(m.order_initial, out, cap);
    case EMGPU_F_ORDER_TRANSITION: return copy_out(m.order_transition, out, cap);
    case EMGPU_F_ZERO_BINS: return copy_out(m.zero_bins, out, cap);
    case EMGPU_F_START: return copy_out(m.start, out, cap);
    case EMGPU_F_G_INITIAL: return copy_out(m.G_initial, out, cap);
    case EMGPU_F_G_TRANSITION: return copy_out(m.G_transition, out, cap);
    case EMGPU_F_TEMPORAL_MAP: {
        std::vector<int> f;
        for (auto &t : m.temporal_map) { f.push_back(t[0]); f.push_back(t[1]); }
        return copy_out(f, out, cap);
    }
    default: return fail(EMGPU_ERR_ARG, "unknown i32 field");
    }
}

int64_t emgpu_model_get_f64(const emgpu_model *h, int32_t field, int32_t node, double *out, int64_t cap) {
    if (!h) return fail(EMGPU_ERR_ARG, "null model");
    const Model &m = h->m;
    auto in_range = [&](size_t n) { return node >= 1 && (size_t)node <= n; };
    switch (field) {
    case EMGPU_F_N_INITIAL: if (!in_range(m.N_initial.size())) break; return copy_out(m.N_initial[node - 1], out, cap);
    case EMGPU_F_ALPHA_INITIAL: if (!in_range(m.A_initial.size())) break; return copy_out(m.A_initial[node - 1], out, cap);
    case EMGPU_F_N_TRANSITION: if (!in_range(m.N_transition.size())) break; return copy_out(m.N_transition[node - 1], out, cap);
    case EMGPU_F_ALPHA_TRANSITION: if (!in_range(m.A_transition.size())) break; return copy_out(m.A_transition[node - 1], out, cap);
    case EMGPU_F_BOUNDARIES: if (!in_range(m.boundaries.size())) break; return copy_out(m.boundaries[node - 1], out, cap);
    case EMGPU_F_RESAMPLE_RATES: return copy_out(m.resample_rates, out, cap);
    default: return fail(EMGPU_ERR_ARG, "unknown f64 field");
    }
    return fail(EMGPU_ERR_ARG, "node out of range");
}

int64_t emgpu_model_get_text(const emgpu_model *h, int32_t field, char *out, int64_t cap) {
    if (!h) return fail(EMGPU_ERR_ARG, "null model");
    const std::vector<std::string> *v = field == EMGPU_F_LABELS_INITIAL ? &h->m.labels_initial
                                      : field == EMGPU_F_LABELS_TRANSITION ? &h->m.labels_transition : nullptr;
    if (!v) return fail(EMGPU_ERR_ARG, "unknown text field");
    std::string s;
    for (size_t i = 0; i < v->size(); i++) { if (i) s.push_back('\n'); s += (*v)[i]; }
    if (out) {
        if ((int64_t)s.size() + 1 > cap) return fail(EMGPU_ERR_ARG, "output buffer too small");
        memcpy(out, s.c_str(), s.size() + 1);
    }
    return (int64_t)s.size() + 1;
}

int emgpu_model_set_f64(emgpu_model *h, int32_t field, int32_t node, const double *v, int64_t n) {
    EMGPU_TRY
    if (!h || (!v && n != 0) || n < 0) return fail(EMGPU_ERR_ARG, "null argument or negative length");
    Model &m = h->m;
    auto assign_same = [&](std::vector<std::vector<double>> &dst) {
        if (node < 1 || (size_t)node > dst.size()) throw Error(EMGPU_ERR_ARG, "node out of range");
        if ((int64_t)dst[node - 1].size() != n) throw Error(EMGPU_ERR_ARG, "size mismatch: tables keep their r x q shape");
        dst[node - 1].assign(v, v + n);
    };
    switch (field) {
    case EMGPU_F_N_INITIAL: assign_same(m.N_initial); break;
    case EMGPU_F_N_TRANSITION: assign_same(m.N_transition); break;
    case EMGPU_F_ALPHA_INITIAL: assign_same(m.A_initial); break;
    case EMGPU_F_ALPHA_TRANSITION: assign_same(m.A_transition); break;
    case EMGPU_F_BOUNDARIES:
        if (node < 1 || node > m.n_initial) throw Error(EMGPU_ERR_ARG, "node out of range");
        // numel(boundaries) == r + 1 in every shipped file (dediscretize.m:33-38 reads params(d+1)); 0 = categorical ('*')
        if (n != 0 && n != (int64_t)m.r_initial[node - 1] + 1) throw Error(EMGPU_ERR_ARG, "boundaries need 0 or r+1 entries");
        // (zero_bins is its own property in the reference, derived once by em_read.m:110-114 and not by
        // set.boundaries: use emgpu_model_set_zero_bins to change it)
        m.boundaries[node - 1].assign(v, v + n);
        break;
    case EMGPU_F_RESAMPLE_RATES:
        if (n != m.n_initial) throw Error(EMGPU_ERR_ARG, "resample_rates needs n_initial entries");
        m.resample_rates.assign(v, v + n);
        break;
    default: throw Error(EMGPU_ERR_ARG, "field is not settable");
    }
    m.version++;
    return EMGPU_OK;
    EMGPU_CATCH
}

int emgpu_model_set_prior(emgpu_model *h, int32_t kind, double value) {
    EMGPU_TRY
    if (!h) return fail(EMGPU_ERR_ARG, "null model");
    if (kind != 0 && kind != 1) return fail(EMGPU_ERR_PRIOR, "Unknown prior, if char expecting prior = 'dbe'");
    h->m.set_prior(kind, value);
    return EMGPU_OK;
    EMGPU_CATCH
}

int emgpu_model_set_transition_stay_prior(emgpu_model *h, double prior) {
    EMGPU_TRY
    if (!h) return fail(EMGPU_ERR_ARG, "null model");
    h->m.set_transition_stay_prior(prior);
    return EMGPU_OK;
    EMGPU_CATCH
}

int emgpu_model_set_start(emgpu_model *h, const int32_t *start, int32_t n) {
    if (!h || !start || n != h->m.n_initial) return fail(EMGPU_ERR_ARG, "start needs n_initial entries");
    for (int i = 0; i < n; i++)
        if (start[i] < 0 || start[i] > h->m.r_initial[i]) return fail(EMGPU_ERR_ARG, "start bin out of range");
    h->m.start.assign(start, start + n);
    h->m.version++;
    return EMGPU_OK;
}

int emgpu_model_start_log_weight(const emgpu_model *h, double *out) {
    EMGPU_TRY
    if (!h || !out) return fail(EMGPU_ERR_ARG, "null argument");
    *out = h->m.start_log_weight();
    return EMGPU_OK;
    EMGPU_CATCH
}

int emgpu_model_set_zero_bins(emgpu_model *h, const int32_t *zero_bins, int32_t n) {
    if (!h || !zero_bins || n != h->m.n_initial) return fail(EMGPU_ERR_ARG, "zero_bins needs n_initial entries");
    for (int i = 0; i < n; i++)
        if (zero_bins[i] < 0 || zero_bins[i] > h->m.r_initial[i]) return fail(EMGPU_ERR_ARG, "zero bin out of range");
    h->m.zero_bins.assign(zero_bins, zero_bins + n);
    h->m.version++;
    return EMGPU_OK;
}

// ------------------------------------------------------------------------------------------------
int emgpu_ctx_create(int32_t device, emgpu_ctx **out) {
    EMGPU_TRY
    if (!out) return fail(EMGPU_ERR_ARG, "null argument");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(EMGPU_ERR_NO_DEVICE, "no HIP device visible: libemgpu has no CPU path");
    if (device < 0 || device >= count) return fail(EMGPU_ERR_ARG, "device index out of range");
    std::unique_ptr<emgpu_ctx> c(new emgpu_ctx());
    c->device = device;
    HIP_OK(hipSetDevice(device));
    HIP_OK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    HIP_OK(hipMalloc((void **)&c->d_status, sizeof(uint32_t)));
    HIP_OK(hipMemset(c->d_status, 0, sizeof(uint32_t)));
    HIP_OK(hipMalloc((void **)&c->d_queue, sizeof(uint32_t)));
    HIP_OK(hipHostMalloc((void **)&c->h_status, sizeof(uint32_t), hipHostMallocDefault));
    *c->h_status = 0;
    *out = c.release();
    return EMGPU_OK;
    EMGPU_CATCH
}

int emgpu_ctx_set_stream(emgpu_ctx *ctx, void *hip_stream) {
    if (!ctx) return fail(EMGPU_ERR_ARG, "null ctx");
    CTX_LOCK(ctx);
    ctx->stream = (hipStream_t)hip_stream; // NULL = the HIP default (null) stream
    return EMGPU_OK;
}

int emgpu_ctx_sync(emgpu_ctx *ctx) {
    EMGPU_TRY
    if (!ctx) return fail(EMGPU_ERR_ARG, "null ctx");
    CTX_LOCK(ctx);
    HIP_OK(hipSetDevice(ctx->device));
    HIP_OK(hipMemcpyAsync(ctx->h_status, ctx->d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_OK(hipMemsetAsync(ctx->d_status, 0, sizeof(uint32_t), ctx->stream));
    HIP_OK(hipStreamSynchronize(ctx->stream));
    const uint32_t st = *ctx->h_status;
    if (st & 1u) return fail(EMGPU_ERR_REJECT_CAP, "rejection loop reached max_attempts for at least one trajectory");
    if (st & 4u) return fail(EMGPU_ERR_PRESET, "Attempt to preset a dependent variable (a row of the start grid presets a node without its parents, or a bin outside 1..r)");
    if (st & 2u) return fail(EMGPU_ERR_EVENT_CAP, "an event list did not fit event_cap rows");
    return EMGPU_OK;
    EMGPU_CATCH
}

int emgpu_ctx_trim(emgpu_ctx *ctx) {
    EMGPU_TRY
    if (!ctx) return fail(EMGPU_ERR_ARG, "null ctx");
    CTX_LOCK(ctx);
    HIP_OK(hipSetDevice(ctx->device));
    HIP_OK(hipStreamSynchronize(ctx->stream));
    for (auto &sc : ctx->scratch) { if (sc.p) HIP_OK(hipFree(sc.p)); sc.p = nullptr; sc.cap = 0; }
    ctx_release_host_side(ctx, false);
    return EMGPU_OK;
    EMGPU_CATCH
}

void emgpu_ctx_free(emgpu_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (auto &kv : ctx->cache) kv.second.free_tables();
    for (auto &sc : ctx->scratch) (void)hipFree(sc.p);
    ctx_release_host_side(ctx, true);
    (void)hipFree(ctx->d_status);
    (void)hipFree(ctx->d_queue);
    (void)hipFree(ctx->d_presets);
    (void)hipFree(ctx->d_layers);
    (void)hipFree(ctx->d_thr_base);
    (void)hipHostFree(ctx->h_status);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    for (int q = 0; q < emgpu_ctx::kSide; q++) {
        if (ctx->side[q]) (void)hipStreamDestroy(ctx->side[q]);
        if (ctx->ev_join[q]) (void)hipEventDestroy(ctx->ev_join[q]);
    }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    delete ctx;
}

const char *emgpu_last_kernel_name(const emgpu_ctx *ctx) { return ctx ? ctx->last_kernel.c_str() : ""; }
int32_t emgpu_last_launch_count(const emgpu_ctx *ctx) { return ctx ? ctx->last_launches : 0; }

} // extern "C"

// ------------------------------------------------------------------------------------------------
// `pinned`: uids of the other models of the current call -- their tables may already sit in a launch's
// argument list (terminal propagation, mixed batches) and must survive the eviction.
// Entries outlive emgpu_model_free (a model does not know the contexts that uploaded it): they are
// reclaimed by this LRU sweep or by emgpu_ctx_free; at most 48 + the models of one call stay resident.
static Uploaded &get_uploaded(emgpu_ctx *ctx, const emgpu_model *h, const std::set<uint64_t> *pinned = nullptr) {
    if (ctx->cache.size() > 48 && ctx->cache.find(h->m.uid) == ctx->cache.end()) {
        // models come and go (their tables stay uploaded): drop the least recently used half
        HIP_OK(hipStreamSynchronize(ctx->stream));
        std::vector<std::pair<uint64_t, uint64_t>> byuse;
        for (auto &kv : ctx->cache)
            if (!pinned || !pinned->count(kv.first)) byuse.push_back({kv.second.last_use, kv.first});
        std::sort(byuse.begin(), byuse.end());
        for (size_t q = 0; q < byuse.size() / 2; q++) {
            ctx->cache[byuse[q].second].free_tables();
            ctx->cache.erase(byuse[q].second);
        }
    }
    Uploaded &u = ctx->cache[h->m.uid];
    u.last_use = ++ctx->use_clock;
    if (u.version == h->m.version && u.d_thr) return u;
    // (re)compile: tables depend on N, alpha, start, boundaries, rates
    CompiledPlan cp = *emgpu::plan_of(h->m);   // (the model keeps its plan: a second ctx, or a model read from the binary cache, does not search the thresholds again)
    HIP_OK(hipStreamSynchronize(ctx->stream)); // nothing in flight may still read the old tables
    u.free_tables();
    // 64 words of slack: k_terminal_propagate reads a row's thresholds in fixed groups (8, or every 6th up to index 41) and masks
    // the ones past the row's end instead of clamping every index
    const size_t nthr = cp.thr.size() + 64;
    HIP_OK(hipMalloc((void **)&u.d_thr, nthr * sizeof(uint32_t)));
    HIP_OK(hipMemset(u.d_thr + cp.thr.size(), 0xFF, 64 * sizeof(uint32_t)));
    HIP_OK(hipMalloc((void **)&u.d_bnd, cp.bnd.size() * sizeof(double)));
    if (!cp.thr.empty()) HIP_OK(hipMemcpy(u.d_thr, cp.thr.data(), cp.thr.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(u.d_bnd, cp.bnd.data(), cp.bnd.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_OK(hipMalloc((void **)&u.d_cthr, (cp.cthr.size() ? cp.cthr.size() : 1) * sizeof(uint32_t)));
    if (!cp.cthr.empty()) HIP_OK(hipMemcpy(u.d_cthr, cp.cthr.data(), cp.cthr.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    u.cp = std::move(cp);
    u.cp.plan.thr = u.d_thr;
    u.cp.plan.bnd = u.d_bnd;
    u.cp.plan.cthr = u.d_cthr;
    HIP_OK(hipMalloc((void **)&u.d_pthr, (u.cp.pthr.size() ? u.cp.pthr.size() : 4) * sizeof(uint32_t)));
    if (!u.cp.pthr.empty()) HIP_OK(hipMemcpy(u.d_pthr, u.cp.pthr.data(), u.cp.pthr.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    u.cp.plan.pthr = u.d_pthr;
    {   // the finished plan (device pointers in place) next to its tables, for k_uncor_fast_mixed
        std::vector<char> pf(emgpu::plan_f_bytes());
        emgpu::plan_f_fill(u.cp.plan, pf.data());
        HIP_OK(hipMalloc(&u.d_planf, pf.size()));
        HIP_OK(hipMemcpy(u.d_planf, pf.data(), pf.size(), hipMemcpyHostToDevice));
    }
    u.version = h->m.version;
    return u;
}

// the log-probability table behind per-sample log-weights, uploaded on first use (with the model version it was built for: get_uploaded
// frees every table when the model changes)
static void ensure_logp(emgpu_ctx *ctx, Uploaded &u, const Model &m) {
    if (u.d_logp) return;
    const std::vector<double> lp = emgpu::initial_log_prob(m, u.lp_off);
    HIP_OK(hipMalloc((void **)&u.d_logp, (lp.size() + 1) * sizeof(double)));
    HIP_OK(hipMemcpyAsync(u.d_logp, lp.data(), lp.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_OK(hipStreamSynchronize(ctx->stream));
}

// slot-th scratch buffer of the ctx, at least `bytes` long (contents undefined; valid until the next request for the same slot)
void *ctx_scratch(emgpu_ctx *ctx, size_t slot, size_t bytes) {
    if (ctx->scratch.size() <= slot) ctx->scratch.resize(slot + 1);
    emgpu_ctx::Scratch &sc = ctx->scratch[slot];
    if (sc.cap < bytes || !sc.p) {
        HIP_OK(hipStreamSynchronize(ctx->stream));
        if (sc.p) { HIP_OK(hipFree(sc.p)); sc.p = nullptr; sc.cap = 0; }
        const size_t want = bytes + bytes / 8 + 256;   // some headroom: batch sizes that wobble do not reallocate
        HIP_OK(hipMalloc(&sc.p, want));
        sc.cap = want;
    }
    return sc.p;
}

static void fill_run(emgpu_ctx *ctx, const Uploaded &u, const Model &m, const emgpu_sample_params *p, EmgpuRun &A) {
    memset(&A, 0, sizeof A);
    if (p->n < 0 || p->sample_time < 1 || p->sample_time > 65535) throw Error(EMGPU_ERR_ARG, "n < 0 or sample_time outside 1..65535");
    if (p->max_attempts < 1) throw Error(EMGPU_ERR_ARG, "max_attempts must be >= 1");
    A.seed = p->seed; A.first_index = p->first_index; A.n = p->n; A.T = p->sample_time;
    A.per_step = p->transition_mode == EMGPU_TRANSITION_PER_STEP;
    A.flags = p->flags; A.max_attempts = p->max_attempts;
    auto pos = [&](int idx1) -> int {
        if (idx1 == 0) return -1;
        if (idx1 < 1 || idx1 > m.n_initial) throw Error(EMGPU_ERR_ARG, "variable index out of range");
        return u.cp.pos_of_var[idx1 - 1];
    };
    A.pos_L = pos(p->idx_L); A.pos_v = pos(p->idx_v); A.pos_dh = pos(p->idx_dh);
    A.n_layers = 0; A.layers = nullptr;
    if (p->layers && p->n_layers > 0) {
        if (p->idx_L == 0) throw Error(EMGPU_ERR_ARG, "layers given without idx_L");
        if (!m.boundaries[p->idx_L - 1].empty() && !(p->flags & EMGPU_FLAG_NO_DEDISC))
            throw Error(EMGPU_ERR_ARG, "layers need L to stay a bin index (isOverwriteZeroBoundaries)");
        if (p->n_layers < m.r_initial[p->idx_L - 1]) throw Error(EMGPU_ERR_ARG, "layers has fewer rows than L has bins");
        const size_t bytes = (size_t)p->n_layers * 2 * sizeof(double);
        if (ctx->d_layers_cap < bytes) {
            HIP_OK(hipStreamSynchronize(ctx->stream));
            if (ctx->d_layers) HIP_OK(hipFree(ctx->d_layers));
            HIP_OK(hipMalloc((void **)&ctx->d_layers, bytes));
            ctx->d_layers_cap = bytes;
        }
        HIP_OK(hipMemcpyAsync(ctx->d_layers, p->layers, bytes, hipMemcpyHostToDevice, ctx->stream));
        HIP_OK(hipStreamSynchronize(ctx->stream)); // p->layers is caller memory
        A.layers = ctx->d_layers; A.n_layers = p->n_layers;
    }
    A.event_cap = p->event_cap;
    A.status = ctx->d_status;
    A.ld = p->n;
    A.indices = p->indices;
}

// Point the run at the caller's buffers: column col_offset of arrays whose trajectory dimension is ld.
static void bind_outputs(EmgpuRun &A, const Model &m, const emgpu_sample_params *p, const emgpu_sample_out *out, int64_t extra_offset = 0) {
    if ((out->ev_count != nullptr) != (out->events != nullptr)) throw Error(EMGPU_ERR_ARG, "ev_count and events go together");
    if (out->events && p->event_cap < 1) throw Error(EMGPU_ERR_ARG, "event_cap must be >= 1");
    const int64_t ld = out->ld ? out->ld : A.n, off = out->col_offset + extra_offset;
    if (ld < 0 || off < 0 || off + A.n > ld) throw Error(EMGPU_ERR_ARG, "col_offset + n exceeds ld");
    A.ld = ld;
    A.col0 = off;
    const size_t o = (size_t)off;
    A.init_bin = out->init_bin ? out->init_bin + o : nullptr;
    A.init_val = out->init_val ? out->init_val + o : nullptr;
    A.dyn_bin = out->dyn_bin ? out->dyn_bin + o : nullptr;
    A.dyn_val = out->dyn_val ? out->dyn_val + 4 * o : nullptr;
    A.ev_count = out->ev_count ? out->ev_count + o : nullptr;
    A.events = out->events ? reinterpret_cast<uint64_t *>(out->events) + o * (size_t)p->event_cap : nullptr;
    A.attempts = out->attempts ? out->attempts + o : nullptr;
    (void)m;
}

static void launch_dbn(emgpu_ctx *ctx, const Uploaded &u, const EmgpuRun &A, hipStream_t stream = nullptr, const EmgpuPresets *presets = nullptr) {
    const char *name = "";
    hipError_t e;
    if (!stream) stream = ctx->stream;
    if (presets) e = emgpu::launch_dbn_generic(u.cp.plan, A, stream, &name, presets);   // a start grid / per-sample log-weights: the general kernel
    else if (emgpu::fast_uncor_eligible(u.cp.plan, A)) e = emgpu::launch_uncor_fast(u.cp.plan, A, stream, &name);
    else if (emgpu::step2_eligible(u.cp.plan, A)) e = emgpu::launch_dbn_step2(u.cp.plan, A, stream, &name);
    else if (emgpu::step_eligible(u.cp.plan, A)) e = emgpu::launch_dbn_step(u.cp.plan, A, stream, &name);
    else e = emgpu::launch_dbn_generic(u.cp.plan, A, stream, &name);
    ctx->last_kernel = name;
    ctx->last_launches++;
    if (e != hipSuccess) throw Error(EMGPU_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
}

// Run fn(d) for d = 0..n-1 on one host thread per device and fold the results: the first
// failing device's status and message become the caller's.
template <typename F>
static int on_devices(int n, F fn) {
    std::vector<int> rc((size_t)n, EMGPU_OK);
    std::vector<std::string> msg((size_t)n);
    auto body = [&](int d) {
        rc[d] = fn(d);
        if (rc[d] != EMGPU_OK) msg[d] = g_err; // g_err is thread-local
    };
    std::vector<std::thread> th;
    for (int d = 1; d < n; d++) th.emplace_back(body, d);
    body(0);
    for (auto &t : th) t.join();
    for (int d = 0; d < n; d++)
        if (rc[d] != EMGPU_OK) return fail(rc[d], "device " + std::to_string(d) + ": " + msg[d]);
    return EMGPU_OK;
}

extern "C" {

int emgpu_sample_dbn_device(emgpu_ctx *ctx, const emgpu_model *h, const emgpu_sample_params *p, const emgpu_sample_out *out) {
    EMGPU_TRY
    if (!ctx || !h || !p || !out) return fail(EMGPU_ERR_ARG, "null argument");
    CTX_LOCK(ctx);
    HIP_OK(hipSetDevice(ctx->device));
    Uploaded &u = get_uploaded(ctx, h);
    EmgpuRun A;
    fill_run(ctx, u, h->m, p, A);
    bind_outputs(A, h->m, p, out);
    const EmgpuPresets *presets = nullptr;
    if (p->start || out->log_weight) {   // a start grid / per-sample log-weights: a small block of device memory the kernel reads them through
        EmgpuPresets Q;
        memset(&Q, 0, sizeof Q);
        Q.start = p->start; Q.log_weight = out->log_weight;
        if (Q.log_weight) { ensure_logp(ctx, u, h->m); Q.logp = u.d_logp; memcpy(Q.lp_off, u.lp_off, sizeof Q.lp_off); }
        if (!ctx->d_presets) HIP_OK(hipMalloc((void **)&ctx->d_presets, sizeof(EmgpuPresets)));
        HIP_OK(hipStreamSynchronize(ctx->stream));   // (an earlier launch may still read the block)
        HIP_OK(hipMemcpyAsync(ctx->d_presets, &Q, sizeof Q, hipMemcpyHostToDevice, ctx->stream));
        HIP_OK(hipStreamSynchronize(ctx->stream));   // Q is a local
        presets = ctx->d_presets;
    }
    ctx->last_launches = 0;
    launch_dbn(ctx, u, A, nullptr, presets);
    return EMGPU_OK;
    EMGPU_CATCH
}

int emgpu_shard_range(int64_t n_total, int32_t rank, int32_t world, int64_t *lo, int64_t *hi) {
    if (n_total < 0 || world < 1 || rank < 0 || rank >= world || !lo || !hi) return fail(EMGPU_ERR_ARG, "bad shard arguments");
    const int64_t base = n_total / world, rem = n_total % world;
    *lo = rank * base + (rank < rem ? rank : rem);
    *hi = *lo + base + (rank < rem ? 1 : 0);
    return EMGPU_OK;
}

int emgpu_device_count(int32_t *count) {
    if (!count) return fail(EMGPU_ERR_ARG, "null argument");
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
    *count = c;
    return EMGPU_OK;
}

int32_t emgpu_mixed_blocks(int64_t n_total, int32_t n_models, int64_t lo, int64_t hi, emgpu_block *blocks) {
    if (n_total < 0 || n_models < 1 || !blocks || lo < 0 || hi > n_total) return fail(EMGPU_ERR_ARG, "bad block arguments");
    int32_t k = 0;
    for (int32_t m = 0; m < n_models; m++) {
        int64_t a, b;
        emgpu_shard_range(n_total, m, n_models, &a, &b);
        const int64_t a2 = a > lo ? a : lo, b2 = b < hi ? b : hi;
        if (b2 > a2) blocks[k++] = emgpu_block{m, 0, (uint64_t)a2, b2 - a2};
    }
    return k;
}

int emgpu_sample_dbn_blocks_device(emgpu_ctx *ctx, const emgpu_model *const *models, int32_t n_models,
                                   const emgpu_sample_params *p, const emgpu_block *blocks, int32_t n_blocks,
                                   const emgpu_sample_out *out) {
    EMGPU_TRY
    if (!ctx || !models || n_models < 1 || !p || (!blocks && n_blocks > 0) || n_blocks < 0 || !out) return fail(EMGPU_ERR_ARG, "null argument");
    if (p->start || out->log_weight) return fail(EMGPU_ERR_UNSUPPORTED, "a start grid / log-weights in a mixed-model batch: sample the models one by one");
    CTX_LOCK(ctx);
    HIP_OK(hipSetDevice(ctx->device));
    std::set<uint64_t> pinned;
    for (int i = 0; i < n_models; i++) {
        if (!models[i]) return fail(EMGPU_ERR_ARG, "null model");
        if (models[i]->m.n_initial != models[0]->m.n_initial || models[i]->m.n_dyn() != models[0]->m.n_dyn())
            return fail(EMGPU_ERR_ARG, "the models of a mixed batch must agree in n_initial and n_dyn (one trace shape)");
        pinned.insert(models[i]->m.uid);
    }
    emgpu_sample_out o = *out;
    if (!o.ld) o.ld = p->n;
    // Everything that can fail -- argument checks, table uploads (get_uploaded may synchronise the ctx stream), the runs and
    // their output bindings -- happens before the first launch: an error leaves nothing in flight.
    struct Live { const Uploaded *u; EmgpuRun A; int64_t col; int shape; };
    std::vector<Live> live;
    for (int b = 0; b < n_blocks; b++) {
        const emgpu_block &B = blocks[b];
        if (B.model < 0 || B.model >= n_models) return fail(EMGPU_ERR_ARG, "block names a model outside the list");
        if (B.n < 0 || B.first_index < p->first_index || B.first_index - p->first_index + (uint64_t)B.n > (uint64_t)p->n)
            return fail(EMGPU_ERR_ARG, "block outside [first_index, first_index + n)");
        if (B.n == 0) continue;
        const emgpu_model *h = models[B.model];
        const Uploaded &u = get_uploaded(ctx, h, &pinned);   // std::map: the reference stays valid, and nothing pinned is evicted
        const int64_t col = (int64_t)(B.first_index - p->first_index);
        emgpu_sample_params q = *p;
        q.first_index = B.first_index; q.n = B.n;
        if (p->indices) q.indices = p->indices + col;   // an index list covers the call's columns: block b draws its own entries
        Live L;
        L.u = &u; L.col = col;
        fill_run(ctx, u, h->m, &q, L.A);
        bind_outputs(L.A, h->m, &q, &o, col);
        // (event lists: k_uncor_fast_ev, one launch per block -- the shared launch writes the dense trace only)
        // (... and stores both dense outputs unconditionally)
        L.shape = (emgpu::fast_uncor_eligible(u.cp.plan, L.A) && L.A.ev_count == nullptr && L.A.dyn_bin != nullptr && L.A.dyn_val != nullptr) ? emgpu::uncor_fast_shape(u.cp.plan) : -1;
        live.push_back(L);
    }
    ctx->last_launches = 0;
    if (live.empty()) return EMGPU_OK;
    // Blocks whose models run on the same k_uncor_fast instance share ONE launch (model id per workgroup); every other block
    // is its own launch.  With several launches they go round-robin over the ctx stream and three side streams, forked from
    // and joined back into the ctx stream with events (a caller sees one in-order operation).
    static const bool no_mixed_env = getenv("EMGPU_DEBUG_NO_MIXED_LAUNCH") != nullptr;
    const bool no_mixed = no_mixed_env || p->indices != nullptr;   // (an index list: one launch per block, k_uncor_fast_idx)
    struct Launch { int shape; std::vector<int> members; };
    std::vector<Launch> launches;
    // the run of a shared launch is common to its blocks but for the index range and the columns: the rejection test's variables
    // must sit at the same topological positions in every member
    auto same_positions = [](const EmgpuRun &a, const EmgpuRun &b) { return a.pos_L == b.pos_L && a.pos_v == b.pos_v && a.pos_dh == b.pos_dh; };
    for (int i = 0; i < (int)live.size(); i++) {
        Launch *into = nullptr;
        if (live[i].shape >= 0 && !no_mixed)
            for (auto &g : launches)
                if (g.shape == live[i].shape && (int)g.members.size() < EMGPU_MAX_MIXED && same_positions(live[g.members[0]].A, live[i].A)) { into = &g; break; }
        if (into) into->members.push_back(i);
        else launches.push_back(Launch{live[i].shape, {i}});
    }
    static const bool no_side = getenv("EMGPU_DEBUG_NO_SIDE_STREAMS") != nullptr;
    const bool fork = launches.size() >= 2 && !no_side;
    int used = 0; // side streams in use
    struct Join {   // runs on every way out: side streams that were forked are joined back even when a launch fails
        emgpu_ctx *ctx; int *used;
        ~Join() {
            for (int q = 0; q < *used; q++) {
                (void)hipEventRecord(ctx->ev_join[q], ctx->side[q]);
                (void)hipStreamWaitEvent(ctx->stream, ctx->ev_join[q], 0);
            }
        }
    } join{ctx, &used};
    if (fork) {
        if (!ctx->ev_fork) {
            HIP_OK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
            for (int q = 0; q < emgpu_ctx::kSide; q++) {
                HIP_OK(hipStreamCreateWithFlags(&ctx->side[q], hipStreamNonBlocking));
                HIP_OK(hipEventCreateWithFlags(&ctx->ev_join[q], hipEventDisableTiming));
            }
        }
        HIP_OK(hipEventRecord(ctx->ev_fork, ctx->stream));
    }
    for (int g = 0; g < (int)launches.size(); g++) {
        hipStream_t st = ctx->stream;
        if (fork) {
            const int lane = g % (emgpu_ctx::kSide + 1);
            if (lane > 0) {
                st = ctx->side[lane - 1];
                if (lane > used) { HIP_OK(hipStreamWaitEvent(st, ctx->ev_fork, 0)); used = lane; }
            }
        }
        const Launch &G = launches[g];
        if (G.members.size() == 1) { launch_dbn(ctx, *live[G.members[0]].u, live[G.members[0]].A, st); continue; }
        // one launch for the group: the call's run with the outputs at column 0, the blocks as (plan, index range, column)
        const void *planf[EMGPU_MAX_MIXED];
        uint64_t first[EMGPU_MAX_MIXED];
        int64_t nn[EMGPU_MAX_MIXED], col[EMGPU_MAX_MIXED];
        for (size_t q = 0; q < G.members.size(); q++) {
            const Live &L = live[G.members[q]];
            planf[q] = L.u->d_planf; first[q] = L.A.first_index; nn[q] = L.A.n; col[q] = L.col;
        }
        const Live &L0 = live[G.members[0]];
        EmgpuRun A = L0.A;
        emgpu_sample_params q0 = *p;
        q0.n = L0.A.n;   // bind_outputs checks col + n against ld: column 0 with the first member's n always fits
        bind_outputs(A, models[0]->m, &q0, &o, 0);
        A.ld = L0.A.ld;
        const char *name = "";
        hipError_t e = emgpu::launch_uncor_fast_mixed(A, (int)G.members.size(), planf, first, nn, col, G.shape, st, &name);
        ctx->last_kernel = name;
        ctx->last_launches++;
        if (e != hipSuccess) throw Error(EMGPU_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    }
    return EMGPU_OK;
    EMGPU_CATCH
}

int emgpu_sample_dbn_multi_device(emgpu_ctx *const *ctxs, int32_t n_ctx, const emgpu_model *h,
                                  const emgpu_sample_params *p, const emgpu_sample_out *outs) {
    EMGPU_TRY
    if (!ctxs || n_ctx < 1 || !h || !p || !outs) return fail(EMGPU_ERR_ARG, "null argument");
    for (int d = 0; d < n_ctx; d++) if (!ctxs[d]) return fail(EMGPU_ERR_ARG, "null ctx");
    return on_devices(n_ctx, [&](int d) {
        int64_t lo, hi;
        emgpu_shard_range(p->n, d, n_ctx, &lo, &hi);
        emgpu_sample_params q = *p;
        q.first_index = p->first_index + (uint64_t)lo; q.n = hi - lo;
        if (p->indices) q.indices = p->indices + lo;   // shard d draws entries [lo, hi) of the list (device pointer valid on every device: managed / peer memory is the caller's business)
        if (p->start) q.start = p->start + (size_t)lo * (size_t)h->m.n_initial;   // (the same for the start grid; outs[d].log_weight is the shard's own)
        return emgpu_sample_dbn_device(ctxs[d], h, &q, &outs[d]);
    });
    EMGPU_CATCH
}

int emgpu_sample_dbn_multi_host(emgpu_ctx *const *ctxs, int32_t n_ctx, const emgpu_model *h,
                                const emgpu_sample_params *p, const emgpu_sample_out *out) {
    EMGPU_TRY
    if (!ctxs || n_ctx < 1 || !h || !p || !out) return fail(EMGPU_ERR_ARG, "null argument");
    for (int d = 0; d < n_ctx; d++) if (!ctxs[d]) return fail(EMGPU_ERR_ARG, "null ctx");
    if (p->n < 0) return fail(EMGPU_ERR_ARG, "n < 0");
    return on_devices(n_ctx, [&](int d) {
        int64_t lo, hi;
        emgpu_shard_range(p->n, d, n_ctx, &lo, &hi);
        emgpu_sample_params q = *p;
        q.first_index = p->first_index + (uint64_t)lo; q.n = hi - lo;
        if (p->indices) q.indices = p->indices + lo;   // host list: shard d uploads and draws its own entries [lo, hi)
        if (p->start) q.start = p->start + (size_t)lo * (size_t)h->m.n_initial;   // ... and its own rows of the start grid
        emgpu_sample_out o = *out;
        o.ld = out->ld ? out->ld : p->n;
        o.col_offset = out->col_offset + lo;
        if (out->log_weight) o.log_weight = out->log_weight + lo;
        return emgpu_sample_dbn_host(ctxs[d], h, &q, &o);
    });
    EMGPU_CATCH
}

// (emgpu_sample_dbn_host, the trace pool and the pinned pool: emgpu_host.cpp)

static void fill_bn(emgpu_ctx *ctx, const Uploaded &u, const Model &m, const emgpu_bn_params *p, EmgpuBnRun &A) {
    memset(&A, 0, sizeof A);
    if (p->n < 0 || p->max_attempts < 1) throw Error(EMGPU_ERR_ARG, "n < 0 or max_attempts < 1");
    A.seed = p->seed; A.first_index = p->first_index; A.n = p->n; A.ld = p->n; A.flags = p->flags; A.max_attempts = p->max_attempts;
    A.has_bounds = p->bounds_sample != nullptr;
    if (p->bounds_sample)
        for (int v = 0; v < m.n_initial; v++) {
            A.bounds[u.cp.pos_of_var[v]][0] = p->bounds_sample[2 * v];
            A.bounds[u.cp.pos_of_var[v]][1] = p->bounds_sample[2 * v + 1];
        }
    A.pos_own_speed = A.pos_int_speed = -1;
    if (p->idx_own_speed > 0 || p->idx_int_speed > 0) {
        if (p->idx_own_speed < 1 || p->idx_own_speed > m.n_initial || p->idx_int_speed < 1 || p->idx_int_speed > m.n_initial)
            throw Error(EMGPU_ERR_ARG, "speed variable index out of range");
        A.pos_own_speed = u.cp.pos_of_var[p->idx_own_speed - 1];
        A.pos_int_speed = u.cp.pos_of_var[p->idx_int_speed - 1];
    }
    A.min1 = p->min_vel1; A.max1 = p->max_vel1; A.min2 = p->min_vel2; A.max2 = p->max_vel2;
    A.status = ctx->d_status;
    A.start = p->start; A.log_weight = p->log_weight;
}

int emgpu_sample_bn_device(emgpu_ctx *ctx, const emgpu_model *h, const emgpu_bn_params *p, uint8_t *out_bin, float *out_val, int32_t *attempts) {
    EMGPU_TRY
    if (!ctx || !h || !p) return fail(EMGPU_ERR_ARG, "null argument");
    CTX_LOCK(ctx);
    HIP_OK(hipSetDevice(ctx->device));
    Uploaded &u = get_uploaded(ctx, h);
    EmgpuBnRun A;
    fill_bn(ctx, u, h->m, p, A);
    A.out_bin = out_bin; A.out_val = out_val; A.attempts = attempts;
    if (A.log_weight) { ensure_logp(ctx, u, h->m); A.logp = u.d_logp; memcpy(A.lp_off, u.lp_off, sizeof A.lp_off); }
    const char *name = "";
    hipError_t e = emgpu::launch_bn(u.cp.plan, A, ctx->stream, &name);
    ctx->last_kernel = name;
    if (e != hipSuccess) return fail(EMGPU_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return EMGPU_OK;
    EMGPU_CATCH
}

int emgpu_sample_bn_host(emgpu_ctx *ctx, const emgpu_model *h, const emgpu_bn_params *p, uint8_t *out_bin, float *out_val, int32_t *attempts) {
    EMGPU_TRY
    if (!ctx || !h || !p) return fail(EMGPU_ERR_ARG, "null argument");
    CTX_LOCK(ctx);
    HIP_OK(hipSetDevice(ctx->device));
    const size_t n = (size_t)(p->n > 0 ? p->n : 0), ni = h->m.n_initial;
    uint8_t *db = nullptr; float *dv = nullptr; int32_t *da = nullptr, *ds = nullptr; double *dw = nullptr;
    int rc;
    try {
        if (out_bin) HIP_OK(hipMalloc((void **)&db, ni * n + 1));
        if (out_val) HIP_OK(hipMalloc((void **)&dv, ni * n * 4 + 4));
        if (attempts) HIP_OK(hipMalloc((void **)&da, n * 4 + 4));
        emgpu_bn_params pd = *p;
        if (p->start && n) {
            HIP_OK(hipMalloc((void **)&ds, n * ni * 4));
            HIP_OK(hipMemcpyAsync(ds, p->start, n * ni * 4, hipMemcpyHostToDevice, ctx->stream));
            HIP_OK(hipStreamSynchronize(ctx->stream));
            pd.start = ds;
        }
        if (p->log_weight) { HIP_OK(hipMalloc((void **)&dw, n * 8 + 8)); pd.log_weight = dw; }
        rc = emgpu_sample_bn_device(ctx, h, &pd, db, dv, da);
        if (rc == EMGPU_OK) {
            if (p->log_weight && n) HIP_OK(hipMemcpyAsync(p->log_weight, dw, n * 8, hipMemcpyDeviceToHost, ctx->stream));
            if (out_bin && n) HIP_OK(hipMemcpyAsync(out_bin, db, ni * n, hipMemcpyDeviceToHost, ctx->stream));
            if (out_val && n) HIP_OK(hipMemcpyAsync(out_val, dv, ni * n * 4, hipMemcpyDeviceToHost, ctx->stream));
            if (attempts && n) HIP_OK(hipMemcpyAsync(attempts, da, n * 4, hipMemcpyDeviceToHost, ctx->stream));
            rc = emgpu_ctx_sync(ctx);
        }
    } catch (...) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(db); (void)hipFree(dv); (void)hipFree(da); (void)hipFree(ds); (void)hipFree(dw);
        throw;
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(db); (void)hipFree(dv); (void)hipFree(da); (void)hipFree(ds); (void)hipFree(dw);
    return rc;
    EMGPU_CATCH
}

int emgpu_debug_terminal_counters(emgpu_ctx *ctx, uint64_t *out, int32_t n) {
    EMGPU_TRY
    if (!ctx || !out || n < 1) return fail(EMGPU_ERR_ARG, "null argument");
    CTX_LOCK(ctx);
    HIP_OK(hipSetDevice(ctx->device));
    HIP_OK(hipStreamSynchronize(ctx->stream));
    const int rc = emgpu::terminal_debug_counters((unsigned long long *)out, n);
    if (rc < 0) return fail(EMGPU_ERR_HIP, "reading the counters failed");
    return rc;
    EMGPU_CATCH
}

// Test hook: k_uncor_track on caller-given initial values and control rows (no sampling, no limits): what the point-mass step does with
// inputs a sampled batch seldom holds (a pitch command clamped to +-90 degrees that changes sign).
int emgpu_debug_uncor_dynamics_host(emgpu_ctx *ctx, int64_t n, int32_t T, int32_t record_stride, int32_t literal, const double dyn[6],
                                    const float *init, const float *controls, double *tracks) {
    EMGPU_TRY
    if (!ctx || !dyn || !init || !controls || !tracks || n < 1 || T < 1 || record_stride < 1 || (10 * T) % record_stride) return fail(EMGPU_ERR_ARG, "bad argument");
    CTX_LOCK(ctx);
    HIP_OK(hipSetDevice(ctx->device));
    const size_t G4 = (size_t)(T + 3) / 4, S = (size_t)(10 * T / record_stride + 1);
    std::vector<float> h_init(5 * (size_t)n), h_dyn(G4 * 3 * (size_t)n * 4, 0.f);
    for (int64_t i = 0; i < n; i++) {
        for (int k = 0; k < 5; k++) h_init[(size_t)k * n + i] = init[i * 5 + k];      // rows: L v \dot v \dot h \dot psi
        for (int c = 0; c < T; c++)
            for (int k = 0; k < 3; k++) h_dyn[((((size_t)c / 4) * 3 + k) * n + i) * 4 + c % 4] = controls[((size_t)i * T + c) * 3 + k];   // rows: \dot h, \dot psi, \dot v
    }
    float *d_init = nullptr, *d_dyn = nullptr; double *d_tr = nullptr, *d_lim = nullptr; uint8_t *d_acc = nullptr;
    int rc = EMGPU_OK;
    try {
        HIP_OK(hipMalloc((void **)&d_init, h_init.size() * 4)); HIP_OK(hipMalloc((void **)&d_dyn, h_dyn.size() * 4));
        HIP_OK(hipMalloc((void **)&d_tr, (size_t)n * S * 8 * 8)); HIP_OK(hipMalloc((void **)&d_lim, 3 * 8)); HIP_OK(hipMalloc((void **)&d_acc, (size_t)n));
        const double lim[3] = {-1e300, 1e300, 1e300};
        HIP_OK(hipMemcpyAsync(d_init, h_init.data(), h_init.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        HIP_OK(hipMemcpyAsync(d_dyn, h_dyn.data(), h_dyn.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        HIP_OK(hipMemcpyAsync(d_lim, lim, sizeof lim, hipMemcpyHostToDevice, ctx->stream));
        EmgpuUTrackRun R;
        memset(&R, 0, sizeof R);
        R.n = n; R.ld = n; R.T = T; R.stride = record_stride;
        R.iL = d_init; R.iV = d_init + n; R.iDV = d_init + 2 * n; R.iDH = d_init + 3 * n; R.iDPsi = d_init + 4 * n;
        R.dyn_val = d_dyn; R.nd = 3; R.sDH = 0; R.sDPsi = 1; R.sDV = 2;
        memcpy(R.dyn, dyn, sizeof R.dyn);
        R.min_alt = -1e300; R.max_alt = 1e300; R.ordered = 0; R.lim = d_lim;
        R.tracks = d_tr; R.S = (int64_t)S; R.accepted = d_acc;
        const char *name = "";
        hipError_t e = emgpu::launch_uncor_track(R, ctx->stream, &name, literal);
        ctx->last_kernel = name;
        if (e != hipSuccess) throw Error(EMGPU_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
        HIP_OK(hipMemcpyAsync(tracks, d_tr, (size_t)n * S * 8 * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_OK(hipStreamSynchronize(ctx->stream));
    } catch (...) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(d_init); (void)hipFree(d_dyn); (void)hipFree(d_tr); (void)hipFree(d_lim); (void)hipFree(d_acc);
        throw;
    }
    (void)hipFree(d_init); (void)hipFree(d_dyn); (void)hipFree(d_tr); (void)hipFree(d_lim); (void)hipFree(d_acc);
    return rc;
    EMGPU_CATCH
}

int emgpu_debug_column_thresholds(const double *weights, int32_t r, uint32_t *out) {
    if (!weights || !out || r < 1 || r > EMGPU_MAX_R) return fail(EMGPU_ERR_ARG, "bad arguments");
    if (r > 1) emgpu::column_thresholds(weights, r, out);
    return EMGPU_OK;
}

uint32_t emgpu_debug_bernoulli_threshold(double rate) { return emgpu::bernoulli_threshold(rate); }

int emgpu_debug_dynamic_column(const emgpu_model *m, int32_t k, int64_t col, int32_t *tvar, int32_t *r, int64_t *q,
                               uint32_t *thr, int32_t *meff, uint32_t *cthr, uint32_t *map) {
    EMGPU_TRY
    if (!m || !tvar || !r || !q || !thr || !meff || !cthr || !map) return fail(EMGPU_ERR_ARG, "null argument");
    const std::shared_ptr<const emgpu::CompiledPlan> cp_keep = emgpu::plan_of(m->m); const emgpu::CompiledPlan &cp = *cp_keep;
    const EmgpuPlan &P = cp.plan;
    if (k < 0 || k >= P.nd) return fail(EMGPU_ERR_ARG, "no such dynamic variable");
    *tvar = (int32_t)P.d_tvar[k] + 1;
    *r = (int32_t)P.d_r[k];
    *q = m->m.q_transition[P.d_tvar[k]];
    if (col < 0 || col >= *q) return fail(EMGPU_ERR_ARG, "no such column");
    const int rm1 = *r - 1;
    for (int t = 0; t < rm1 && t < 15; t++) thr[t] = cp.thr[P.d_off[k] + (size_t)col * rm1 + t];
    *meff = (int32_t)P.d_meff[k];
    *map = 0u;
    if (*meff > 0) {
        const uint32_t *c = cp.cthr.data() + P.d_coff[k] + (size_t)col * (*meff + 1);
        for (int t = 0; t < *meff; t++) cthr[t] = c[t];
        *map = c[*meff];
    }
    return EMGPU_OK;
    EMGPU_CATCH
}

int emgpu_debug_parent_masks(const emgpu_model *m, uint32_t *cur_mask, uint32_t *new_mask) {
    EMGPU_TRY
    if (!m || !cur_mask || !new_mask) return fail(EMGPU_ERR_ARG, "null argument");
    const std::shared_ptr<const emgpu::CompiledPlan> cp_keep = emgpu::plan_of(m->m); const emgpu::CompiledPlan &cp = *cp_keep;
    emgpu::step_parent_masks(cp.plan, cur_mask, new_mask);
    return EMGPU_OK;
    EMGPU_CATCH
}

int emgpu_debug_padded_column(const emgpu_model *m, int32_t k, int64_t col, int32_t *width, uint32_t *words) {
    EMGPU_TRY
    if (!m || !width || !words) return fail(EMGPU_ERR_ARG, "null argument");
    const std::shared_ptr<const emgpu::CompiledPlan> cp_keep = emgpu::plan_of(m->m); const emgpu::CompiledPlan &cp = *cp_keep;
    const EmgpuPlan &P = cp.plan;
    if (k < 0 || k >= P.nd) return fail(EMGPU_ERR_ARG, "no such dynamic variable");
    if (col < 0 || col >= m->m.q_transition[P.d_tvar[k]]) return fail(EMGPU_ERR_ARG, "no such column");
    *width = (int32_t)P.d_pw[k];
    for (int t = 0; t < *width; t++) words[t] = cp.pthr[P.d_poff[k] + (size_t)col * (size_t)*width + t];
    return EMGPU_OK;
    EMGPU_CATCH
}

int emgpu_debug_pk_column(const emgpu_model *m, int32_t k, int64_t col, uint32_t *words) {
    EMGPU_TRY
    if (!m || !words) return fail(EMGPU_ERR_ARG, "null argument");
    const std::shared_ptr<const emgpu::CompiledPlan> cp_keep = emgpu::plan_of(m->m); const emgpu::CompiledPlan &cp = *cp_keep;
    const EmgpuPlan &P = cp.plan;
    if (k < 0 || k >= P.nd || P.d_pw[k] == 0) return fail(EMGPU_ERR_ARG, "no such padded dynamic variable");
    if (col < 0 || col >= m->m.q_transition[P.d_tvar[k]]) return fail(EMGPU_ERR_ARG, "no such column");
    for (int t = 0; t < 4; t++) words[t] = cp.pthr[P.d_poffpk[k] + (size_t)col * 4 + t];
    return EMGPU_OK;
    EMGPU_CATCH
}

// Upload / validate the trajectory models of a terminal call and publish their table pointers in ctx->d_thr_base.
// Returns the first model's uploaded plan (the shapes every model shares).
static uint32_t rel_piv(const EmgpuPlan &P, int k) { return P.d_pivoff[k] ? P.d_pivoff[k] - P.d_off[0] : 0u; } // pivot rows relative to the first dynamic table
static uint32_t rel_c8(const EmgpuPlan &P, int k) { return P.d_c8off[k] ? P.d_c8off[k] - P.d_off[0] : 0u; }     // compact rows likewise
static const Uploaded *terminal_tables(emgpu_ctx *ctx, const emgpu_model *const *models, int32_t n_models, const std::set<uint64_t> *also_pinned = nullptr) {
    std::vector<const uint32_t *> bases;
    const Uploaded *first = nullptr;
    std::set<uint64_t> pinned; // the table pointers collected below must survive the cache's LRU sweep
    if (also_pinned) pinned = *also_pinned;
    for (int i = 0; i < n_models; i++) {
        if (!models[i]) throw Error(EMGPU_ERR_ARG, "null model");
        pinned.insert(models[i]->m.uid);
    }
    for (int i = 0; i < n_models; i++) {
        Uploaded &u = get_uploaded(ctx, models[i], &pinned);
        const EmgpuPlan &P = u.cp.plan;
        if (P.ni != 6 || P.nd != 3 || P.depend) throw Error(EMGPU_ERR_UNSUPPORTED, "trajectory model must have 6 initial and 3 independent dynamic variables");
        for (int q = 0; q < 6; q++)
            if (P.i_var[q] != q) throw Error(EMGPU_ERR_UNSUPPORTED, "trajectory model initial network must be in index order");
        {   // createEncounter.m:93-265 propagates heading, altitude and speed (variables 4, 5, 6): the kernel handles them by name
            bool seen[3] = {false, false, false};
            for (int k = 0; k < 3; k++)
                if (P.d_ivar[k] >= 3 && P.d_ivar[k] <= 5) seen[P.d_ivar[k] - 3] = true;
            if (!(seen[0] && seen[1] && seen[2])) throw Error(EMGPU_ERR_UNSUPPORTED, "trajectory model: the dynamic variables must be heading, altitude and speed (variables 4, 5, 6)");
        }
        if (P.d_ivar[0] > 5 || P.i_nb[1] < 3 || P.i_nb[2] < 3 || P.i_nb[3] < 3 || P.i_nb[4] < 3 || P.i_nb[5] < 3)
            throw Error(EMGPU_ERR_UNSUPPORTED, "distance, bearing, heading, altitude and speed need boundaries");
        if (P.i_nb[1] > 66 || P.i_nb[2] > 66 || P.i_nb[3] > 66 || P.i_nb[4] > 66 || P.i_nb[5] > 66)
            throw Error(EMGPU_ERR_UNSUPPORTED, "more than 64 cut points in a trajectory-model variable");
        if (!first) first = &u;
        else {
            const EmgpuPlan &Q = first->cp.plan;
            // the initial networks may differ (2 or 3 intents): only the dynamic tables' relative layout must agree
            bool same = (P.d_off[1] - P.d_off[0]) == (Q.d_off[1] - Q.d_off[0]) && (P.d_off[2] - P.d_off[0]) == (Q.d_off[2] - Q.d_off[0]) &&
                        memcmp(P.d_r, Q.d_r, sizeof P.d_r) == 0 && rel_piv(P, 0) == rel_piv(Q, 0) && rel_piv(P, 1) == rel_piv(Q, 1) && rel_piv(P, 2) == rel_piv(Q, 2) &&
                        rel_c8(P, 0) == rel_c8(Q, 0) && rel_c8(P, 1) == rel_c8(Q, 1) && rel_c8(P, 2) == rel_c8(Q, 2) &&
                        memcmp(P.d_stride_static, Q.d_stride_static, sizeof P.d_stride_static) == 0 &&
                        memcmp(P.d_stride_cur, Q.d_stride_cur, sizeof P.d_stride_cur) == 0 && u.cp.bnd == first->cp.bnd && memcmp(P.i_boff, Q.i_boff, sizeof P.i_boff) == 0 &&
                        memcmp(P.d_ivar, Q.d_ivar, sizeof P.d_ivar) == 0 && memcmp(P.d_tvar, Q.d_tvar, sizeof P.d_tvar) == 0;
            if (!same) throw Error(EMGPU_ERR_UNSUPPORTED, "trajectory models differ in shape or boundaries");
        }
        bases.push_back(u.d_thr + P.d_off[0]); // tables of the dynamic variables, relative to the first one
    }
    first = &get_uploaded(ctx, models[0], &pinned); // std::map keeps references valid and nothing pinned was erased
    if (ctx->d_thr_base_cap < bases.size()) {
        HIP_OK(hipStreamSynchronize(ctx->stream));
        if (ctx->d_thr_base) HIP_OK(hipFree(ctx->d_thr_base));
        HIP_OK(hipMalloc((void **)&ctx->d_thr_base, bases.size() * sizeof(void *)));
        ctx->d_thr_base_cap = bases.size();
    }
    HIP_OK(hipMemcpyAsync(ctx->d_thr_base, bases.data(), bases.size() * sizeof(void *), hipMemcpyHostToDevice, ctx->stream));
    HIP_OK(hipStreamSynchronize(ctx->stream));
    return first;
}

int emgpu_propagate_terminal_device(emgpu_ctx *ctx, const emgpu_model *const *models, int32_t n_models,
                                    const emgpu_term_params *p, const double *geo, const int32_t *model_of,
                                    float *traj, int32_t *rows) {
    EMGPU_TRY
    if (!ctx || !models || n_models < 1 || !p || !geo || !model_of || !traj || !rows) return fail(EMGPU_ERR_ARG, "null argument");
    if (p->n < 0 || p->cap < 2 || p->max_resample < 1) return fail(EMGPU_ERR_ARG, "bad n / cap / max_resample");
    if (p->n >= ((int64_t)1 << 29)) return fail(EMGPU_ERR_ARG, "more than 2^29 encounters in one call");
    CTX_LOCK(ctx);
    HIP_OK(hipSetDevice(ctx->device));
    const Uploaded *first = terminal_tables(ctx, models, n_models);
    EmgpuTermRun A;
    memset(&A, 0, sizeof A);
    A.seed = p->seed; A.first_index = p->first_index; A.n = p->n; A.geo = geo; A.model_of = model_of; A.thr_base = ctx->d_thr_base;
    A.tmax_s = p->tmax_s; A.max_resample = p->max_resample; A.cap = p->cap;
    memcpy(A.dl, p->dyn_limits, sizeof A.dl);
    A.traj = traj; A.rows = rows; A.status = ctx->d_status; A.queue = ctx->d_queue;
    const char *name = "";
    if ((p->flags & EMGPU_FLAG_LOCAL_SMOOTH) && EMGPU_TERMINAL_BLOCK_ROWS(p->cap) > 256) return fail(EMGPU_ERR_UNSUPPORTED, "EMGPU_FLAG_LOCAL_SMOOTH: cap above 128");
    hipError_t e = emgpu::launch_terminal_propagate(first->cp.plan, A, ctx->stream, &name);
    ctx->last_kernel = name;
    if (e == hipSuccess && (p->flags & EMGPU_FLAG_LOCAL_SMOOTH)) {
        e = emgpu::launch_terminal_smooth(traj, rows, 2 * p->n, p->cap, ctx->stream);
        ctx->last_kernel += " + k_terminal_smooth";
    }
    if (e != hipSuccess) return fail(EMGPU_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return EMGPU_OK;
    EMGPU_CATCH
}

int emgpu_propagate_terminal_host(emgpu_ctx *ctx, const emgpu_model *const *models, int32_t n_models,
                                  const emgpu_term_params *p, const double *geo, const int32_t *model_of,
                                  float *out, int32_t *rows) {
    EMGPU_TRY
    if (!ctx || !p || !geo || !model_of || !out || !rows) return fail(EMGPU_ERR_ARG, "null argument");
    if (p->cap < 2) return fail(EMGPU_ERR_ARG, "bad cap");
    const size_t out_bytes = (size_t)(p->n > 0 ? p->n : 0) * 2 * (size_t)EMGPU_TERMINAL_BLOCK_ROWS(p->cap) * 5 * sizeof(float);
    CTX_LOCK(ctx);
    HIP_OK(hipSetDevice(ctx->device));
    const size_t n = (size_t)(p->n > 0 ? p->n : 0), nl = 4 * n;
    double *dg = nullptr; int32_t *dm = nullptr, *dr = nullptr; float *dout = nullptr;
    int rc;
    try {
        HIP_OK(hipMalloc((void **)&dg, n * 12 * sizeof(double) + 8));
        HIP_OK(hipMalloc((void **)&dm, nl * 4 + 4));
        HIP_OK(hipMalloc((void **)&dr, nl * 4 + 4));
        HIP_OK(hipMalloc((void **)&dout, out_bytes + 4));
        if (n) {
            HIP_OK(hipMemcpyAsync(dg, geo, n * 12 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            HIP_OK(hipMemcpyAsync(dm, model_of, nl * 4, hipMemcpyHostToDevice, ctx->stream));
            HIP_OK(hipMemsetAsync(dout, 0, out_bytes, ctx->stream));
        }
        rc = emgpu_propagate_terminal_device(ctx, models, n_models, p, dg, dm, dout, dr);
        if (rc == EMGPU_OK && n) {
            HIP_OK(hipMemcpyAsync(out, dout, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
            HIP_OK(hipMemcpyAsync(rows, dr, nl * 4, hipMemcpyDeviceToHost, ctx->stream));
        }
        if (rc == EMGPU_OK) rc = emgpu_ctx_sync(ctx);
    } catch (...) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(dg); (void)hipFree(dm); (void)hipFree(dr); (void)hipFree(dout);
        throw;
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(dg); (void)hipFree(dm); (void)hipFree(dr); (void)hipFree(dout);
    return rc;
    EMGPU_CATCH
}

int emgpu_sample_terminal_device(emgpu_ctx *ctx, const emgpu_model *gm, const emgpu_model *const *traj_models, int32_t n_traj_models,
                                 const emgpu_tsample_params *p, uint8_t *geom_bin, float *geom_val, double *geo, int32_t *model_of,
                                 float *traj, int32_t *rows, int32_t *attempts) {
    EMGPU_TRY
    if (!ctx || !gm || !traj_models || !p || !geom_val || !geo || !model_of || !traj || !rows) return fail(EMGPU_ERR_ARG, "null argument");
    if (p->n < 0 || p->cap < 2 || p->max_resample < 1 || p->max_attempts < 1) return fail(EMGPU_ERR_ARG, "bad n / cap / max_resample / max_attempts");
    if (p->n >= ((int64_t)1 << 29)) return fail(EMGPU_ERR_ARG, "more than 2^29 encounters in one call");
    if (n_traj_models != 10) return fail(EMGPU_ERR_ARG, "the terminal model has 10 trajectory models (CorTerminalModel.m:84-100)");
    const Model &g = gm->m;
    for (int k = 0; k < 12; k++) if (p->idx[k] < 1 || p->idx[k] > g.n_initial) return fail(EMGPU_ERR_ARG, "geometry variable index out of range");
    CTX_LOCK(ctx);
    HIP_OK(hipSetDevice(ctx->device));
    if (p->n == 0) return EMGPU_OK;
    std::set<uint64_t> pinned{gm->m.uid};   // the trajectory tables' pointers are published before the geometry model's upload
    for (int i = 0; i < n_traj_models; i++) if (traj_models[i]) pinned.insert(traj_models[i]->m.uid);
    const Uploaded *first = terminal_tables(ctx, traj_models, n_traj_models, &pinned);
    Uploaded &ug = get_uploaded(ctx, gm, &pinned);
    // geometry draw (sample.m:29-77)
    emgpu_bn_params bp;
    memset(&bp, 0, sizeof bp);
    bp.seed = p->seed; bp.first_index = p->first_index; bp.n = p->n;
    bp.max_attempts = p->max_attempts; bp.bounds_sample = p->bounds_sample;
    bp.idx_own_speed = p->idx[3]; bp.idx_int_speed = p->idx[9];
    bp.min_vel1 = p->dyn_limits[0][0]; bp.max_vel1 = p->dyn_limits[0][1]; bp.min_vel2 = p->dyn_limits[1][0]; bp.max_vel2 = p->dyn_limits[1][1];
    EmgpuBnRun B;
    fill_bn(ctx, ug, g, &bp, B);
    B.out_bin = geom_bin; B.out_val = geom_val; B.attempts = attempts;
    const char *name = "";
    hipError_t e = emgpu::launch_bn(ug.cp.plan, B, ctx->stream, &name);
    if (e != hipSuccess) return fail(EMGPU_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    std::string kernels = name;
    // createEncounter.m:21-49
    EmgpuTGeoRun G;
    memset(&G, 0, sizeof G);
    G.n = p->n; G.val = geom_val; G.geo = geo; G.model_of = model_of;
    for (int k = 0; k < 12; k++) G.idx[k] = p->idx[k] - 1;
    e = emgpu::launch_terminal_geo(G, ctx->stream);
    if (e != hipSuccess) return fail(EMGPU_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    kernels += " + k_terminal_geo";
    // PropagateTrajectory x 4 (createEncounter.m:52-72)
    EmgpuTermRun A;
    memset(&A, 0, sizeof A);
    A.seed = p->seed; A.first_index = p->first_index; A.n = p->n; A.geo = geo; A.model_of = model_of; A.thr_base = ctx->d_thr_base;
    A.tmax_s = p->tmax_s; A.max_resample = p->max_resample; A.cap = p->cap;
    memcpy(A.dl, p->dyn_limits, sizeof A.dl);
    A.traj = traj; A.rows = rows; A.status = ctx->d_status; A.queue = ctx->d_queue;
    e = emgpu::launch_terminal_propagate(first->cp.plan, A, ctx->stream, &name);
    if (e != hipSuccess) return fail(EMGPU_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    ctx->last_kernel = kernels + " + " + name;
    ctx->last_launches = 3;
    if (p->flags & EMGPU_FLAG_LOCAL_SMOOTH) {
        e = emgpu::launch_terminal_smooth(traj, rows, 2 * p->n, p->cap, ctx->stream);
        if (e != hipSuccess) return fail(EMGPU_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
        ctx->last_kernel = kernels + " + k_terminal_smooth + " + name;   // (the dominant kernel stays last in the list)
        ctx->last_launches = 4;
    }
    return EMGPU_OK;
    EMGPU_CATCH
}

int emgpu_track_terminal_host(emgpu_ctx *ctx, const emgpu_model *gm, const emgpu_model *const *traj_models, int32_t n_traj_models,
                              const emgpu_ttrack_params *p, double *sample, double *traj, int32_t cap2, int32_t *len,
                              double *meta, int32_t *attempts) {
    EMGPU_TRY
    if (!ctx || !gm || !traj_models || !p) return fail(EMGPU_ERR_ARG, "null argument");
    if (p->n < 0 || p->max_resample < 1 || p->max_track_attempts < 1 || p->max_attempts < 1 || !(p->tmax_s >= 1) || (traj && cap2 < 2))
        return fail(EMGPU_ERR_ARG, "bad n / caps / tmax_s");
    // CheckCumTurn (CorTerminalModel.m:135-185) keeps the merged track's heading differences per lane: 2 (tmax_s + 3) - 2 <= 248
    if (p->tmax_s > 122) return fail(EMGPU_ERR_UNSUPPORTED, "tmax_s > 122 (the reference default is 120)");
    if (n_traj_models != 10) return fail(EMGPU_ERR_ARG, "the terminal model has 10 trajectory models (CorTerminalModel.m:84-100)");
    const Model &g = gm->m;
    for (int k = 0; k < 12; k++) if (p->idx[k] < 1 || p->idx[k] > g.n_initial) return fail(EMGPU_ERR_ARG, "geometry variable index out of range");
    CTX_LOCK(ctx);
    HIP_OK(hipSetDevice(ctx->device));
    const size_t n = (size_t)p->n, ni = (size_t)g.n_initial;
    if (n == 0) return EMGPU_OK;
    const int cap = (int)p->tmax_s + 3;
    size_t slot = 0;
    auto dalloc = [&](size_t bytes) { return ctx_scratch(ctx, slot++, bytes ? bytes : 1); };   // kept by the ctx between calls
    int rc = EMGPU_OK;
    try {
        float *d_val = (float *)dalloc(ni * n * 4);
        double *d_geo = (double *)dalloc(n * 12 * 8);
        int32_t *d_mo = (int32_t *)dalloc(4 * n * 4), *d_rows = (int32_t *)dalloc(4 * n * 4);
        float *d_out = (float *)dalloc((size_t)2 * n * (size_t)EMGPU_TERMINAL_BLOCK_ROWS(cap) * 5 * 4);
        uint8_t *d_acc = (uint8_t *)dalloc(n);
        uint64_t *d_gidx[2] = {(uint64_t *)dalloc(n * 8), (uint64_t *)dalloc(n * 8)};
        int64_t *d_slot[2] = {(int64_t *)dalloc(n * 8), (int64_t *)dalloc(n * 8)};
        uint32_t *d_count = (uint32_t *)dalloc(4 * emgpu::compact_scratch_words((int64_t)n));
        double *d_sample = sample ? (double *)dalloc(n * ni * 8) : nullptr;
        double *d_traj = traj ? (double *)dalloc(n * 2 * (size_t)cap2 * 6 * 8) : nullptr;
        int32_t *d_len = len ? (int32_t *)dalloc(n * 2 * 4) : nullptr;
        double *d_meta = meta ? (double *)dalloc(n * 4 * 8) : nullptr;
        int32_t *d_att = (int32_t *)dalloc(n * 4);
        std::set<uint64_t> pinned{gm->m.uid};   // the trajectory tables' pointers are published before the geometry model's upload
        for (int i = 0; i < n_traj_models; i++) if (traj_models[i]) pinned.insert(traj_models[i]->m.uid);
        const Uploaded *first = terminal_tables(ctx, traj_models, n_traj_models, &pinned);
        Uploaded &ug = get_uploaded(ctx, gm, &pinned);
        emgpu_bn_params bp;
        memset(&bp, 0, sizeof bp);
        bp.max_attempts = p->max_attempts; bp.bounds_sample = p->bounds_sample;
        bp.idx_own_speed = p->idx[3]; bp.idx_int_speed = p->idx[9];
        bp.min_vel1 = p->dyn_limits[0][0]; bp.max_vel1 = p->dyn_limits[0][1]; bp.min_vel2 = p->dyn_limits[1][0]; bp.max_vel2 = p->dyn_limits[1][1];
        size_t count = n;
        std::string kernels;
        for (int j = 0; j < p->max_track_attempts && count > 0; j++) {
            const int cur = j & 1;
            const uint64_t seed = p->seed + (uint64_t)j;
            const uint64_t *ind = j ? d_gidx[cur] : nullptr;
            // geometry draw (sample.m:29-77)
            bp.seed = seed; bp.first_index = p->first_index; bp.n = (int64_t)count;
            EmgpuBnRun B;
            fill_bn(ctx, ug, g, &bp, B);
            B.out_val = d_val; B.ld = (int64_t)count; B.indices = ind;
            const char *name = "";
            hipError_t e = emgpu::launch_bn(ug.cp.plan, B, ctx->stream, &name);
            if (e != hipSuccess) throw Error(EMGPU_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
            kernels = name;
            // createEncounter.m:21-49
            EmgpuTGeoRun G;
            memset(&G, 0, sizeof G);
            G.n = (int64_t)count; G.val = d_val; G.geo = d_geo; G.model_of = d_mo;
            for (int k = 0; k < 12; k++) G.idx[k] = p->idx[k] - 1;
            e = emgpu::launch_terminal_geo(G, ctx->stream);
            if (e != hipSuccess) throw Error(EMGPU_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
            // PropagateTrajectory x 4 (createEncounter.m:52-72)
            EmgpuTermRun A;
            memset(&A, 0, sizeof A);
            A.seed = seed; A.first_index = p->first_index; A.n = (int64_t)count; A.geo = d_geo; A.model_of = d_mo; A.thr_base = ctx->d_thr_base;
            A.tmax_s = p->tmax_s; A.max_resample = p->max_resample; A.cap = cap;
            memcpy(A.dl, p->dyn_limits, sizeof A.dl);
            A.traj = d_out; A.rows = d_rows; A.status = ctx->d_status; A.queue = ctx->d_queue; A.indices = ind; A.quiet = 1;
            e = emgpu::launch_terminal_propagate(first->cp.plan, A, ctx->stream, &name);
            if (e != hipSuccess) throw Error(EMGPU_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
            kernels += std::string(" + ") + name;
            if (p->flags & EMGPU_FLAG_LOCAL_SMOOTH) {   // createEncounter.m:88-89 (stand-in): the filters read the smoothed speed and altitude
                e = emgpu::launch_terminal_smooth(d_out, d_rows, 2 * (int64_t)count, cap, ctx->stream);
                if (e != hipSuccess) throw Error(EMGPU_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
                kernels += " + k_terminal_smooth";
            }
            // the filters (track.m:62-145)
            EmgpuTFilterRun F;
            memset(&F, 0, sizeof F);
            F.n = (int64_t)count; F.tracks = d_out; F.rows = d_rows; F.cap = cap; F.geo = d_geo; F.val = d_val; F.n_i = (int32_t)ni;
            memcpy(F.dl, p->dyn_limits, sizeof F.dl);
            for (int a = 0; a < 2; a++) { F.max_cum_turn[a] = p->max_cum_turn_deg[a]; F.pitch[a] = p->pitch_deg[a]; }
            F.min_enc_time_s = p->min_enc_time_s; F.thres_dist_ft = p->thres_dist_ft; F.thres_alt_low_ft = p->thres_alt_low_ft; F.thres_vertrate_ft_s = p->thres_vertrate_ft_s;
            F.slot = j ? d_slot[cur] : nullptr; F.accepted = d_acc;
            F.sample = d_sample; F.traj = d_traj; F.cap2 = cap2; F.len = d_len; F.meta = d_meta; F.attempts = d_att;
            F.attempt_no = j + 1; F.last_round = (j + 1 == p->max_track_attempts);
            e = emgpu::launch_terminal_filter(F, ctx->stream, &name);
            if (e != hipSuccess) throw Error(EMGPU_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
            ctx->last_kernel = kernels + " + " + name;
            HIP_OK(hipMemsetAsync(d_count, 0, 4, ctx->stream));
            e = emgpu::launch_compact_rejected((int64_t)count, p->first_index, d_acc, ind, j ? d_slot[cur] : nullptr, d_gidx[cur ^ 1], d_slot[cur ^ 1], d_count, ctx->stream);
            if (e != hipSuccess) throw Error(EMGPU_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
            uint32_t hc = 0;
            HIP_OK(hipMemcpyAsync(&hc, d_count, 4, hipMemcpyDeviceToHost, ctx->stream));
            HIP_OK(hipStreamSynchronize(ctx->stream));
            count = hc;
        }
        rc = emgpu_ctx_sync(ctx);   // the geometry draw's own rejection cap
        std::string msg = g_err;
        auto back = [&](void *dst, const void *src, size_t bytes) { if (dst && bytes) HIP_OK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream)); };
        back(sample, d_sample, n * ni * 8); back(traj, d_traj, n * 2 * (size_t)cap2 * 6 * 8); back(len, d_len, n * 2 * 4);
        back(meta, d_meta, n * 4 * 8); back(attempts, d_att, n * 4);
        HIP_OK(hipStreamSynchronize(ctx->stream));
        if (rc != EMGPU_OK) g_err = msg;
        else if (count > 0) rc = fail(EMGPU_ERR_REJECT_CAP, "terminal track: " + std::to_string(count) + " encounters were still rejected after max_track_attempts");
    } catch (...) {
        (void)hipStreamSynchronize(ctx->stream);
        throw;
    }
    (void)hipStreamSynchronize(ctx->stream);
    return rc;
    EMGPU_CATCH
}

static int track_params_ok(const emgpu_track_params *p) {
    if (p->n < 0 || p->T < 1) return fail(EMGPU_ERR_ARG, "bad n / T");
    return EMGPU_OK;
}

int emgpu_sample2track_device(emgpu_ctx *ctx, const emgpu_track_params *p, const float *alt0, const float *speed0,
                              const float *dyn_val, double *xyz, uint8_t *flags, double *speed_minmax) {
    EMGPU_TRY
    if (!ctx || !p || !alt0 || !speed0 || !dyn_val) return fail(EMGPU_ERR_ARG, "null argument");
    if (int rc = track_params_ok(p)) return rc;
    if (p->nd < 1 || p->slot_vertrate < 0 || p->slot_vertrate >= p->nd || p->slot_acc < 0 || p->slot_acc >= p->nd ||
        p->slot_turnrate < 0 || p->slot_turnrate >= p->nd)
        return fail(EMGPU_ERR_ARG, "bad dense-trace rows");
    CTX_LOCK(ctx);
    HIP_OK(hipSetDevice(ctx->device));
    EmgpuTrackRun A{};
    A.n = p->n; A.T = p->T;
    A.ur_speed = p->ur_speed; A.ur_vertrate = p->ur_vertrate; A.ur_heading = p->ur_heading;
    A.min_speed = p->min_speed; A.max_speed = p->max_speed;
    A.alt0_f = alt0; A.speed0_f = speed0; A.dyn_val = dyn_val;
    A.nd = p->nd; A.s_vr = p->slot_vertrate; A.s_acc = p->slot_acc; A.s_tr = p->slot_turnrate;
    A.xyz = xyz; A.flags = flags; A.vmm = speed_minmax;
    const char *name = "";
    const hipError_t e = emgpu::launch_sample2track(A, true, ctx->stream, &name);
    ctx->last_kernel = name;
    if (e != hipSuccess) return fail(EMGPU_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return EMGPU_OK;
    EMGPU_CATCH
}

int emgpu_sample2track_host(emgpu_ctx *ctx, const emgpu_track_params *p, const double *alt0, const double *speed0,
                            const double *updates, double *xyz, uint8_t *flags, double *speed_minmax) {
    EMGPU_TRY
    if (!ctx || !p || !alt0 || !speed0 || !updates) return fail(EMGPU_ERR_ARG, "null argument");
    if (int rc = track_params_ok(p)) return rc;
    CTX_LOCK(ctx);
    HIP_OK(hipSetDevice(ctx->device));
    const size_t n = (size_t)p->n, T = (size_t)p->T;
    if (n == 0) return EMGPU_OK;
    // [n][T][3] -> [T][3][n] so that a wave reads 64 consecutive doubles
    std::vector<double> planar(T * 3 * n), hx;
    for (size_t i = 0; i < n; i++)
        for (size_t t = 0; t < T; t++)
            for (size_t c = 0; c < 3; c++) planar[(t * 3 + c) * n + i] = updates[(i * T + t) * 3 + c];
    double *d_in = nullptr, *d_xyz = nullptr, *d_vmm = nullptr; uint8_t *d_fl = nullptr;
    int rc = EMGPU_OK;
    try {
        HIP_OK(hipMalloc((void **)&d_in, (planar.size() + 2 * n) * sizeof(double)));
        HIP_OK(hipMalloc((void **)&d_xyz, (T + 1) * 3 * n * sizeof(double)));
        HIP_OK(hipMalloc((void **)&d_vmm, 2 * n * sizeof(double)));
        HIP_OK(hipMalloc((void **)&d_fl, n));
        HIP_OK(hipMemcpyAsync(d_in, planar.data(), planar.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIP_OK(hipMemcpyAsync(d_in + planar.size(), alt0, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIP_OK(hipMemcpyAsync(d_in + planar.size() + n, speed0, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        EmgpuTrackRun A{};
        A.n = p->n; A.T = p->T;
        A.ur_speed = p->ur_speed; A.ur_vertrate = p->ur_vertrate; A.ur_heading = p->ur_heading;
        A.min_speed = p->min_speed; A.max_speed = p->max_speed;
        A.upd = d_in; A.alt0_d = d_in + planar.size(); A.speed0_d = d_in + planar.size() + n;
        A.xyz = d_xyz; A.flags = d_fl; A.vmm = d_vmm;
        const char *name = "";
        const hipError_t e = emgpu::launch_sample2track(A, false, ctx->stream, &name);
        ctx->last_kernel = name;
        if (e != hipSuccess) rc = fail(EMGPU_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
        if (rc == EMGPU_OK) {
            if (xyz) {
                hx.resize((T + 1) * 3 * n);
                HIP_OK(hipMemcpyAsync(hx.data(), d_xyz, hx.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            }
            if (flags) HIP_OK(hipMemcpyAsync(flags, d_fl, n, hipMemcpyDeviceToHost, ctx->stream));
            std::vector<double> hv;
            if (speed_minmax) {
                hv.resize(2 * n);
                HIP_OK(hipMemcpyAsync(hv.data(), d_vmm, 2 * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            }
            rc = emgpu_ctx_sync(ctx);
            if (rc == EMGPU_OK && xyz)
                for (size_t i = 0; i < n; i++)
                    for (size_t t = 0; t <= T; t++)
                        for (size_t c = 0; c < 3; c++) xyz[(i * (T + 1) + t) * 3 + c] = hx[(t * 3 + c) * n + i];
            if (rc == EMGPU_OK && speed_minmax)
                for (size_t i = 0; i < n; i++) { speed_minmax[2 * i] = hv[i]; speed_minmax[2 * i + 1] = hv[n + i]; }
        }
    } catch (...) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(d_in); (void)hipFree(d_xyz); (void)hipFree(d_vmm); (void)hipFree(d_fl);
        throw;
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_in); (void)hipFree(d_xyz); (void)hipFree(d_vmm); (void)hipFree(d_fl);
    return rc;
    EMGPU_CATCH
}

static emgpu::UncorTrackVars track_vars(const emgpu_utrack_params *p) {
    emgpu::UncorTrackVars tv;
    tv.idxG = p->idx_G; tv.idxA = p->idx_A; tv.idxL = p->idx_L; tv.idxV = p->idx_v; tv.idxDV = p->idx_dv; tv.idxDH = p->idx_dh; tv.idxDPsi = p->idx_dpsi;
    tv.is_rotorcraft = p->is_rotorcraft != 0;
    return tv;
}

int emgpu_uncor_dynamic_limits(const emgpu_model *h, const emgpu_utrack_params *vars, const double *initial,
                               double up_min, double up_max, double v_min, double v_max, double out[3]) {
    EMGPU_TRY
    if (!h || !vars || !initial || !out) return fail(EMGPU_ERR_ARG, "null argument");
    const emgpu::UncorLimits L = emgpu::build_uncor_limits(h->m, track_vars(vars));
    const double *e = L.table.data();
    if (L.ordered) {
        auto disc = [](double x, const double *cut, int n) { return (int)emgpu_discretize_bayes(x, cut, n); };
        const int dG = (int)initial[vars->idx_G - 1], dA = (int)initial[vars->idx_A - 1];
        int l0, l1, b0, b1;
        if (L.discL) l0 = l1 = (int)initial[vars->idx_L - 1];
        else { l0 = disc(up_min, L.cutL, L.ncL); l1 = disc(up_max, L.cutL, L.ncL); }
        b0 = disc(v_min * 0.592484, L.cutV, L.ncV); b1 = disc(v_max * 0.592484, L.cutV, L.ncV);
        if (dG < 1 || dG > L.rG || dA < 1 || dA > L.rA || l0 < 1 || l1 > L.rL || l0 > l1 || b0 > b1) return fail(EMGPU_ERR_ARG, "initial values outside the model's bins");
        e += ((((((size_t)(dG - 1) * L.rA + (size_t)(dA - 1)) * L.rL + (size_t)(l0 - 1)) * L.rL + (size_t)(l1 - 1)) * L.rV + (size_t)(b0 - 1)) * L.rV + (size_t)(b1 - 1)) * 3;
    }
    out[0] = e[0]; out[1] = e[1]; out[2] = e[2];
    return EMGPU_OK;
    EMGPU_CATCH
}

// The rounds of UncorEncounterModel.m:419-471 (see the header).  d_*: device outputs (any may be null).
static int track_uncor_rounds(emgpu_ctx *ctx, const emgpu_model *h, const emgpu_utrack_params *p, double *d_tracks, double *d_limits, int32_t *d_attempts) {
    const Model &m = h->m;
    if (p->n < 0 || p->sample_time < 1 || p->sample_time > 65535) throw Error(EMGPU_ERR_ARG, "n < 0 or sample_time outside 1..65535");
    if (p->max_track_attempts < 1 || p->max_attempts < 1 || p->record_stride < 1 || (10 * p->sample_time) % p->record_stride)
        throw Error(EMGPU_ERR_ARG, "max_track_attempts / max_attempts must be >= 1 and record_stride must divide 10 * sample_time");
    const emgpu::UncorTrackVars tv = track_vars(p);
    const std::array<uint64_t, 3> lkey = {m.uid, m.version,
                                          (uint64_t)(uint8_t)tv.idxG | ((uint64_t)(uint8_t)tv.idxA << 8) | ((uint64_t)(uint8_t)tv.idxL << 16) | ((uint64_t)(uint8_t)tv.idxV << 24) |
                                              ((uint64_t)(uint8_t)tv.idxDV << 32) | ((uint64_t)(uint8_t)tv.idxDH << 40) | ((uint64_t)(uint8_t)tv.idxDPsi << 48) | ((uint64_t)tv.is_rotorcraft << 56)};
    if (ctx->limits_cache.size() > 32 && !ctx->limits_cache.count(lkey)) ctx->limits_cache.clear();
    auto lit = ctx->limits_cache.find(lkey);
    if (lit == ctx->limits_cache.end()) lit = ctx->limits_cache.emplace(lkey, emgpu::build_uncor_limits(m, tv)).first;
    const emgpu::UncorLimits &L = lit->second;
    if (m.n_dyn() < 3) throw Error(EMGPU_ERR_ARG, "dynvar:empty: the model needs dynamic variables for acceleration, vertical rate and turn rate");
    auto row_of = [&](int idx) {   // row of the temporal map == row of the dense trace
        for (size_t k = 0; k < m.temporal_map.size(); k++) if (m.temporal_map[k][0] == idx) return (int)k;
        throw Error(EMGPU_ERR_ARG, "dynvar:empty: \\dot v, \\dot h and \\dot \\psi must be dynamic variables");
    };
    const int sDV = row_of(p->idx_dv), sDH = row_of(p->idx_dh), sDPsi = row_of(p->idx_dpsi);
    const size_t n = (size_t)p->n, ni = (size_t)m.n_initial, nd = (size_t)m.n_dyn(), T = (size_t)p->sample_time, G4 = (T + 3) / 4;
    if (n == 0) return EMGPU_OK;
    size_t slot = 0;
    auto dalloc = [&](size_t bytes) { return ctx_scratch(ctx, slot++, bytes ? bytes : 1); };   // kept by the ctx between calls
    int rc = EMGPU_OK;
    try {
        float *d_iv = (float *)dalloc(ni * n * 4), *d_dv = (float *)dalloc(G4 * nd * n * 16);
        double *d_lim = (double *)dalloc(L.table.size() * 8);
        uint8_t *d_acc = (uint8_t *)dalloc(n);
        uint64_t *d_gidx[2] = {(uint64_t *)dalloc(n * 8), (uint64_t *)dalloc(n * 8)};
        int64_t *d_slot[2] = {(int64_t *)dalloc(n * 8), (int64_t *)dalloc(n * 8)};
        uint32_t *d_count = (uint32_t *)dalloc(4 * emgpu::compact_scratch_words((int64_t)n));
        HIP_OK(hipMemcpyAsync(d_lim, L.table.data(), L.table.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        HIP_OK(hipStreamSynchronize(ctx->stream)); // L.table is a local
        Uploaded &u = get_uploaded(ctx, h);
        EmgpuUTrackRun R;
        memset(&R, 0, sizeof R);
        R.T = (int32_t)T; R.stride = p->record_stride; R.nd = (int32_t)nd; R.sDV = sDV; R.sDH = sDH; R.sDPsi = sDPsi;
        {   // UncorEncounterModel.m:397-414
            const std::vector<double> &bL = m.boundaries[p->idx_L - 1], &bV = m.boundaries[p->idx_v - 1], &bDH = m.boundaries[p->idx_dh - 1];
            R.min_alt = bL.empty() ? 0.0 : *std::min_element(bL.begin(), bL.end());
            R.max_alt = bL.empty() ? INFINITY : *std::max_element(bL.begin(), bL.end());
            R.dyn[0] = 1.7; R.dyn[1] = *std::max_element(bV.begin(), bV.end()) * 1.68780972222222;
            R.dyn[2] = *std::min_element(bDH.begin(), bDH.end()) / 60.0; R.dyn[3] = *std::max_element(bDH.begin(), bDH.end()) / 60.0;
            R.dyn[4] = 3.0 * (3.14159265358979323846 / 180.0); R.dyn[5] = 1000000.0;
        }
        R.ordered = L.ordered; R.rG = L.rG; R.rA = L.rA; R.rL = L.rL; R.rV = L.rV; R.ncL = L.ncL; R.ncV = L.ncV; R.discL = L.discL; R.discV = L.discV;
        memcpy(R.cutL, L.cutL, sizeof R.cutL); memcpy(R.cutV, L.cutV, sizeof R.cutV);
        R.lim = d_lim; R.tracks = d_tracks; R.S = (int64_t)(10 * T / (size_t)p->record_stride + 1); R.limits = d_limits;
        R.accepted = d_acc; R.attempts = d_attempts;
        size_t count = n;
        for (int j = 0; j < p->max_track_attempts && count > 0; j++) {
            const int cur = j & 1;
            emgpu_sample_params sp;
            memset(&sp, 0, sizeof sp);
            sp.seed = p->seed + (uint64_t)j;                                  // :428  seed = seed + 1
            sp.first_index = p->first_index; sp.n = (int64_t)count; sp.sample_time = p->sample_time;
            sp.flags = p->flags & EMGPU_FLAG_QUANTIZE500; sp.max_attempts = p->max_attempts;
            sp.idx_L = p->idx_L; sp.idx_v = p->idx_v; sp.idx_dh = p->idx_dh;
            sp.indices = j ? d_gidx[cur] : nullptr;
            EmgpuRun A;
            fill_run(ctx, u, m, &sp, A);
            A.init_val = d_iv; A.dyn_val = d_dv; A.ld = (int64_t)count;
            launch_dbn(ctx, u, A);                                            // :424  self.sample(1, sample_time, 'seed', seed)
            const std::string sampler = ctx->last_kernel;
            R.n = (int64_t)count; R.ld = (int64_t)count;
            auto row = [&](int idx) -> const float * { return idx > 0 ? d_iv + (size_t)(idx - 1) * count : nullptr; };
            R.iG = row(p->idx_G); R.iA = row(p->idx_A); R.iL = row(p->idx_L); R.iV = row(p->idx_v);
            R.iDV = row(p->idx_dv); R.iDH = row(p->idx_dh); R.iDPsi = row(p->idx_dpsi);
            R.dyn_val = d_dv; R.slot = j ? d_slot[cur] : nullptr;
            R.attempt_no = j + 1; R.last_round = (j + 1 == p->max_track_attempts);
            const char *name = "";
            hipError_t e = emgpu::launch_uncor_track(R, ctx->stream, &name);
            if (e != hipSuccess) throw Error(EMGPU_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
            ctx->last_kernel = sampler + " + " + name;
            HIP_OK(hipMemsetAsync(d_count, 0, 4, ctx->stream));
            e = emgpu::launch_compact_rejected((int64_t)count, p->first_index, d_acc, j ? d_gidx[cur] : nullptr, j ? d_slot[cur] : nullptr,
                                               d_gidx[cur ^ 1], d_slot[cur ^ 1], d_count, ctx->stream);
            if (e != hipSuccess) throw Error(EMGPU_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
            uint32_t hc = 0;
            HIP_OK(hipMemcpyAsync(&hc, d_count, 4, hipMemcpyDeviceToHost, ctx->stream));
            HIP_OK(hipStreamSynchronize(ctx->stream));
            count = hc;
        }
        rc = emgpu_ctx_sync(ctx);   // the sampler's own rejection cap
        if (rc == EMGPU_OK && count > 0) rc = fail(EMGPU_ERR_REJECT_CAP, "track: " + std::to_string(count) + " trajectories were still rejected after max_track_attempts");
    } catch (...) {
        (void)hipStreamSynchronize(ctx->stream);
        throw;
    }
    (void)hipStreamSynchronize(ctx->stream);
    return rc;
}

int emgpu_track_uncor_device(emgpu_ctx *ctx, const emgpu_model *h, const emgpu_utrack_params *p, double *tracks, double *limits, int32_t *attempts) {
    EMGPU_TRY
    if (!ctx || !h || !p) return fail(EMGPU_ERR_ARG, "null argument");
    CTX_LOCK(ctx);
    HIP_OK(hipSetDevice(ctx->device));
    return track_uncor_rounds(ctx, h, p, tracks, limits, attempts);
    EMGPU_CATCH
}

int emgpu_track_uncor_host(emgpu_ctx *ctx, const emgpu_model *h, const emgpu_utrack_params *p, double *tracks, double *limits, int32_t *attempts) {
    EMGPU_TRY
    if (!ctx || !h || !p) return fail(EMGPU_ERR_ARG, "null argument");
    CTX_LOCK(ctx);
    HIP_OK(hipSetDevice(ctx->device));
    const size_t n = (size_t)(p->n > 0 ? p->n : 0);
    if (p->record_stride < 1 || p->sample_time < 1) return fail(EMGPU_ERR_ARG, "bad record_stride / sample_time");
    const size_t S = (size_t)(10 * p->sample_time / p->record_stride + 1);
    double *dt = nullptr, *dl = nullptr; int32_t *da = nullptr;
    int rc;
    try {
        if (tracks) HIP_OK(hipMalloc((void **)&dt, n * S * 8 * sizeof(double) + 8));
        if (limits) HIP_OK(hipMalloc((void **)&dl, n * 3 * sizeof(double) + 8));
        HIP_OK(hipMalloc((void **)&da, n * 4 + 4));
        rc = track_uncor_rounds(ctx, h, p, dt, dl, da);
        if (rc == EMGPU_OK || rc == EMGPU_ERR_REJECT_CAP) {
            const std::string msg = g_err;
            if (tracks && n) HIP_OK(hipMemcpyAsync(tracks, dt, n * S * 8 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            if (limits && n) HIP_OK(hipMemcpyAsync(limits, dl, n * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            if (attempts && n) HIP_OK(hipMemcpyAsync(attempts, da, n * 4, hipMemcpyDeviceToHost, ctx->stream));
            HIP_OK(hipStreamSynchronize(ctx->stream));
            if (rc != EMGPU_OK) g_err = msg;
        }
    } catch (...) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(dt); (void)hipFree(dl); (void)hipFree(da);
        throw;
    }
    (void)hipFree(dt); (void)hipFree(dl); (void)hipFree(da);
    return rc;
    EMGPU_CATCH
}

int32_t emgpu_discretize_bayes(double x, const double *thresholds, int32_t n) {
    // discretize_bayes.m:17-21
    if (n <= 0 || !thresholds) return 1;
    if (x >= thresholds[n - 1]) return n + 1;
    for (int i = 0; i < n; i++)
        if (x < thresholds[i]) return i + 1;
    return n + 1;
}

int64_t emgpu_asub2ind(const int32_t *siz, const int32_t *x, int32_t n) {
    // asub2ind.m:13-14
    int64_t k = 1, ndx = 1;
    for (int i = 0; i < n; i++) { ndx += k * (int64_t)(x[i] - 1); k *= siz[i]; }
    return ndx;
}

} // extern "C"
