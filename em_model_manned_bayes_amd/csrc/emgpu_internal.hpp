// emgpu_internal.hpp -- what the translation units behind include/emgpu.h share: the handle structs, the error plumbing and the
// ctx's device scratch.  Not installed; nothing outside csrc/ includes it.
#pragma once
#include <hip/hip_runtime.h>

#include <array>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../include/emgpu.h"
#include "emgpu_launch.h"
#include "emgpu_model.hpp"

using emgpu::CompiledPlan;
using emgpu::Error;
using emgpu::Model;

struct emgpu_model {
    Model m;
};

namespace emgpu_detail {
std::string &last_error();                       // the calling thread's message (emgpu_last_error)
int fail(int code, const std::string &msg);      // records msg, returns code
} // namespace emgpu_detail
using emgpu_detail::fail;
#define g_err (emgpu_detail::last_error())

#define EMGPU_TRY try {
#define EMGPU_CATCH                                                      \
    }                                                                    \
    catch (const Error &e) { return fail(e.code, e.what()); }            \
    catch (const std::bad_alloc &) { return fail(EMGPU_ERR_ARG, "out of host memory"); } \
    catch (const std::exception &e) { return fail(EMGPU_ERR_ARG, e.what()); }

#define HIP_OK(expr)                                                                                  \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) throw Error(EMGPU_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

struct Uploaded {
    uint64_t version = 0;
    uint64_t last_use = 0;
    CompiledPlan cp;
    uint32_t *d_thr = nullptr;
    uint32_t *d_cthr = nullptr;
    uint32_t *d_pthr = nullptr;
    double *d_bnd = nullptr;
    void *d_planf = nullptr; // the plan itself (+ k_uncor_fast's resample thresholds) for launches that serve several models
    double *d_logp = nullptr; // log P of the initial network (emgpu::initial_log_prob), uploaded when a call first asks for log-weights
    uint32_t lp_off[EMGPU_MAX_NI] = {0};
    void free_tables() {
        (void)hipFree(d_thr); (void)hipFree(d_cthr); (void)hipFree(d_pthr); (void)hipFree(d_bnd); (void)hipFree(d_planf); (void)hipFree(d_logp);
        d_thr = d_cthr = d_pthr = nullptr; d_bnd = nullptr; d_planf = nullptr; d_logp = nullptr;
    }
};

struct emgpu_ctx {
    std::recursive_mutex mu; // serialises calls on this ctx (the *_host entry points re-enter through *_device)
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    uint32_t *d_status = nullptr;
    uint32_t *d_queue = nullptr;  // k_terminal_propagate: the launch's track queue (one word, zeroed by the launcher)
    EmgpuPresets *d_presets = nullptr;   // the start grid / log-weight block of the last DBN call that had one
    uint32_t *h_status = nullptr; // pinned
    std::map<uint64_t, Uploaded> cache; // by Model::uid
    uint64_t use_clock = 0;
    std::string last_kernel;
    int32_t last_launches = 0;
    double *d_layers = nullptr;
    size_t d_layers_cap = 0;
    const uint32_t **d_thr_base = nullptr; // terminal propagation: per-model table pointers
    size_t d_thr_base_cap = 0;
    // Side streams for the blocks of a mixed batch (created on first use): independent launches that share the ctx stream's
    // ordering at both ends, so that one block's tail runs under the next block's head instead of in front of it.
    static constexpr int kSide = 3;
    // Device scratch of the round drivers (UncorEncounterModel.track / CorTerminalModel.track), kept between calls and grown on demand:
    // a fresh hipMalloc + hipFree of several gigabytes per call cost tens of milliseconds, at random (measured: 29 vs 127 ms per call)
    struct Scratch { void *p = nullptr; size_t cap = 0; };
    std::vector<Scratch> scratch;
    // getDynamicLimits.m as a table, per (model uid, model version, the track variables): building it walks N_initial{v} and
    // N_initial{\dot h} over every (G, A, L range, v range) -- 4 ms on the host for uncor_1200code_v2p1, per call before it was kept
    std::map<std::array<uint64_t, 3>, emgpu::UncorLimits> limits_cache;
    hipStream_t side[kSide] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[kSide] = {nullptr, nullptr, nullptr};
    // ---- the trace pool (emgpu_trace_alloc / emgpu_trace_free, emgpu_host.cpp): device blocks whose PLACEMENT has been measured.  A block
    // given back stays here and the next request it fits takes it without a new probe; emgpu_ctx_trim / emgpu_ctx_free release them.
    struct TraceBlock { void *p = nullptr; size_t bytes = 0; float ms = 0.f; bool probed = false; };
    std::vector<TraceBlock> trace_pool;
    std::set<void *> device_blocks;   // emgpu_device_alloc: blocks a caller holds (released with the ctx at the latest)
    // ---- the pipeline of the host-pointer entry points: a copy stream beside the launch stream, two chunk buffers on the device (blocks of
    // the trace pool), two pinned staging buffers, and pinned blocks handed to callers (emgpu_host_alloc)
    hipStream_t copy_stream = nullptr;
    TraceBlock chunk_buf[2];
    void *h_stage[2] = {nullptr, nullptr};
    size_t h_stage_cap = 0;
    uint64_t *h_total = nullptr;   // pinned: rows of a chunk's packed event lists
    struct HostBlock { void *p = nullptr; size_t bytes = 0; bool in_use = false; };
    std::vector<HostBlock> host_pool;
    emgpu_host_stats_t host_stats{};   // phases of the last emgpu_sample_dbn_host call
};

// emgpu_capi.cpp
void *ctx_scratch(emgpu_ctx *ctx, size_t slot, size_t bytes);   // slot-th scratch buffer of the ctx, at least `bytes` long
// emgpu_host.cpp
void ctx_release_host_side(emgpu_ctx *ctx, bool everything);    // trim (false: pools and staging) / free (true: streams and events too)

#define CTX_LOCK(ctx) std::lock_guard<std::recursive_mutex> _ctx_lock((ctx)->mu)

