// emgpu_model.hpp -- host-side model IR (what em_read.m returns / EncounterModel.m holds) and the
// plan compiler.  No HIP in this header.
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "emgpu_plan.h"

namespace emgpu {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &what) : std::runtime_error(what), code(c) {}
};

struct CompiledPlan;
struct Model {
    // em_read.m:47-107 fields
    std::vector<std::string> labels_initial, labels_transition;
    int n_initial = 0, n_transition = 0;
    std::vector<uint8_t> G_initial, G_transition; // row-major [parent][child]
    std::vector<int> r_initial, r_transition;
    std::vector<int> order_initial, order_transition; // 1-based ids (bn_sort.m)
    std::vector<std::array<int, 2>> temporal_map;     // 1-based (var@t, var@t+1)
    // N{i}: r_i x q_i column-major.  N_transition is indexed by variable id (0..n_transition-1),
    // empty for nodes without a table (em_read.m:92).
    std::vector<std::vector<double>> N_initial, N_transition;
    std::vector<std::vector<double>> A_initial, A_transition; // alpha (dirichlet_*), same shapes
    std::vector<int64_t> q_initial, q_transition;
    std::vector<std::vector<double>> boundaries;
    std::vector<int> zero_bins; // 0 = none
    std::vector<double> resample_rates;
    std::vector<int> start; // 0 = unset
    uint64_t version = 1;   // bumped by every setter: invalidates uploaded plans
    uint64_t uid = 0;       // process-unique id (a freed model's address can be reused: never key a cache by pointer)
    // the plan compiled for `plan_version` of this model, when somebody has it already (load_bin restores it from the cache file:
    // the threshold search of compile_plan is most of what loading a model costs)
    mutable std::shared_ptr<const CompiledPlan> plan_cache;
    mutable uint64_t plan_version = 0;

    int n_dyn() const { return (int)temporal_map.size(); }
    bool is_dynvar_depend() const;
    int rows_initial(int v) const { return r_initial[v]; }
    int rows_transition(int v) const { return r_transition[v]; }

    void finalize();                         // orders, q's, default alphas, start
    void set_prior(int kind, double value);  // bn_dirichlet_prior.m:18-37
    void set_transition_stay_prior(double p); // setTransitionPriors.m:12-33
    double start_log_weight() const;          // log P(preset values) under the model: the importance weight of `start`
};

Model *load_txt(const char *path, const int32_t *idx_zero, int n_idx, bool overwrite);
// Binary model cache (em_read.m:47-107 is what it saves re-doing): the parsed model -- every field em_read returns, the
// priors and `start` -- plus its compiled plan, in one file tagged with the library's source hash.  load_bin refuses (EMGPU_ERR_PARSE) a
// file written by other sources: the caller then reads the .txt again.
void save_bin(const Model &m, const char *path, const char *src_hash);
Model *load_bin(const char *path, const char *src_hash);
// the model's plan: the cached one when it is current, else compiled now (and kept)
std::shared_ptr<const CompiledPlan> plan_of(const Model &m);
std::vector<int> bn_sort(const std::vector<uint8_t> &G, int n);
std::vector<int> extract_zero_bins(const std::vector<std::vector<double>> &b);

// uniform32: the ONLY definition of the uniform on the host side.
inline double uniform32(uint32_t x) {
    if (x > 0xFFFFFFFEu) x = 0xFFFFFFFEu;
    return ((double)x + 0.5) * (1.0 / 4294967296.0);
}

struct CompiledPlan {
    EmgpuPlan plan{};             // thr / bnd pointers are filled at upload time
    std::vector<uint32_t> thr;
    std::vector<uint32_t> cthr;   // compacted tables of the dynamic variables
    std::vector<uint32_t> pthr;   // ... padded to 4 / 8 words per column (EmgpuPlan::d_pw)
    std::vector<double> bnd;
    std::vector<int> pos_of_var;  // variable id (0-based) -> topological position
};
// Throws Error(EMGPU_ERR_UNSUPPORTED / EMGPU_ERR_PRESET / ...) when the model cannot be planned.
CompiledPlan compile_plan(const Model &m);

// @UncorEncounterModel/getDynamicLimits.m as a table (emgpu_limits.cpp)
struct UncorTrackVars {
    int idxG = 0, idxA = 0, idxL = 0, idxV = 0, idxDV = 0, idxDH = 0, idxDPsi = 0; // 1-based, 0 = absent (UncorEncounterModel.m:385-391)
    bool is_rotorcraft = false;                                                     // :181-185
};
struct UncorLimits {
    bool ordered = false, discL = false, discV = false;
    int rG = 0, rA = 0, rL = 0, rV = 0, ncL = 0, ncV = 0;
    double cutL[16] = {0}, cutV[16] = {0};
    std::vector<double> table; // ordered: [rG][rA][rL][rL][rV][rV][3], else [3]: minVel_ft_s maxVel_ft_s maxVertRate_ft_s
};
UncorLimits build_uncor_limits(const Model &m, const UncorTrackVars &tv);

// log P(bin | column) of the initial network as the kernels index it: node after node BY TOPOLOGICAL POSITION, column after column (the
// column number of asub2ind.m:13-14, i.e. the plan's strides), r entries each; off[p] = the first entry of position p.  P is the
// column-normalised N + alpha; an all-zero column draws bin 1 with certainty (select_random.m:17-20: sthres = 0).  The table behind the
// per-sample log-weights of a start grid (InitStartTerminal.m:57-90).
std::vector<double> initial_log_prob(const Model &m, uint32_t off[EMGPU_MAX_NI]);

// Quantile thresholds of one CPT column (r weights): out[r-1].
void column_thresholds(const double *w, int r, uint32_t *out);
uint32_t bernoulli_threshold(double rate);

} // namespace emgpu
