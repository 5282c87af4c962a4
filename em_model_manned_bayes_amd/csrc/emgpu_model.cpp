// emgpu_model.cpp -- model .txt loader (replaces em_read.m), priors (bn_dirichlet_prior.m,
// setTransitionPriors.m), topological sort (bn_sort.m) and the plan compiler that turns CPT
// columns into u32 quantile thresholds for the HIP kernels.  Host only.
#include "emgpu_model.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <queue>

#include "../../include/emgpu.h"

namespace emgpu {

// ---------------------------------------------------------------------------------------------
// bn_sort.m:17-20  toposort(digraph(G),'Order','stable').  'stable' is taken to mean the
// lexicographically smallest topological order (Kahn's algorithm, lowest ready index first);
// identity for every upper-triangular G.  MATLAB is not available to confirm.
// ---------------------------------------------------------------------------------------------
std::vector<int> bn_sort(const std::vector<uint8_t> &G, int n) {
    std::vector<int> indeg(n, 0), order;
    for (int p = 0; p < n; p++)
        for (int c = 0; c < n; c++)
            if (G[(size_t)p * n + c]) indeg[c]++;
    std::priority_queue<int, std::vector<int>, std::greater<int>> ready;
    for (int i = 0; i < n; i++)
        if (indeg[i] == 0) ready.push(i);
    while (!ready.empty()) {
        int i = ready.top();
        ready.pop();
        order.push_back(i + 1);
        for (int c = 0; c < n; c++)
            if (G[(size_t)i * n + c] && --indeg[c] == 0) ready.push(c);
    }
    if ((int)order.size() != n) throw Error(EMGPU_ERR_SORT, "Network could not be hierarchically sorted");
    return order;
}

// em_read.m:143-156
std::vector<int> extract_zero_bins(const std::vector<std::vector<double>> &b) {
    std::vector<int> z(b.size(), 0);
    for (size_t i = 0; i < b.size(); i++)
        if (b[i].size() > 2)
            for (size_t j = 1; j < b[i].size(); j++)
                if (b[i][j - 1] < 0 && b[i][j] > 0) z[i] = (int)j;
    return z;
}

// em_read.m:158-177
static std::vector<std::array<int, 2>> extract_temporal_map(const std::vector<std::string> &labels) {
    std::vector<std::array<int, 2>> tm;
    for (size_t ii = 0; ii < labels.size(); ii++) {
        size_t t = labels[ii].find("(t)");
        if (t == std::string::npos) continue;
        std::string base = labels[ii].substr(0, t + 1);
        for (const char *suffix : {"t+1)", "t-1)"}) {
            std::string pat = base + suffix;
            for (size_t k = 0; k < labels.size(); k++)
                if (labels[k].find(pat) != std::string::npos) tm.push_back({(int)ii + 1, (int)k + 1});
        }
    }
    return tm;
}

static std::string rtrim(std::string s) {
    while (!s.empty() && (s.back() == ' ' || s.back() == '\t')) s.pop_back();
    return s;
}
static std::string trim(std::string s) {
    s = rtrim(s);
    size_t i = 0;
    while (i < s.size() && (s[i] == ' ' || s[i] == '\t')) i++;
    return s.substr(i);
}

// One number of a model file, without strtod: the files hold counts (integers of up to ten digits), boundaries and rates (a few
// decimals).  Digits are gathered into a 64-bit integer; with at most 19 significant digits, a value below 2^53 and a power of ten up to
// 10^22 the result is ONE correctly rounded multiply or divide of two exact doubles (Clinger's fast path) -- the same double strtod
// returns.  Anything else (longer mantissas, big exponents, inf / nan, hex) goes to strtod.  0.5 MB of counts: 2.5 ms instead of 11.
static bool scan_number_fast(const char *&p, double &out) {
    static const double kPow10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    const char *q = p;
    bool neg = false;
    if (*q == '-' || *q == '+') { neg = *q == '-'; q++; }
    uint64_t mant = 0;
    int digits = 0, exp10 = 0;
    bool any = false;
    while (*q >= '0' && *q <= '9') { any = true; if (mant || *q != '0') { if (++digits > 19) return false; mant = mant * 10 + (uint64_t)(*q - '0'); } q++; }
    if (*q == '.') {
        q++;
        while (*q >= '0' && *q <= '9') { any = true; if (mant || *q != '0') { if (++digits > 19) return false; mant = mant * 10 + (uint64_t)(*q - '0'); } exp10--; q++; }
    }
    if (!any) return false;
    if (*q == 'e' || *q == 'E') {
        const char *e = q + 1;
        bool eneg = false;
        if (*e == '-' || *e == '+') { eneg = *e == '-'; e++; }
        if (!(*e >= '0' && *e <= '9')) return false;
        int ev = 0;
        while (*e >= '0' && *e <= '9') { if (ev > 10000) return false; ev = ev * 10 + (*e - '0'); e++; }
        exp10 += eneg ? -ev : ev;
        q = e;
    }
    // what may follow a number in these files: a separator or the end of the line
    if (!(*q == 0 || *q == ' ' || *q == '\t' || *q == ',' || *q == '\r' || *q == '\n')) return false;
    if (mant > (1ull << 53) || exp10 > 22 || exp10 < -22) return false;
    double v = (double)mant;
    v = exp10 >= 0 ? v * kPow10[exp10] : v / kPow10[-exp10];
    out = neg ? -v : v;
    p = q;
    return true;
}

// textscan(line,'%f','Delimiter',' ')
static void scan_numbers(const std::string &line, std::vector<double> &out) {
    const char *p = line.c_str();
    char *end;
    for (;;) {
        while (*p == ' ' || *p == '\t' || *p == ',') p++;
        if (!*p) break;
        double v;
        if (!scan_number_fast(p, v)) {
            v = strtod(p, &end);
            if (end == p) throw Error(EMGPU_ERR_PARSE, std::string("cannot parse number near '") + std::string(p).substr(0, 16) + "'");
            p = end;
        }
        out.push_back(v);
    }
}

static std::vector<std::string> split_labels(const std::string &line) { // strtrim(strsplit(line, ','))
    std::vector<std::string> out;
    size_t pos = 0;
    for (;;) {
        size_t c = line.find(',', pos);
        out.push_back(trim(line.substr(pos, c == std::string::npos ? std::string::npos : c - pos)));
        if (c == std::string::npos) break;
        pos = c + 1;
    }
    return out;
}

static int64_t parent_count(const std::vector<uint8_t> &G, int n, const std::vector<int> &r, int child) {
    int64_t q = 1; // getdims, em_read.m:200-206
    for (int p = 0; p < n; p++)
        if (G[(size_t)p * n + child]) q *= r[p];
    return q;
}

bool Model::is_dynvar_depend() const {
    for (auto &a : temporal_map)
        for (auto &b : temporal_map)
            if (G_transition[(size_t)(a[1] - 1) * n_transition + (b[1] - 1)]) return true;
    return false;
}

void Model::set_prior(int kind, double value) {
    auto fill = [&](std::vector<std::vector<double>> &A, const std::vector<std::vector<double>> &N, const std::vector<int> &r) {
        A.resize(N.size());
        for (size_t i = 0; i < N.size(); i++) {
            A[i].assign(N[i].size(), 0.0);
            if (N[i].empty()) continue;
            double q = (double)N[i].size() / r[i];
            double p = kind == 1 ? 1.0 / ((double)r[i] * q) : value; // bn_dirichlet_prior.m:22-25 / :31-35
            std::fill(A[i].begin(), A[i].end(), p);
        }
    };
    fill(A_initial, N_initial, r_initial);
    if (n_transition > 0) fill(A_transition, N_transition, r_transition);
    version++;
}

// Importance weight of EncounterModel.start (UncorEncounterModel.m:204 `start`, RUN_uncor.m:43-48; InitStartTerminal.m grids):
// a preset node is legal only when all its parents are preset (bn_sample.m:45-47), so the probability of the forced values
// under the model is the same for every sample of a call: sum over preset nodes of log P(x_i = start_i | preset parents),
// P from N + alpha like select_random (an all-zero column selects bin 1 with probability 1, select_random.m:17-20).
double Model::start_log_weight() const {
    double lw = 0.0;
    for (int v = 0; v < n_initial; v++) {
        if (start.empty() || start[(size_t)v] == 0) continue;
        int64_t col = 0, stride = 1;
        for (int p = 0; p < n_initial; p++) {                  // asub2ind.m:13-14 over the parents in ascending index
            if (!G_initial[(size_t)p * n_initial + v]) continue;
            if (start[(size_t)p] == 0) throw Error(EMGPU_ERR_PRESET, "Attempt to preset a dependent variable"); // bn_sample.m:45-47
            col += stride * (start[(size_t)p] - 1);
            stride *= r_initial[(size_t)p];
        }
        const int r = r_initial[(size_t)v];
        double tot = 0.0, w = 0.0;
        for (int k = 0; k < r; k++) {
            const double x = N_initial[(size_t)v][(size_t)(col * r + k)] + A_initial[(size_t)v][(size_t)(col * r + k)];
            tot += x;
            if (k == start[(size_t)v] - 1) w = x;
        }
        if (tot > 0) lw += std::log(w / tot);
        else lw += (start[(size_t)v] == 1) ? 0.0 : -INFINITY;
    }
    return lw;
}

std::vector<double> initial_log_prob(const Model &m, uint32_t off[EMGPU_MAX_NI]) {
    std::vector<double> lp;
    for (int p = 0; p < m.n_initial && p < EMGPU_MAX_NI; p++) {
        const int v = m.order_initial[(size_t)p] - 1, r = m.r_initial[(size_t)v];
        off[p] = (uint32_t)lp.size();
        const std::vector<double> &N = m.N_initial[(size_t)v], &A = m.A_initial[(size_t)v];
        const int64_t q = m.q_initial[(size_t)v];
        for (int64_t c = 0; c < q; c++) {
            double tot = 0.0;
            for (int k = 0; k < r; k++) tot += N[(size_t)(c * r + k)] + A[(size_t)(c * r + k)];
            for (int k = 0; k < r; k++) {
                const double w = N[(size_t)(c * r + k)] + A[(size_t)(c * r + k)];
                lp.push_back(tot > 0 ? std::log(w / tot) : (k == 0 ? 0.0 : -INFINITY));
            }
        }
    }
    return lp;
}

void Model::set_transition_stay_prior(double prior) {
    // setTransitionPriors.m:12-33
    for (auto &tm : temporal_map) {
        int ii = tm[1] - 1, jj = tm[0] - 1;
        if (N_transition[ii].empty()) continue;
        bool has_par = false;
        for (int p = 0; p < n_transition; p++) has_par |= G_transition[(size_t)p * n_transition + ii] != 0;
        if (!has_par) continue;
        int rj = r_transition[jj];
        int64_t n = q_transition[ii];
        if ((int64_t)rj * n != (int64_t)N_transition[ii].size())
            throw Error(EMGPU_ERR_ARG, "setTransitionPriors: r of the variable at t differs from its t+1 table");
        std::fill(A_transition[ii].begin(), A_transition[ii].end(), 0.0);
        int64_t nn = n / rj;
        for (int kk = 1; kk <= rj; kk++)
            for (int64_t c = nn * (kk - 1) + 1; c <= nn * kk; c++) A_transition[ii][(size_t)(c - 1) * rj + (kk - 1)] = prior;
    }
    version++;
}

void Model::finalize() {
    static std::atomic<uint64_t> next_uid{1};
    uid = next_uid.fetch_add(1);
    if (n_initial <= 0) throw Error(EMGPU_ERR_PARSE, "model has no initial network");
    if ((int)r_initial.size() != n_initial) throw Error(EMGPU_ERR_PARSE, "r_initial size mismatch");
    order_initial = bn_sort(G_initial, n_initial);
    q_initial.resize(n_initial);
    for (int i = 0; i < n_initial; i++) {
        q_initial[i] = parent_count(G_initial, n_initial, r_initial, i);
        if ((int64_t)N_initial[i].size() != q_initial[i] * r_initial[i]) throw Error(EMGPU_ERR_PARSE, "N_initial size mismatch");
    }
    if (n_transition > 0) {
        if ((int)r_transition.size() != n_transition) throw Error(EMGPU_ERR_PARSE, "r_transition size mismatch");
        order_transition = bn_sort(G_transition, n_transition);
        q_transition.assign(n_transition, 0);
        N_transition.resize(n_transition);
        for (int i = n_initial; i < n_transition; i++) {
            q_transition[i] = parent_count(G_transition, n_transition, r_transition, i);
            if ((int64_t)N_transition[i].size() != q_transition[i] * r_transition[i]) throw Error(EMGPU_ERR_PARSE, "N_transition size mismatch");
        }
        if (temporal_map.empty() && !labels_transition.empty()) temporal_map = extract_temporal_map(labels_transition);
    }
    if (boundaries.empty()) boundaries.assign(n_initial, {});
    if (zero_bins.empty()) zero_bins = extract_zero_bins(boundaries);
    if (resample_rates.empty()) resample_rates.assign(n_initial, 0.0);
    if (start.empty()) start.assign(n_initial, 0);
    set_prior(0, 0.0); // EncounterModel.m:45 prior = 0
}

// ---------------------------------------------------------------------------------------------
// em_read.m:47-141
// ---------------------------------------------------------------------------------------------
Model *load_txt(const char *path, const int32_t *idx_zero, int n_idx, bool overwrite) {
    FILE *f = fopen(path, "rb");
    if (!f) throw Error(EMGPU_ERR_IO, std::string("cannot open ") + path);
    std::string text;
    char buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, got);
    fclose(f);
    // textscan(fid,'%s','EndOfLine','\r\n','Whitespace','\r\n'): one string per non-empty line
    std::vector<std::string> lines;
    size_t pos = 0;
    while (pos <= text.size()) {
        size_t e = text.find_first_of("\r\n", pos);
        if (e == std::string::npos) e = text.size();
        if (e > pos) lines.push_back(text.substr(pos, e - pos));
        pos = e + 1;
    }
    std::unique_ptr<Model> m(new Model());
    std::vector<double> flat_init, flat_trans;
    bool have_bnd = false;
    size_t n_fields = 0;
    for (size_t i = 0; i < lines.size(); i++) {
        if (lines[i].find('#') == std::string::npos) continue;
        std::string field = rtrim(lines[i]);
        size_t row = i + 1;
        auto need = [&](size_t k) {
            if (row + k > lines.size()) throw Error(EMGPU_ERR_PARSE, "truncated section " + field);
        };
        n_fields++;
        if (field == "# labels_initial") {
            need(1);
            m->labels_initial = split_labels(lines[row]);
            m->n_initial = (int)m->labels_initial.size();
        } else if (field == "# G_initial") {
            need(m->n_initial);
            for (int k = 0; k < m->n_initial; k++) {
                std::vector<double> v;
                scan_numbers(lines[row + k], v);
                if ((int)v.size() != m->n_initial) throw Error(EMGPU_ERR_PARSE, "G_initial row width");
                for (double x : v) m->G_initial.push_back(x != 0);
            }
        } else if (field == "# r_initial") {
            need(1);
            std::vector<double> v;
            scan_numbers(lines[row], v);
            for (double x : v) m->r_initial.push_back((int)x);
        } else if (field == "# N_initial") {
            need(1);
            scan_numbers(lines[row], flat_init);
        } else if (field == "# labels_transition") {
            need(1);
            m->labels_transition = split_labels(lines[row]);
            m->n_transition = (int)m->labels_transition.size();
        } else if (field == "# G_transition") {
            need(m->n_transition);
            for (int k = 0; k < m->n_transition; k++) {
                std::vector<double> v;
                scan_numbers(lines[row + k], v);
                if ((int)v.size() != m->n_transition) throw Error(EMGPU_ERR_PARSE, "G_transition row width");
                for (double x : v) m->G_transition.push_back(x != 0);
            }
        } else if (field == "# r_transition") {
            need(1);
            std::vector<double> v;
            scan_numbers(lines[row], v);
            for (double x : v) m->r_transition.push_back((int)x);
        } else if (field == "# N_transition") {
            need(1);
            scan_numbers(lines[row], flat_trans);
        } else if (field == "# boundaries") {
            need(m->n_initial);
            m->boundaries.resize(m->n_initial);
            for (int k = 0; k < m->n_initial; k++) {
                std::string s = trim(lines[row + k]);
                if (s != "*") scan_numbers(s, m->boundaries[k]); // '*' -> textscan yields empty (em_read.m:97-99)
            }
            have_bnd = true;
        } else if (field == "# resample_rates") {
            need(1);
            scan_numbers(lines[row], m->resample_rates);
        } else {
            throw Error(EMGPU_ERR_PARSE, "Unknown field: " + field); // em_read.m:104
        }
    }
    if (n_fields == 0 || m->n_initial == 0) throw Error(EMGPU_ERR_PARSE, "no '# labels_initial' section");
    if ((int)m->G_initial.size() != m->n_initial * m->n_initial || (int)m->r_initial.size() != m->n_initial)
        throw Error(EMGPU_ERR_PARSE, "G_initial / r_initial missing or malformed");
    // array2cells (em_read.m:191-198)
    {
        m->N_initial.resize(m->n_initial);
        size_t idx = 0;
        for (int i = 0; i < m->n_initial; i++) {
            size_t cnt = (size_t)parent_count(m->G_initial, m->n_initial, m->r_initial, i) * m->r_initial[i];
            if (idx + cnt > flat_init.size()) throw Error(EMGPU_ERR_PARSE, "N_initial too short");
            m->N_initial[i].assign(flat_init.begin() + idx, flat_init.begin() + idx + cnt);
            idx += cnt;
        }
        if (idx != flat_init.size()) throw Error(EMGPU_ERR_PARSE, "N_initial has trailing values");
    }
    if (m->n_transition > 0) {
        if ((int)m->G_transition.size() != m->n_transition * m->n_transition || (int)m->r_transition.size() != m->n_transition)
            throw Error(EMGPU_ERR_PARSE, "G_transition / r_transition missing or malformed");
        m->N_transition.resize(m->n_transition);
        size_t idx = 0;
        for (int i = m->n_initial; i < m->n_transition; i++) {
            size_t cnt = (size_t)parent_count(m->G_transition, m->n_transition, m->r_transition, i) * m->r_transition[i];
            if (idx + cnt > flat_trans.size()) throw Error(EMGPU_ERR_PARSE, "N_transition too short");
            m->N_transition[i].assign(flat_trans.begin() + idx, flat_trans.begin() + idx + cnt);
            idx += cnt;
        }
        if (idx != flat_trans.size()) throw Error(EMGPU_ERR_PARSE, "N_transition has trailing values");
    }
    if (have_bnd) {
        m->zero_bins = extract_zero_bins(m->boundaries); // before the overwrite, em_read.m:116-121
        if (overwrite) {
            static const int32_t dflt[3] = {1, 2, 3};
            if (!idx_zero) { idx_zero = dflt; n_idx = 3; }
            for (int k = 0; k < n_idx; k++) {
                if (idx_zero[k] < 1 || idx_zero[k] > m->n_initial) throw Error(EMGPU_ERR_ARG, "idxZeroBoundaries out of range");
                m->boundaries[idx_zero[k] - 1].clear();
            }
        }
    }
    m->finalize();
    return m.release();
}

// ---------------------------------------------------------------------------------------------
// Binary model cache: [magic "EMGPUBIN"][u32 format][source hash, 0-terminated, 32 bytes][u64 checksum of what follows][model][u8 has_plan][plan]
// The source hash says WHO wrote the file (the plan's layout belongs to one build); the checksum says the bytes are the ones that were
// written (a truncated copy, a flipped bit: the plan's raw offsets index host and device tables, so a damaged file is refused as a whole
// and the caller reads the .txt again); the range checks in load_bin cover what the host code indexes with before any plan is used.
// ---------------------------------------------------------------------------------------------
namespace {
constexpr uint32_t kBinFormat = 2;
constexpr size_t kBinHeader = 8 + 4 + 32 + 8;   // magic, format, tag, checksum
uint64_t payload_checksum(const char *p, size_t n) {   // FNV-1a over 8-byte words (the tail byte by byte), folded with the length
    uint64_t h = 0xcbf29ce484222325ull ^ (uint64_t)n;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, p + i, 8); h = (h ^ w) * 0x100000001b3ull; h ^= h >> 29; }
    for (; i < n; i++) h = (h ^ (uint8_t)p[i]) * 0x100000001b3ull;
    return h;
}
struct Writer {
    std::string buf;
    void raw(const void *p, size_t n) { buf.append((const char *)p, n); }
    template <typename T> void pod(const T &v) { raw(&v, sizeof v); }
    template <typename T> void vec(const std::vector<T> &v) { pod<uint64_t>(v.size()); if (!v.empty()) raw(v.data(), v.size() * sizeof(T)); }
    void str(const std::string &t) { pod<uint64_t>(t.size()); raw(t.data(), t.size()); }
};
struct Reader {
    const char *p, *end;
    void raw(void *dst, size_t n) {
        if ((size_t)(end - p) < n) throw Error(EMGPU_ERR_PARSE, "binary model cache: truncated file");
        memcpy(dst, p, n); p += n;
    }
    template <typename T> T pod() { T v; raw(&v, sizeof v); return v; }
    template <typename T> void vec(std::vector<T> &v) {
        const uint64_t n = pod<uint64_t>();
        if (n > (uint64_t)(end - p) / sizeof(T)) throw Error(EMGPU_ERR_PARSE, "binary model cache: truncated file");
        v.resize((size_t)n);
        if (n) raw(v.data(), (size_t)n * sizeof(T));
    }
    std::string str() {
        const uint64_t n = pod<uint64_t>();
        if (n > (uint64_t)(end - p)) throw Error(EMGPU_ERR_PARSE, "binary model cache: truncated file");
        std::string t(p, (size_t)n); p += n; return t;
    }
};
template <typename W> void put_tables(W &w, const std::vector<std::vector<double>> &t) { w.template pod<uint64_t>(t.size()); for (auto &x : t) w.vec(x); }
void get_tables(Reader &r, std::vector<std::vector<double>> &t) { const uint64_t n = r.pod<uint64_t>(); if (n > 4096) throw Error(EMGPU_ERR_PARSE, "binary model cache: bad table count"); t.resize((size_t)n); for (auto &x : t) r.vec(x); }
} // namespace

std::shared_ptr<const CompiledPlan> plan_of(const Model &m) {
    // One lock for every model: the multi-device entry points run one host thread per device on the SAME model, and each of them comes
    // here (get_uploaded); unguarded, the threads of a cold model would compile and assign the shared_ptr while the others copy it.
    // Compiling under the lock means the second thread waits for the first one's plan instead of compiling its own.
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (!m.plan_cache || m.plan_version != m.version) {
        m.plan_cache = std::make_shared<const CompiledPlan>(compile_plan(m));
        m.plan_version = m.version;
    }
    return m.plan_cache;
}

void save_bin(const Model &m, const char *path, const char *src_hash) {
    Writer w;
    w.raw("EMGPUBIN", 8);
    w.pod<uint32_t>(kBinFormat);
    char tag[32] = {0};
    strncpy(tag, src_hash ? src_hash : "", sizeof tag - 1);
    w.raw(tag, sizeof tag);
    w.pod<uint64_t>(0);   // the checksum, filled in below
    w.pod<int32_t>(m.n_initial); w.pod<int32_t>(m.n_transition);
    w.pod<uint64_t>(m.labels_initial.size()); for (auto &t : m.labels_initial) w.str(t);
    w.pod<uint64_t>(m.labels_transition.size()); for (auto &t : m.labels_transition) w.str(t);
    w.vec(m.G_initial); w.vec(m.G_transition); w.vec(m.r_initial); w.vec(m.r_transition);
    w.vec(m.order_initial); w.vec(m.order_transition);
    w.pod<uint64_t>(m.temporal_map.size()); for (auto &a : m.temporal_map) { w.pod<int32_t>(a[0]); w.pod<int32_t>(a[1]); }
    put_tables(w, m.N_initial); put_tables(w, m.N_transition); put_tables(w, m.A_initial); put_tables(w, m.A_transition);
    w.vec(m.q_initial); w.vec(m.q_transition);
    put_tables(w, m.boundaries);
    w.vec(m.zero_bins); w.vec(m.resample_rates); w.vec(m.start);
    // the plan (device pointers are not part of it: they are filled at upload)
    bool has_plan = true;
    std::shared_ptr<const CompiledPlan> cp;
    try { cp = plan_of(m); } catch (const Error &) { has_plan = false; }   // a model the kernels do not take is still a model
    w.pod<uint8_t>(has_plan ? 1 : 0);
    if (has_plan) {
        EmgpuPlan P = cp->plan;
        P.thr = nullptr; P.cthr = nullptr; P.pthr = nullptr; P.bnd = nullptr;
        w.pod(P);
        w.vec(cp->thr); w.vec(cp->cthr); w.vec(cp->pthr); w.vec(cp->bnd); w.vec(cp->pos_of_var);
    }
    const uint64_t sum = payload_checksum(w.buf.data() + kBinHeader, w.buf.size() - kBinHeader);
    memcpy(&w.buf[kBinHeader - 8], &sum, 8);
    const std::string tmp = std::string(path) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) throw Error(EMGPU_ERR_IO, std::string("cannot write ") + tmp);
    const bool ok = fwrite(w.buf.data(), 1, w.buf.size(), f) == w.buf.size();
    if (fclose(f) != 0 || !ok || rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); throw Error(EMGPU_ERR_IO, std::string("cannot write ") + path); }
}

Model *load_bin(const char *path, const char *src_hash) {
    FILE *f = fopen(path, "rb");
    if (!f) throw Error(EMGPU_ERR_IO, std::string("cannot open ") + path);
    std::string text;
    char buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, got);
    fclose(f);
    Reader r{text.data(), text.data() + text.size()};
    char magic[8], tag[32];
    r.raw(magic, 8);
    if (memcmp(magic, "EMGPUBIN", 8) != 0) throw Error(EMGPU_ERR_PARSE, "not a binary model cache");
    if (r.pod<uint32_t>() != kBinFormat) throw Error(EMGPU_ERR_PARSE, "binary model cache: another format version");
    r.raw(tag, sizeof tag);
    tag[sizeof tag - 1] = 0;
    if (strcmp(tag, src_hash ? src_hash : "") != 0)
        throw Error(EMGPU_ERR_PARSE, std::string("binary model cache written by other sources (") + tag + "): read the .txt again");
    const uint64_t want_sum = r.pod<uint64_t>();   // (read first: the operands of != are unsequenced, and pod() moves r.p past the checksum)
    if (want_sum != payload_checksum(r.p, (size_t)(r.end - r.p)))
        throw Error(EMGPU_ERR_PARSE, "binary model cache: checksum mismatch (damaged file): read the .txt again");
    std::unique_ptr<Model> m(new Model());
    m->n_initial = r.pod<int32_t>(); m->n_transition = r.pod<int32_t>();
    if (m->n_initial < 1 || m->n_initial > 4096 || m->n_transition < 0 || m->n_transition > 4096) throw Error(EMGPU_ERR_PARSE, "binary model cache: bad sizes");
    { uint64_t n = r.pod<uint64_t>(); if (n > 4096) throw Error(EMGPU_ERR_PARSE, "binary model cache: bad sizes"); m->labels_initial.resize((size_t)n); for (auto &t : m->labels_initial) t = r.str(); }
    { uint64_t n = r.pod<uint64_t>(); if (n > 4096) throw Error(EMGPU_ERR_PARSE, "binary model cache: bad sizes"); m->labels_transition.resize((size_t)n); for (auto &t : m->labels_transition) t = r.str(); }
    r.vec(m->G_initial); r.vec(m->G_transition); r.vec(m->r_initial); r.vec(m->r_transition);
    r.vec(m->order_initial); r.vec(m->order_transition);
    { uint64_t n = r.pod<uint64_t>(); if (n > 4096) throw Error(EMGPU_ERR_PARSE, "binary model cache: bad sizes"); m->temporal_map.resize((size_t)n); for (auto &a : m->temporal_map) { a[0] = r.pod<int32_t>(); a[1] = r.pod<int32_t>(); } }
    get_tables(r, m->N_initial); get_tables(r, m->N_transition); get_tables(r, m->A_initial); get_tables(r, m->A_transition);
    r.vec(m->q_initial); r.vec(m->q_transition);
    get_tables(r, m->boundaries);
    r.vec(m->zero_bins); r.vec(m->resample_rates); r.vec(m->start);
    const size_t ni = (size_t)m->n_initial, nt = (size_t)m->n_transition;
    if (m->G_initial.size() != ni * ni || m->r_initial.size() != ni || m->N_initial.size() != ni || m->A_initial.size() != ni || m->boundaries.size() != ni ||
        m->zero_bins.size() != ni || m->resample_rates.size() != ni || m->start.size() != ni || m->order_initial.size() != ni || m->q_initial.size() != ni ||
        m->G_transition.size() != nt * nt || m->r_transition.size() != nt || (nt && (m->N_transition.size() != nt || m->A_transition.size() != nt || m->q_transition.size() != nt)))
        throw Error(EMGPU_ERR_PARSE, "binary model cache: inconsistent sizes");
    // what host code indexes with: orders are permutations of 1..n, a node's tables hold r x q entries, boundaries r + 1 values or none
    auto is_perm = [](const std::vector<int> &o, size_t n) {
        if (o.size() != n) return false;
        std::vector<uint8_t> seen(n, 0);
        for (int v : o) { if (v < 1 || (size_t)v > n || seen[(size_t)v - 1]) return false; seen[(size_t)v - 1] = 1; }
        return true;
    };
    bool ok = is_perm(m->order_initial, ni) && (m->order_transition.empty() || is_perm(m->order_transition, nt));
    for (size_t v = 0; ok && v < ni; v++) {
        const int64_t r_ = m->r_initial[v], q_ = m->q_initial[v];
        ok = r_ >= 1 && r_ <= EMGPU_MAX_R && q_ >= 1 && q_ <= ((int64_t)1 << 40) / r_ && m->N_initial[v].size() == (size_t)(r_ * q_) && m->A_initial[v].size() == (size_t)(r_ * q_) &&
             (m->boundaries[v].empty() || m->boundaries[v].size() == (size_t)r_ + 1) && m->zero_bins[v] >= 0 && m->zero_bins[v] <= r_ && m->start[v] >= 0 && m->start[v] <= r_;
    }
    for (size_t v = 0; ok && v < nt; v++) {
        const int64_t r_ = m->r_transition[v];
        ok = r_ >= 1 && r_ <= EMGPU_MAX_R;
        if (ok && !m->N_transition[v].empty()) {
            const int64_t q_ = m->q_transition[v];
            ok = q_ >= 1 && q_ <= ((int64_t)1 << 40) / r_ && m->N_transition[v].size() == (size_t)(r_ * q_) && m->A_transition[v].size() == (size_t)(r_ * q_);
        }
    }
    for (auto &a : m->temporal_map) ok = ok && a[0] >= 1 && (size_t)a[0] <= ni && a[1] > (int)ni && (size_t)a[1] <= nt;
    if (!ok) throw Error(EMGPU_ERR_PARSE, "binary model cache: values out of range");
    static std::atomic<uint64_t> bin_uid{1ull << 40};   // (finalize() numbers the models it builds from 1: two ranges, never equal)
    m->uid = bin_uid.fetch_add(1);
    m->version = 1;
    if (r.pod<uint8_t>()) {
        auto cp = std::make_shared<CompiledPlan>();
        cp->plan = r.pod<EmgpuPlan>();
        r.vec(cp->thr); r.vec(cp->cthr); r.vec(cp->pthr); r.vec(cp->bnd); r.vec(cp->pos_of_var);
        bool fits = cp->plan.ni == m->n_initial && cp->thr.size() >= cp->plan.thr_total && cp->cthr.size() >= cp->plan.cthr_total && cp->pthr.size() >= cp->plan.pthr_total &&
                    cp->pos_of_var.size() == ni;
        for (size_t v = 0; fits && v < ni; v++) fits = cp->pos_of_var[v] >= 0 && (size_t)cp->pos_of_var[v] < ni;
        for (int q = 0; fits && q < cp->plan.ni; q++)
            fits = cp->plan.i_var[q] < ni && cp->plan.i_off[q] <= cp->thr.size() && (size_t)cp->plan.i_boff[q] + cp->plan.i_nb[q] <= cp->bnd.size();
        if (!fits) throw Error(EMGPU_ERR_PARSE, "binary model cache: plan does not fit the model");
        m->plan_cache = cp;
        m->plan_version = m->version;
    }
    if (r.p != r.end) throw Error(EMGPU_ERR_PARSE, "binary model cache: trailing bytes");
    return m.release();
}

// ---------------------------------------------------------------------------------------------
// Quantile thresholds (select_random.m:17-20 folded into integers)
// ---------------------------------------------------------------------------------------------
void column_thresholds(const double *w, int r, uint32_t *out) {
    double s[EMGPU_MAX_R];
    double acc = 0.0;
    for (int k = 0; k < r; k++) { acc += w[k]; s[k] = acc; } // cumsum
    const double total = s[r - 1];
    for (int k = 0; k < r - 1; k++) {
        const double sk = s[k];
        // pred(x') := !(s_k >= total * u(x')) : "bin k is NOT selected, look further"
        auto pred = [&](uint32_t x) { return !(sk >= total * uniform32(x)); };
        if (sk >= total) { out[k] = 0xFFFFFFFFu; continue; }   // fl(total*u) <= total for u < 1
        if (pred(0u)) { out[k] = 0u; continue; }
        uint32_t lo = 0u, hi = 0xFFFFFFFFu;                      // hi == "never"
        while (lo < hi) {
            uint32_t mid = lo + (hi - lo) / 2u;                  // mid <= 2^32-2
            if (pred(mid)) hi = mid; else lo = mid + 1u;
        }
        out[k] = lo;
    }
}

uint32_t bernoulli_threshold(double rate) {
    // hit(x') := u(x') < rate (resample_events.m:24); R = #{x' in [0, 2^32-2] : hit}
    auto hit = [&](uint32_t x) { return uniform32(x) < rate; };
    if (!(rate > 0.0) || !hit(0u)) return 0u;
    uint32_t lo = 0u, hi = 0xFFFFFFFFu;
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo) / 2u;
        if (!hit(mid)) hi = mid; else lo = mid + 1u;
    }
    return lo;
}

CompiledPlan compile_plan(const Model &m) {
    CompiledPlan cp;
    EmgpuPlan &P = cp.plan;
    const int ni = m.n_initial, nt = m.n_transition, nd = m.n_dyn();
    if (ni > EMGPU_MAX_NI) throw Error(EMGPU_ERR_UNSUPPORTED, "n_initial exceeds EMGPU_MAX_NI");
    if (nd > EMGPU_MAX_ND) throw Error(EMGPU_ERR_UNSUPPORTED, "more dynamic variables than EMGPU_MAX_ND");
    for (int r : m.r_initial) if (r < 1 || r > EMGPU_MAX_R) throw Error(EMGPU_ERR_UNSUPPORTED, "r out of range");
    for (int r : m.r_transition) if (r < 1 || r > EMGPU_MAX_R) throw Error(EMGPU_ERR_UNSUPPORTED, "r out of range");
    P.ni = ni; P.nd = nd; P.depend = (nt > 0 && nd > 0) ? (int)m.is_dynvar_depend() : 0;
    // dbn_sample.m:36 hands r_transition to bn_sample; CorTerminalModel/sample.m:34 hands r_initial
    const std::vector<int> &r_par = nt > 0 ? m.r_transition : m.r_initial;
    cp.pos_of_var.assign(ni, -1);
    for (int p = 0; p < ni; p++) cp.pos_of_var[m.order_initial[p] - 1] = p;
    // boundaries table
    std::vector<int> boff(ni, 0);
    for (int v = 0; v < ni; v++) {
        boff[v] = (int)cp.bnd.size();
        if (m.boundaries[v].size() > 255) throw Error(EMGPU_ERR_UNSUPPORTED, "too many boundaries");
        if (!m.boundaries[v].empty() && (int)m.boundaries[v].size() < m.r_initial[v] + 1)
            throw Error(EMGPU_ERR_ARG, "boundaries shorter than r+1 (dediscretize.m:33-38 would fail)");
        cp.bnd.insert(cp.bnd.end(), m.boundaries[v].begin(), m.boundaries[v].end());
    }
    cp.bnd.push_back(0.0);
    if (cp.bnd.size() > 65535) throw Error(EMGPU_ERR_UNSUPPORTED, "boundary table too large");

    double w[EMGPU_MAX_R];
    auto emit_node = [&](const std::vector<double> &N, const std::vector<double> &A, int r, int64_t q) -> uint32_t {
        size_t off = cp.thr.size();
        if (off + (size_t)q * (r - 1) > 0xFFFFFFF0ull) throw Error(EMGPU_ERR_UNSUPPORTED, "threshold table too large");
        cp.thr.resize(off + (size_t)q * (r - 1));
        for (int64_t j = 0; j < q; j++) {
            for (int k = 0; k < r; k++) w[k] = N[(size_t)j * r + k] + A[(size_t)j * r + k]; // bn_sample.m:55
            if (r > 1) column_thresholds(w, r, cp.thr.data() + off + (size_t)j * (r - 1));
        }
        return (uint32_t)off;
    };

    for (int p = 0; p < ni; p++) {
        int v = m.order_initial[p] - 1;
        P.i_var[p] = (uint8_t)v;
        P.i_r[p] = (uint8_t)m.r_initial[v];
        P.i_start[p] = (uint8_t)m.start[v];
        if (m.start[v] < 0 || m.start[v] > m.r_initial[v]) throw Error(EMGPU_ERR_ARG, "start bin out of range");
        P.i_nb[p] = (uint8_t)m.boundaries[v].size();
        P.i_zero[p] = (uint8_t)m.zero_bins[v];
        P.i_skip[p] = (uint8_t)((int)m.boundaries[v].size() == m.r_initial[v] - 2); // dbn_hierarchical_sample.m:26
        P.i_boff[p] = (uint16_t)boff[v];
        int64_t stride = 1;
        int n_par = 0, n_par_set = 0;
        for (int u = 0; u < ni; u++) {
            if (!m.G_initial[(size_t)u * ni + v]) continue;
            int q = cp.pos_of_var[u];
            if (q >= p) throw Error(EMGPU_ERR_SORT, "parent after child in topological order");
            P.i_stride[p][q] = (uint32_t)stride;
            stride *= r_par[u];
            n_par++;
            if (m.start[u] != 0) n_par_set++;
        }
        if (m.start[v] != 0 && n_par > 0 && n_par_set < n_par)
            throw Error(EMGPU_ERR_PRESET, "Attempt to preset a dependent variable"); // bn_sample.m:45-47
        if (stride != m.q_initial[v] && nt > 0) {
            // r_transition(parents) disagrees with the table width: the reference would index out of range
            bool same = true;
            for (int u = 0; u < ni; u++) same &= (m.r_transition[u] == m.r_initial[u]);
            if (!same) throw Error(EMGPU_ERR_ARG, "r_transition(1:n_initial) differs from r_initial");
        }
        P.i_off[p] = emit_node(m.N_initial[v], m.A_initial[v], m.r_initial[v], m.q_initial[v]);
    }

    // dynamic variables in sampling order (dbn_sample.m:68-69: order_transition, dynamic only)
    if (nd > 0) {
        std::vector<int> k_of_tvar(nt, -1), k_of_ivar(ni, -1);
        int k = 0;
        for (int oi = 0; oi < nt; oi++) {
            int tv = m.order_transition[oi] - 1;
            int row = -1;
            for (int q = 0; q < nd; q++)
                if (m.temporal_map[q][1] - 1 == tv) {
                    if (row >= 0) throw Error(EMGPU_ERR_UNSUPPORTED, "variable appears twice in the temporal map");
                    row = q;
                }
            if (row < 0) continue;
            int iv = m.temporal_map[row][0] - 1;
            if (iv < 0 || iv >= ni || tv < ni || k_of_ivar[iv] >= 0) throw Error(EMGPU_ERR_UNSUPPORTED, "unsupported temporal map");
            if (m.N_transition[tv].empty()) throw Error(EMGPU_ERR_ARG, "dynamic variable without a transition table");
            P.d_tvar[k] = (uint8_t)tv; P.d_ivar[k] = (uint8_t)iv; P.d_ipos[k] = (uint8_t)cp.pos_of_var[iv];
            P.d_r[k] = (uint8_t)m.r_transition[tv]; P.d_row[k] = (uint8_t)row;
            P.d_nb[k] = (uint8_t)m.boundaries[iv].size(); P.d_zero[k] = (uint8_t)m.zero_bins[iv]; P.d_boff[k] = (uint16_t)boff[iv];
            if (m.r_transition[tv] != m.r_initial[iv]) throw Error(EMGPU_ERR_UNSUPPORTED, "t and t+1 copies of a variable differ in r");
            k_of_tvar[tv] = k; k_of_ivar[iv] = k;
            k++;
        }
        if (k != nd) throw Error(EMGPU_ERR_UNSUPPORTED, "temporal map rows not found in order_transition");
        // emission order: ascending initial variable id (dbn_sample.m:86 / :152 loop 1:n_initial)
        std::vector<int> ks(nd);
        for (int q = 0; q < nd; q++) ks[q] = q;
        std::sort(ks.begin(), ks.end(), [&](int a, int b) { return P.d_ivar[a] < P.d_ivar[b]; });
        for (int e = 0; e < nd; e++) P.d_emit[e] = (uint8_t)ks[e];
        for (k = 0; k < nd; k++) {
            int tv = P.d_tvar[k];
            int64_t stride = 1;
            for (int u = 0; u < nt; u++) {
                if (!m.G_transition[(size_t)u * nt + tv]) continue;
                if (u < ni) {
                    if (k_of_ivar[u] >= 0) P.d_stride_cur[k][k_of_ivar[u]] = (uint32_t)stride;
                    else P.d_stride_static[k][cp.pos_of_var[u]] = (uint32_t)stride;
                } else {
                    int kp = k_of_tvar[u];
                    if (kp < 0 || kp >= k) throw Error(EMGPU_ERR_UNSUPPORTED, "transition parent is not an earlier dynamic variable");
                    P.d_stride_new[k][kp] = (uint32_t)stride;
                }
                stride *= m.r_transition[u];
            }
            P.d_off[k] = emit_node(m.N_transition[tv], m.A_transition[tv], m.r_transition[tv], m.q_transition[tv]);
        }
        // Compacted tables: per column keep only the distinct thresholds that can fire (0 < X < 2^32-1)
        // and a nibble map from "how many fired" to the bin.  bin(x') = 1 + #{X == 0} + sum of the
        // multiplicities of the distinct thresholds <= x' -- exactly the full table's answer.
        for (k = 0; k < nd; k++) {
            const int r = P.d_r[k], rm1 = r - 1;
            const int64_t q = m.q_transition[P.d_tvar[k]];
            const uint32_t *X = cp.thr.data() + P.d_off[k];
            int meff = 0;
            for (int64_t j = 0; j < q; j++) {
                int d = 0;
                uint32_t prev = 0u;
                for (int t = 0; t < rm1; t++) {
                    const uint32_t x = X[(size_t)j * rm1 + t];
                    if (x != 0u && x != 0xFFFFFFFFu && x != prev) { d++; prev = x; }
                }
                meff = d > meff ? d : meff;
            }
            if (meff < 1) meff = 1;
            if (meff > 7 || r > 15) { P.d_meff[k] = 0; continue; } // the map has 8 nibbles
            P.d_meff[k] = (uint8_t)meff;
            P.d_coff[k] = (uint32_t)cp.cthr.size();
            cp.cthr.resize(cp.cthr.size() + (size_t)q * (meff + 1));
            uint32_t *C = cp.cthr.data() + P.d_coff[k];
            for (int64_t j = 0; j < q; j++) {
                uint32_t *c = C + (size_t)j * (meff + 1);
                int d = 0, bin = 1;
                uint32_t prev = 0u, map = 0u;
                for (int t = 0; t < rm1; t++) bin += X[(size_t)j * rm1 + t] == 0u;   // thresholds that always fire
                map = (uint32_t)bin;
                for (int t = 0; t < rm1; t++) {
                    const uint32_t x = X[(size_t)j * rm1 + t];
                    if (x == 0u || x == 0xFFFFFFFFu) continue;
                    if (x != prev) { c[d++] = x; prev = x; }
                    bin++;
                    map = (map & ~(0xFu << (4 * d))) | ((uint32_t)bin << (4 * d));
                }
                // pad with copies of the last real threshold (they fire together with it, so the count jumps to
                // meff and no new tie value appears: a tie with a high halfword costs the kernels an exact redo);
                // a column without real thresholds is padded with "never"
                for (int t = d; t < meff; t++) { c[t] = d > 0 ? c[d - 1] : 0xFFFFFFFFu; map |= (uint32_t)bin << (4 * (t + 1)); }
                c[meff] = map;
            }
            // the padded form (EmgpuPlan::d_pw)
            const int mfix = meff <= 3 ? 3 : (meff <= 6 ? 6 : 0);
            P.d_pw[k] = 0;
            if (mfix) {
                const int W = mfix == 3 ? 4 : 8;
                P.d_pw[k] = (uint8_t)W;
                P.d_poff[k] = (uint32_t)cp.pthr.size();
                cp.pthr.resize(cp.pthr.size() + (size_t)q * W);
                C = cp.cthr.data() + P.d_coff[k]; // (resize above may not move cthr, but stay safe)
                uint32_t *Pd = cp.pthr.data() + P.d_poff[k];
                for (int64_t j = 0; j < q; j++) {
                    const uint32_t *c = C + (size_t)j * (meff + 1);
                    uint32_t *o = Pd + (size_t)j * W;
                    for (int t = 0; t < mfix; t++) o[t] = t < meff ? c[t] : c[meff - 1];
                    uint32_t lo = 0u, hi = 0u;
                    for (int b = 0; b <= mfix; b++) {
                        const int n = mfix - b, nn = n < meff ? n : meff;
                        const uint32_t e = (c[meff] >> (4 * nn)) & 15u;
                        if (b < 4) lo |= e << (8 * b); else hi |= e << (8 * (b - 4));
                    }
                    o[mfix] = lo;
                    if (W == 8) o[7] = hi;
                }
                {   // the packed-compare form (EmgpuPlan::d_poffpk): T' pairs + the nibble map by fired count
                    P.d_poffpk[k] = (uint32_t)cp.pthr.size();
                    cp.pthr.resize(cp.pthr.size() + (size_t)q * 4);
                    C = cp.cthr.data() + P.d_coff[k];
                    uint32_t *Pk = cp.pthr.data() + P.d_poffpk[k];
                    for (int64_t j = 0; j < q; j++) {
                        const uint32_t *c = C + (size_t)j * (meff + 1);
                        uint32_t *o = Pk + (size_t)j * 4;
                        uint32_t tq[6], prev = 0u;
                        for (int t = 0; t < 6; t++) {
                            uint32_t v = 0xFFFFu;
                            if (t < meff) {
                                const uint32_t h = c[t] >> 16;
                                v = h ? h - 1u : 0u;
                                if (t > 0 && v <= prev) v = prev + 1u;
                                v = v > 0xFFFFu ? 0xFFFFu : v;
                            }
                            tq[t] = v; prev = v;
                        }
                        if (W == 4) {
                            // columns of at most 3 thresholds: the PLAIN form {H0, H1, H2, map} -- the thresholds' high halves as 32-bit
                            // words (0x10000 beyond the column's: never reached), the bins 7 bits apart.  With a_t = H_t - x_h (plain
                            // subtracts): a_t < 0 <=> threshold t fired, a_t == 0 <=> the low halfword decides; (a_t >>> 29) is 7 or 0,
                            // their sum is the bit offset of the bin.  Three subtracts, three shifts, v_add3, v_bfe, v_min3: the packed
                            // form of the same decision costs a third more issue cycles (VOP3P / SDWA ops issue at half the rate).
                            for (int t = 0; t < 3; t++) o[t] = t < meff ? (c[t] >> 16) : 0x10000u;
                            uint32_t m7 = 0u;
                            for (int n = 0; n <= 3; n++) m7 |= ((c[meff] >> (4 * (n < meff ? n : meff))) & 15u) << (7 * n);
                            o[3] = m7;
                            continue;
                        }
                        for (int w = 0; w < 3; w++) o[w] = tq[2 * w] | (tq[2 * w + 1] << 16);
                        uint32_t nib = 0u;
                        for (int n = 0; n <= 6; n++) nib |= ((c[meff] >> (4 * (n < meff ? n : meff))) & 15u) << (4 * n);
                        o[3] = nib;
                    }
                }
                Pd = cp.pthr.data() + P.d_poff[k];   // (the resize above may have moved the buffer)
            }
        }
    }
    // resample_events.m:24
    int na = 0;
    for (int v = 0; v < ni; v++) {
        if (!(m.resample_rates[v] > 0.0)) continue;
        P.a_var[na] = (uint8_t)v; P.a_pos[na] = (uint8_t)cp.pos_of_var[v];
        P.a_dyn[na] = -1;
        for (int k = 0; k < nd; k++) if (P.d_ivar[k] == v) P.a_dyn[na] = (int8_t)k;
        P.a_R[na] = bernoulli_threshold(m.resample_rates[v]);
        na++;
    }
    P.nact = na;
    P.thr_total = (uint32_t)cp.thr.size();
    for (int k = 0; k < nd; k++) {   // pivot rows (EmgpuPlan::d_pivoff), appended past thr_total
        P.d_pivoff[k] = 0;
        const int rm1 = (int)P.d_r[k] - 1;
        if (rm1 <= 8 || rm1 > 48) continue;
        const int64_t q = m.q_transition[P.d_tvar[k]];
        const int ngrp = (rm1 + 5) / 6;
        P.d_pivoff[k] = (uint32_t)cp.thr.size();
        cp.thr.resize(cp.thr.size() + (size_t)q * 8);
        for (int64_t j = 0; j < q; j++)
            for (int g = 0; g < 8; g++)
                cp.thr[P.d_pivoff[k] + (size_t)j * 8 + g] = (g < ngrp - 1) ? cp.thr[P.d_off[k] + (size_t)j * rm1 + 6 * g + 5] : 0xFFFFFFFFu;
    }
    for (int k = 0; k < nd; k++) {   // compact rows of the long-row variables (EmgpuPlan::d_c8off), appended after the pivot rows
        P.d_c8off[k] = 0;
        const int rm1 = (int)P.d_r[k] - 1;
        if (rm1 <= 8 || rm1 > 48) continue;   // (the aligned 32-byte form of a 6-threshold row was measured too: the 24-byte rows it replaces are faster, +3 %)
        const int64_t q = m.q_transition[P.d_tvar[k]];
        P.d_c8off[k] = (uint32_t)cp.thr.size();
        cp.thr.resize(cp.thr.size() + (size_t)q * 8);
        for (int64_t j = 0; j < q; j++) {
            const uint32_t *X = cp.thr.data() + P.d_off[k] + (size_t)j * rm1;
            uint32_t *o = cp.thr.data() + P.d_c8off[k] + (size_t)j * 8;
            uint32_t dist[6], map[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int d = 0, bin = 1;
            bool over = false;
            uint32_t prev = 0u;
            for (int t = 0; t < rm1; t++) bin += X[t] == 0u;   // thresholds that always fire
            map[0] = (uint32_t)bin;
            for (int t = 0; t < rm1; t++) {
                const uint32_t x = X[t];
                if (x == 0u || x == 0xFFFFFFFFu) continue;
                if (x != prev) {
                    if (d == 6) { over = true; break; }
                    dist[d++] = x; prev = x;
                }
                bin++;
                map[d] = (uint32_t)bin;
            }
            for (int t = d; t < 6; t++) { dist[t] = 0xFFFFFFFFu; map[t + 1] = map[d]; }   // "never": the count stops at d
            for (int t = 0; t < 6; t++) o[t] = over ? 0xFFFFFFFFu : dist[t];
            o[6] = map[0] | (map[1] << 8) | (map[2] << 16) | (map[3] << 24);
            o[7] = map[4] | (map[5] << 8) | (map[6] << 16) | (over ? 0xFF000000u : 0u);
        }
    }
    P.cthr_total = (uint32_t)cp.cthr.size();
    P.pthr_total = (uint32_t)cp.pthr.size();
    return cp;
}

} // namespace emgpu
