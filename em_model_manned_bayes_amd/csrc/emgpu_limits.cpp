// emgpu_limits.cpp -- @UncorEncounterModel/getDynamicLimits.m:1-130 on the host: the speed and vertical-rate limits that
// UncorEncounterModel.track rejects against (UncorEncounterModel.m:459-470), evaluated once per model for every argument
// combination the expression can see, so that the track kernel only indexes a table with the track's own minima / maxima.
#include <cmath>

#include "../../include/emgpu.h"
#include "emgpu_model.hpp"

namespace emgpu {

// one evaluation: dG, dA 1-based bins; dL = l0..l1, dV = v0..v1 (the `min(d):1:max(d)` ranges of :38,50)
static void limits_from_sums(const Model &m, const UncorTrackVars &tv, const std::vector<double> &v_initial, const std::vector<double> &dh_initial, double out[3]) {
    const std::vector<double> &bV = m.boundaries[tv.idxV - 1], &bDH = m.boundaries[tv.idxDH - 1];
    auto pct_bins = [](const std::vector<double> &w, int &k_min, int &k_max, double &tot) {           // :94-99, :118-122
        tot = 0; for (double x : w) tot += x;
        double cs = 0; k_min = k_max = 0;
        for (size_t i = 0; i < w.size(); i++) {
            cs += 100.0 * w[i] / tot;
            if (!k_min && cs >= 1.0) k_min = (int)i + 1;
            if (!k_max && cs >= 99.0) k_max = (int)i + 1;
        }
    };
    int k0, k1; double tot;
    pct_bins(v_initial, k0, k1, tot);
    double min_speed = bV[(size_t)k0] * 1.68780972222222, max_speed = bV[(size_t)k1] * 1.68780972222222; // boundaries(k + 1)  :102-103
    if (tv.is_rotorcraft && max_speed > 304) max_speed = 304;                                        // :107-112
    if (!tv.is_rotorcraft && min_speed < 30) min_speed = 30;
    pct_bins(dh_initial, k0, k1, tot);
    const double a = std::fabs(bDH[(size_t)k0] / 60.0), b = std::fabs(bDH[(size_t)k1] / 60.0);
    double max_vr = a > b ? a : b;
    if (std::isnan(max_vr) || !(tot > 0)) max_vr = 0;                                                // :124-127
    out[0] = min_speed; out[1] = max_speed; out[2] = max_vr;
}

UncorLimits build_uncor_limits(const Model &m, const UncorTrackVars &tv) {
    UncorLimits L;
    auto need = [&](int idx, const char *what) {
        if (idx < 1 || idx > m.n_initial) throw Error(EMGPU_ERR_ARG, std::string("track: the model has no variable ") + what);
    };
    need(tv.idxL, "\"L\""); need(tv.idxV, "\"v\""); need(tv.idxDV, "\"\\dot v\""); need(tv.idxDH, "\"\\dot h\""); need(tv.idxDPsi, "\"\\dot \\psi\"");
    if (m.boundaries[tv.idxV - 1].empty() || m.boundaries[tv.idxDH - 1].empty())
        throw Error(EMGPU_ERR_UNSUPPORTED, "track: v and \\dot h need boundaries (getDynamicLimits.m:102,124 index them)");
    const int rV = m.r_initial[tv.idxV - 1], rDH = m.r_initial[tv.idxDH - 1];
    const std::vector<double> &NV = m.N_initial[tv.idxV - 1], &NDH = m.N_initial[tv.idxDH - 1];
    const int64_t qV = m.q_initial[tv.idxV - 1], qDH = m.q_initial[tv.idxDH - 1];
    auto cuts = [&](int idx, double *cut, const char *what) {                  // em_read.m:130-136
        const std::vector<double> &b = m.boundaries[idx - 1];
        // cut[] has 16 entries and the kernel's discretize reads cut[n - 1]: 1..15 cut points, one fewer than the variable has bins
        if (b.size() < 3 || b.size() - 2 > 15) throw Error(EMGPU_ERR_UNSUPPORTED, std::string("track: ") + what + " needs 2..16 bins (3..17 boundaries)");
        if ((int)b.size() - 1 != m.r_initial[idx - 1]) throw Error(EMGPU_ERR_UNSUPPORTED, std::string("track: the boundaries of ") + what + " do not match its number of bins");
        int n = 0;
        for (size_t i = 1; i + 1 < b.size(); i++) cut[n++] = b[i];
        return n;
    };
    L.discL = m.boundaries[tv.idxL - 1].empty(); L.discV = false;
    L.ncL = L.discL ? 0 : cuts(tv.idxL, L.cutL, "L");
    L.ncV = cuts(tv.idxV, L.cutV, "v");
    const bool is_idx = tv.idxG > 0 && tv.idxA > 0;
    L.ordered = is_idx && tv.idxG == 1 && tv.idxA == 2 && tv.idxL == 3 && tv.idxV == 4 && tv.idxDH == 6;   // :14,17
    std::vector<double> v_initial((size_t)rV), dh_initial((size_t)rDH);
    if (!L.ordered) {                                                                                      // :85-88
        for (int64_t c = 0; c < qV; c++) for (int i = 0; i < rV; i++) v_initial[(size_t)i] += NV[(size_t)(c * rV + i)];
        for (int64_t c = 0; c < qDH; c++) for (int i = 0; i < rDH; i++) dh_initial[(size_t)i] += NDH[(size_t)(c * rDH + i)];
        L.table.resize(3);
        limits_from_sums(m, tv, v_initial, dh_initial, L.table.data());
        return L;
    }
    if (!m.boundaries[0].empty() || !m.boundaries[1].empty())
        throw Error(EMGPU_ERR_UNSUPPORTED, "track: G and A are expected to be categorical ('*' boundaries) as in every shipped model");
    const int rG = m.r_initial[0], rA = m.r_initial[1], rL = m.r_initial[2];
    L.rG = rG; L.rA = rA; L.rL = rL; L.rV = rV;
    const int64_t nGA_V = qV / rG / rA, nGA_DH = qDH / rG / rA;
    if (qV % ((int64_t)rG * rA) || nGA_V < rL || nGA_DH % ((int64_t)rL * rV))
        throw Error(EMGPU_ERR_UNSUPPORTED, "track: N_initial{v} / N_initial{\\dot h} are not conditioned on (G, A, L[, v]) the way getDynamicLimits.m:57-79 slices them");
    const int64_t nLV = nGA_DH / rL / rV;
    L.table.assign((size_t)rG * rA * rL * rL * rV * rV * 3, 0.0);
    for (int dG = 1; dG <= rG; dG++)
        for (int dA = 1; dA <= rA; dA++)
            for (int l0 = 1; l0 <= rL; l0++)
                for (int l1 = l0; l1 <= rL; l1++) {
                    std::fill(v_initial.begin(), v_initial.end(), 0.0);
                    for (int dl = l0; dl <= l1; dl++) {                                                    // :57-66
                        const int64_t col = (dG - 1) + (int64_t)rG * ((dA - 1) + (int64_t)rA * (dl - 1));
                        for (int i = 0; i < rV; i++) v_initial[(size_t)i] += NV[(size_t)(col * rV + i)];
                    }
                    for (int b0 = 1; b0 <= rV; b0++)
                        for (int b1 = b0; b1 <= rV; b1++) {
                            std::fill(dh_initial.begin(), dh_initial.end(), 0.0);
                            for (int dl = l0; dl <= l1; dl++)                                              // :69-79
                                for (int dv = b0; dv <= b1; dv++)
                                    for (int64_t rest = 0; rest < nLV; rest++) {
                                        const int64_t c3 = (dl - 1) + (int64_t)rL * ((dv - 1) + (int64_t)rV * rest);
                                        const int64_t col = (dG - 1) + (int64_t)rG * ((dA - 1) + (int64_t)rA * c3);
                                        for (int i = 0; i < rDH; i++) dh_initial[(size_t)i] += NDH[(size_t)(col * rDH + i)];
                                    }
                            const size_t idx = (((((size_t)(dG - 1) * rA + (size_t)(dA - 1)) * rL + (size_t)(l0 - 1)) * rL + (size_t)(l1 - 1)) * rV + (size_t)(b0 - 1)) * rV + (size_t)(b1 - 1);
                            limits_from_sums(m, tv, v_initial, dh_initial, &L.table[idx * 3]);
                        }
                }
    return L;
}

} // namespace emgpu
