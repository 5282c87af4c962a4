"""oracle/pyref.py -- second, independently written restatement (pure Python + numpy).

TEST INFRASTRUCTURE ONLY.  Small cases only (pure-Python loops).

Purpose: pin oracle/em_oracle.c.  This file follows the reference MATLAB
line by line with MATLAB-shaped data (cell arrays -> lists, matrices -> numpy)
and draws from numpy's MT19937 (numpy.random.RandomState(seed).random_sample()
== MATLAB rng(seed,'twister'); rand).  tests/test_oracle_pinning.py requires the
C oracle in MT19937 mode to reproduce this file draw-for-draw.
"parity unpinned" against real MATLAB output (no MATLAB/Octave available).
"""
import numpy as np


class Rand:
    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)
        self.count = 0

    def rand(self, n=None):
        if n is None:
            self.count += 1
            return self.rs.random_sample()
        self.count += n
        return self.rs.random_sample(n)


def asub2ind(siz, x):
    # asub2ind.m:13-14
    siz = np.asarray(siz, dtype=float).ravel()
    x = np.asarray(x, dtype=float).ravel()
    k = np.concatenate([[1.0], np.cumprod(siz[:-1])])
    return int(k @ (x - 1) + 1)


def select_random(weights, R):
    # select_random.m:14-20
    r = R.rand()
    s = np.cumsum(weights)
    sthres = s[-1] * r
    return int(np.nonzero(s >= sthres)[0][0]) + 1


def bn_sample(G, r, N, alpha, start, order, R):
    # bn_sample.m:39-57 (num_samples = 1); N, alpha: lists of r_i x q_i arrays; start: list, None = []
    n = len(N)
    S = np.zeros(n)
    for i in order:                      # 1-based ids
        parents = G[:, i - 1]
        j = 1
        if start[i - 1] is not None:
            if parents.any() and sum(1 for p in np.nonzero(parents)[0] if start[p] is not None) < parents.sum():
                raise RuntimeError("Attempt to preset a dependent variable")
            S[i - 1] = start[i - 1]
        else:
            if parents.any():
                pm = np.nonzero(parents)[0]
                j = asub2ind(np.asarray(r)[pm], S[pm])
            S[i - 1] = select_random(N[i - 1][:, j - 1] + alpha[i - 1][:, j - 1], R)
    return S


def dbn_sample(p, di, dt, t_max, start, R):
    # dbn_sample.m:25-166
    G_t = p["G_transition"]; tm = p["temporal_map"]; r_t = p["r_transition"]
    n_i = p["n_initial"]; N_t = p["N_transition"]; order_t = p["order_transition"]
    initial = bn_sample(p["G_initial"], r_t, p["N_initial"], di, start, p["order_initial"], R)
    dyn = tm[:, 1]
    x = np.concatenate([initial, np.zeros(len(dyn))])
    delta_t = 0
    depend = G_t[np.ix_(dyn - 1, dyn - 1)].any()
    events = []
    if depend:
        for t in range(2, t_max + 1):
            delta_t += 1
            x_old = x.copy()
            for i in order_t:
                if (i == dyn).any():
                    parents = G_t[:, i - 1]
                    j = 1
                    if parents.any():
                        pm = np.nonzero(parents)[0]
                        j = asub2ind(np.asarray(r_t)[pm], x[pm])
                    x[i - 1] = select_random(N_t[i - 1][:, j - 1] + dt[i - 1][:, j - 1], R)
            x[tm[:, 0] - 1] = x[tm[:, 1] - 1]
            if (x[:n_i] != x_old[:n_i]).any():
                for i in range(1, n_i + 1):
                    if x[i - 1] != x_old[i - 1]:
                        events.append([delta_t, i, x[i - 1]])
                        delta_t = 0
    else:
        s = {}
        sthres = {}
        for ii in order_t:
            if (ii == dyn).any():
                parents = G_t[:, ii - 1]
                if parents.any():
                    pm = np.nonzero(parents)[0]
                    j = asub2ind(np.asarray(r_t)[pm], x[pm])
                else:
                    j = 1
                weights = N_t[ii - 1][:, j - 1] + dt[ii - 1][:, j - 1]
                s[ii] = np.cumsum(weights)
                sthres[ii] = s[ii][-1] * R.rand(t_max)
        ia = [k + 1 for k, v in enumerate(order_t) if (v == dyn).any()]   # intersect(...,'stable') positions
        for t in range(2, t_max + 1):
            delta_t += 1
            x_old = x.copy()
            for ii in ia:
                x[ii - 1] = int(np.nonzero(s[ii] >= sthres[ii][t - 1])[0][0]) + 1
            x[tm[:, 0] - 1] = x[tm[:, 1] - 1]
            if (x[:n_i] != x_old[:n_i]).any():
                for ii in range(1, n_i + 1):
                    if x[ii - 1] != x_old[ii - 1]:
                        events.append([delta_t, ii, x[ii - 1]])
                        delta_t = 0
    return initial, np.array(events, dtype=float).reshape(-1, 3)


def resample_events(initial, events, rates, R):
    # resample_events.m:10-37
    n = events.shape[0]
    newevents = []
    x = np.array(initial, dtype=float)
    rates = np.asarray(rates, dtype=float)
    for ii in range(n):
        holdtime = int(events[ii, 0])
        if holdtime == 0:
            newevents.append(list(events[ii]))
        else:
            delta_t = 0
            for _ in range(holdtime):
                changes = np.nonzero(R.rand(len(rates)) < rates)[0]
                delta_t += 1
                if len(changes):
                    for q, c in enumerate(changes):
                        newevents.append([delta_t if q == 0 else 0, c + 1, x[c]])
                    delta_t = 0
            newevents.append([delta_t, events[ii, 1], events[ii, 2]])
        if events[ii, 1] > 0:
            x[int(events[ii, 1]) - 1] = events[ii, 2]
    return np.array(newevents, dtype=float).reshape(-1, 3)


def dediscretize(d, parameters, zero_bins, R):
    # dediscretize.m:7-40 (scalar)
    if len(parameters) == 0:
        return d
    if zero_bins and zero_bins == d:
        return 0.0
    dd = int(d)
    a = parameters[dd - 1]
    b = parameters[dd]
    return a + (b - a) * R.rand()


def dbn_hierarchical_sample(p, di, dt, sample_time, dedisc_params, zero_bins, rates, start, R):
    # dbn_hierarchical_sample.m:9-37
    initial, events = dbn_sample(p, di, dt, sample_time, start, R)
    if events.shape[0] == 0:
        events = np.array([[sample_time, 0, 0]], dtype=float)
    else:
        events = np.vstack([events, [sample_time - events[:, 0].sum(), 0, 0]])
    events = resample_events(initial, events, rates, R)
    for ii in range(len(initial)):
        if len(dedisc_params[ii]) == p["N_initial"][ii].shape[0] - 2:
            pass
        else:
            initial[ii] = dediscretize(initial[ii], dedisc_params[ii], zero_bins[ii], R)
    for ii in range(events.shape[0] - 1):
        v = int(events[ii, 1])
        events[ii, 2] = dediscretize(events[ii, 2], dedisc_params[v - 1], zero_bins[v - 1], R)
    return initial, events


def events2samples(initial, events):
    # events2samples.m:9-26
    n = len(initial)
    d = np.zeros((n, int(events[:, 0].sum())))
    x = np.array(initial, dtype=float)
    t = 0
    for ev in events:
        delta_t = int(ev[0])
        if ev[1] == 0:
            t = t + 1
            d[:, t - 1: t + delta_t - 1] = x[:, None]
        else:
            if delta_t > 0:
                d[:, t: t + delta_t] = x[:, None]
                t = t + delta_t
            x[int(ev[1]) - 1] = ev[2]
    return d


def events2controls(initial, events, temporal_map):
    # events2controls.m:11-31
    vars_ = temporal_map[:, 0]
    x = np.array(initial, dtype=float)
    rows = []
    t = 0
    for ev in events:
        delta_t = ev[0]
        if delta_t > 0:
            rows.append([t] + list(x[vars_ - 1]))
            t = t + delta_t
        if ev[1] > 0:
            x[int(ev[1]) - 1] = ev[2]
    return np.array(rows, dtype=float).reshape(-1, 1 + len(vars_))


def uncor_sample(p, n_samples, sample_time, seed, start=None, prior=0.0):
    """UncorEncounterModel.m:192-313 with default options (no layers / quantize)."""
    R = Rand(seed)
    ni = p["n_initial"]
    labs = p["labels_initial"]
    idxV = labs.index('"v"'); idxDH = labs.index('"\\dot h"')
    Ni = [p["N_initial"][v] for v in range(ni)]
    Nt = [p["N_transition"].get(v) for v in range(p["n_transition"])]
    di = [np.full(N.shape, prior) for N in Ni]
    dt = [None if N is None else np.full(N.shape, prior) for N in Nt]
    pp = dict(p); pp["N_initial"] = Ni; pp["N_transition"] = Nt
    if start is None:
        start = [None] * ni
    zb = [int(z) for z in p["zero_bins"]]
    out = []
    for _ in range(n_samples):
        while True:
            initial, events = dbn_hierarchical_sample(pp, di, dt, sample_time, p["boundaries"], zb,
                                                      p["resample_rates"], start, R)
            if initial[idxV] * 1.68781 > abs(initial[idxDH]) / 60:
                break
        samples = events2samples(initial, events)
        controls = events2controls(initial, events, p["temporal_map"])
        out.append((initial, events, samples, controls))
    return out, R.count
