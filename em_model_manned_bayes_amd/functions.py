"""Function-level mirror of the reference's procedural API (code/matlab/*.m).

Same names and argument meaning as the MATLAB functions.  The sampling functions (bn_sample,
dbn_sample, dbn_hierarchical_sample) run on the GPU through libemgpu; the small deterministic
helpers (index math, priors, cut points, event-list formatting) are host-side numpy, as they are
host-side MATLAB in the reference.

MATLAB draws from one global rand stream; here every sampling call takes `seed` (Philox key) and
`first_index` (global index of the first sample).  With seed=None the module-level stream set by
rng(seed) is used and advanced, so successive calls give fresh samples.
"""
import numpy as np

from . import _lib as L
from . import native
from .em_io import Parms

# --------------------------------------------------------------------------------------------
# module-level stream (stand-in for MATLAB's global rng state)
# --------------------------------------------------------------------------------------------
_stream = {"seed": 0x5EED, "next_index": 0}


def rng(seed):
    """rng(seed,'twister') equivalent: reseed the module-level stream."""
    _stream["seed"] = int(seed) & (2**64 - 1)
    _stream["next_index"] = 0


def _take(seed, n):
    """(seed, first_index) for a call that consumes n sample indices."""
    if seed is None or (isinstance(seed, float) and np.isnan(seed)):
        first = _stream["next_index"]
        _stream["next_index"] += int(n)
        return _stream["seed"], first
    return int(seed) & (2**64 - 1), 0


# --------------------------------------------------------------------------------------------
# index helpers
# --------------------------------------------------------------------------------------------
def asub2ind(siz, x):
    """ndx = asub2ind(siz, x)  (asub2ind.m:13-14)"""
    siz = np.ascontiguousarray(np.asarray(siz, dtype=np.int32).reshape(-1))
    x = np.ascontiguousarray(np.asarray(x, dtype=np.int32).reshape(-1))
    return int(L.lib().emgpu_asub2ind(siz.ctypes.data, x.ctypes.data, siz.size))


def aind2sub(siz, ndx):
    """x = aind2sub(siz, ndx)  (aind2sub.m:8-18)"""
    siz = np.asarray(siz, dtype=np.int64).reshape(-1)
    k = np.concatenate([[1], np.cumprod(siz[:-1])])
    x = np.zeros(siz.size, dtype=np.int64)
    for ii in range(siz.size - 1, -1, -1):
        vi = (ndx - 1) % k[ii] + 1
        x[ii] = (ndx - vi) // k[ii] + 1
        ndx = vi
    return x


def bn_sort(G):
    """order = bn_sort(G)  (bn_sort.m:17-20; 'stable' taken as the lexicographically smallest order)."""
    G = np.asarray(G) != 0
    n = G.shape[0]
    m = native.NativeModel.from_arrays(G, np.full(n, 1, dtype=np.int32), [np.ones((1, 1)) for _ in range(n)])
    return m.get_i32(L.F_ORDER_INITIAL)


def discretize_bayes(x, thresholds):
    """d = discretize_bayes(x, thresholds)  (discretize_bayes.m:14-22)"""
    th = np.ascontiguousarray(np.asarray(thresholds, dtype=np.float64).reshape(-1))
    xs = np.asarray(x, dtype=np.float64)
    out = np.array([L.lib().emgpu_discretize_bayes(float(v), th.ctypes.data, th.size) for v in xs.reshape(-1)], dtype=np.float64)
    return out.reshape(xs.shape) if xs.shape else float(out[0])


def hierarchical_cutpoints(cutpoints_coarse, boundaries, n):
    """cutpoints_fine = hierarchical_cutpoints(cutpoints_coarse, boundaries, n)  (hierarchical_cutpoints.m:5-15)"""
    th = np.concatenate([[boundaries[0]], np.asarray(cutpoints_coarse, dtype=np.float64).reshape(-1), [boundaries[1]]])
    return [th[i - 1] + np.arange(1, n) * ((th[i] - th[i - 1]) / n) for i in range(1, len(th))]


def hierarchical_discretize(x, cutpoints_coarse, cutpoints_fine, zero_bins=None, wrap=0):
    """[d, repeat, change] = hierarchical_discretize(...)  (hierarchical_discretize.m:10-49)"""
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    d = np.asarray(discretize_bayes(x, cutpoints_coarse), dtype=np.float64).reshape(-1)
    if cutpoints_fine is None or len(cutpoints_fine) == 0:
        return d, 0, 0
    if wrap:
        d = 1 + np.mod(d - 1, len(cutpoints_coarse))
    zb = [] if zero_bins is None else list(np.atleast_1d(zero_bins))
    f = np.zeros(x.size)
    repeat = change = 0
    for ii in range(x.size):
        cf = np.asarray(cutpoints_fine[int(d[ii]) - 1], dtype=np.float64)
        f[ii] = cf.size + 1 if x[ii] >= cf[-1] else int(np.nonzero(x[ii] < cf)[0][0]) + 1
        if ii > 0 and d[ii] not in zb and d[ii - 1] == d[ii]:
            if f[ii - 1] == f[ii]:
                repeat += 1
            else:
                change += 1
    return d, repeat, change


# --------------------------------------------------------------------------------------------
# priors
# --------------------------------------------------------------------------------------------
def bn_dirichlet_prior(N, prior=0):
    """alpha = bn_dirichlet_prior(N, prior)  (bn_dirichlet_prior.m:18-37)"""
    alpha = []
    if isinstance(prior, str):
        if prior.lower() != "dbe":
            raise L.EmgpuError(L.ERR_PRIOR, "Unknown prior of %s, if char expecting prior = 'dbe'" % prior)
        for Ni in N:
            Ni = np.asarray(Ni)
            alpha.append(np.full(Ni.shape, 1.0 / Ni.size) if Ni.size else np.zeros(Ni.shape))
    elif isinstance(prior, (int, float, np.floating, np.integer)):
        for Ni in N:
            alpha.append(np.full(np.asarray(Ni).shape, float(prior)))
    else:
        raise L.EmgpuError(L.ERR_PRIOR, "Second argument must be a char or double. It was a %s" % type(prior).__name__)
    return alpha


def setTransitionPriors(G, r, temporal_map, prior):
    """alpha = setTransitionPriors(G, r, temporal_map, prior)  (setTransitionPriors.m:12-33)"""
    G = np.asarray(G) != 0
    r = np.asarray(r, dtype=np.int64).reshape(-1)
    tm = np.asarray(temporal_map, dtype=np.int64).reshape(-1, 2)
    alpha = [None] * G.shape[0]
    for ii in range(1, G.shape[0] + 1):
        rows = np.nonzero(tm[:, 1] == ii)[0]
        if rows.size and G[:, ii - 1].any():
            n = int(np.prod(r[G[:, ii - 1]]))
            jj = int(tm[rows[0], 0])
            a = np.zeros((int(r[jj - 1]), n))
            nn = n // int(r[jj - 1])
            for kk in range(1, int(r[jj - 1]) + 1):
                a[kk - 1, nn * (kk - 1): nn * kk] = prior
            alpha[ii - 1] = a
    return alpha


# --------------------------------------------------------------------------------------------
# event-list formatting (events2samples.m, events2controls.m) -- host-side, vectorised
# --------------------------------------------------------------------------------------------
def events2samples(initial, events):
    """d = events2samples(initial, events): n_initial x T  (events2samples.m:9-26)"""
    initial = np.asarray(initial, dtype=np.float64).reshape(-1)
    events = np.asarray(events, dtype=np.float64).reshape(-1, 3)
    T = int(events[:, 0].sum())
    d = np.zeros((initial.size, T))
    x = initial.copy()
    t = 0
    for dt, var, val in events:
        dt = int(dt)
        if var == 0:
            d[:, t: t + dt] = x[:, None]
        else:
            if dt > 0:
                d[:, t: t + dt] = x[:, None]
                t += dt
            x[int(var) - 1] = val
    return d


def events2controls(initial, events, mdl):
    """controls = events2controls(initial, events, mdl): rows [t, x(temporal_map(:,1))]  (events2controls.m:11-31)"""
    tm = np.asarray(mdl["temporal_map"] if isinstance(mdl, dict) else mdl.temporal_map).reshape(-1, 2)
    vars_ = tm[:, 0].astype(int) - 1
    x = np.asarray(initial, dtype=np.float64).reshape(-1).copy()
    events = np.asarray(events, dtype=np.float64).reshape(-1, 3)
    rows = []
    t = 0.0
    for dt, var, val in events:
        if dt > 0:
            rows.append(np.concatenate([[t], x[vars_]]))
            t += dt
        if var > 0:
            x[int(var) - 1] = val
    return np.array(rows, dtype=np.float64).reshape(-1, 1 + vars_.size)


# --------------------------------------------------------------------------------------------
# sampling functions (GPU)
# --------------------------------------------------------------------------------------------
def _model_of(parms, dirichlet_initial=None, dirichlet_transition=None, start=None):
    """Native model for a parms struct (em_read output / EncounterModel.struct()), with alpha and start applied."""
    if isinstance(parms, dict) and parms.get("native") is not None and dirichlet_initial is None and dirichlet_transition is None:
        m = parms["native"]
    else:
        nt = int(parms.get("n_transition", 0) or 0)
        m = native.NativeModel.from_arrays(
            parms["G_initial"], parms["r_initial"] if "r_initial" in parms else parms["r_transition"][: parms["n_initial"]],
            parms["N_initial"], parms.get("G_transition") if nt else None, parms.get("r_transition") if nt else None,
            parms.get("N_transition") if nt else None, parms.get("temporal_map") if nt else None,
            parms.get("boundaries"), parms.get("zero_bins"), parms.get("resample_rates"),
            parms.get("labels_initial"), parms.get("labels_transition"))
    if dirichlet_initial is not None:
        for v, a in enumerate(dirichlet_initial):
            if a is not None and np.asarray(a).size:
                m.set_f64(L.F_ALPHA_INITIAL, v + 1, np.asarray(a, dtype=np.float64).T.reshape(-1))
    if dirichlet_transition is not None:
        for v, a in enumerate(dirichlet_transition):
            if a is not None and np.asarray(a).size:
                m.set_f64(L.F_ALPHA_TRANSITION, v + 1, np.asarray(a, dtype=np.float64).T.reshape(-1))
    m.set_start(start if start is not None else [None] * m.n_initial)
    return m


def bn_sample(G, r, N, alpha, num_samples, start=None, order=None, seed=None, ctx=None):
    """S = bn_sample(G, r, N, alpha, num_samples, start, order)  (bn_sample.m:1): num_samples x n bins.
    `order` is recomputed by the library (bn_sort) and only checked for length."""
    n = len(N)
    if start is None:
        start = [None] * n
    assert len(start) == n and (order is None or len(order) == n)  # bn_sample.m:32-34
    r = np.asarray(r, dtype=np.int32).reshape(-1)
    m = native.NativeModel.from_arrays(G, r[:n], N)
    for v, a in enumerate(alpha):
        m.set_f64(L.F_ALPHA_INITIAL, v + 1, np.asarray(a, dtype=np.float64).T.reshape(-1))
    m.set_start(start)
    s, first = _take(seed, num_samples)
    ob, _, _ = native.sample_bn_host(ctx or native.default_context(), m, num_samples, s, first_index=first, dediscretize=False, max_attempts=1)
    return ob.astype(np.float64)


def dbn_sample(parms, dirichlet_initial, dirichlet_transition, t_max, start=None, seed=None, num_samples=1,
               transition_mode=L.TRANSITION_REFERENCE_AUTO, ctx=None):
    """[initial, events] = dbn_sample(parms, dirichlet_initial, dirichlet_transition, t_max, start)  (dbn_sample.m:1)
    events rows are (dt, variable, new bin).  num_samples > 1 returns lists."""
    m = _model_of(parms, dirichlet_initial, dirichlet_transition, start)
    s, first = _take(seed, num_samples)
    flags = L.FLAG_NO_RESAMPLE | L.FLAG_NO_DEDISC | L.FLAG_NO_TERMINATOR
    res = native.sample_dbn_host(ctx or native.default_context(), m, num_samples, int(t_max), s, first_index=first,
                                 want_dense=False, want_events=True, flags=flags, transition_mode=transition_mode,
                                 event_cap=m.n_initial * int(t_max) + 1, max_attempts=1)
    inits = res["init_bin"].astype(np.float64)
    evs = [np.stack([e["dt"].astype(np.float64), e["var"].astype(np.float64), e["bin"].astype(np.float64)], axis=1) for e in res["events"]]
    if num_samples == 1:
        return inits[0], evs[0]
    return inits, evs


def dbn_hierarchical_sample(parms, dirichlet_initial, dirichlet_transition, sample_time, dediscretize_parameters=None,
                            zero_bins=None, resample_rates=None, start=None, seed=None, num_samples=1,
                            transition_mode=L.TRANSITION_REFERENCE_AUTO, ctx=None):
    """[initial, events] = dbn_hierarchical_sample(parms, di, dt, sample_time, dediscretize_parameters, zero_bins,
    resample_rates, start)  (dbn_hierarchical_sample.m:1).  events rows: (dt, variable, dediscretised value),
    terminated by [dt 0 0]."""
    p = parms
    if dediscretize_parameters is not None or zero_bins is not None or resample_rates is not None:
        p = Parms({k: v for k, v in dict(parms).items() if k != "native"})
        if dediscretize_parameters is not None:
            p["boundaries"] = [np.asarray(b, dtype=np.float64).reshape(-1) if b is not None else np.zeros(0) for b in dediscretize_parameters]
        if zero_bins is not None:
            p["zero_bins"] = list(zero_bins)
        if resample_rates is not None:
            p["resample_rates"] = np.asarray(resample_rates, dtype=np.float64).reshape(-1)
    m = _model_of(p, dirichlet_initial, dirichlet_transition, start)
    s, first = _take(seed, num_samples)
    res = native.sample_dbn_host(ctx or native.default_context(), m, num_samples, int(sample_time), s, first_index=first,
                                 want_dense=False, want_events=True, transition_mode=transition_mode, max_attempts=1)
    inits = res["init_val"].astype(np.float64)
    evs = [np.stack([e["dt"].astype(np.float64), e["var"].astype(np.float64), e["value"].astype(np.float64)], axis=1) for e in res["events"]]
    if num_samples == 1:
        return inits[0], evs[0]
    return inits, evs
