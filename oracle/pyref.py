"""oracle/pyref.py -- second, independently written restatement (pure Python + numpy).

TEST INFRASTRUCTURE ONLY.  Small cases only (pure-Python loops).

Purpose: pin oracle/em_oracle.c.  This file follows the reference MATLAB
line by line with MATLAB-shaped data (cell arrays -> lists, matrices -> numpy)
and draws from numpy's MT19937 (numpy.random.RandomState(seed).random_sample()
== MATLAB rng(seed,'twister'); rand).  tests/test_oracle.py
(test_c_oracle_matches_numpy_restatement_draw_for_draw, test_terminal_restatements_agree_draw_for_draw, ...) requires the
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


# ================================================================================================
# Round 4: the terminal model and the legacy track builder, restated a second time
# (@CorTerminalModel/sample.m:29-77, createEncounter.m:52-329, CorTerminalModel.m:117-316, track.m:62-145, sample2track.m:183-243).
# MATLAB built-ins are restated where they are used: sind / cosd reduce the angle in degrees (n = round(x / 90)), wrapTo360 / wrapTo180 /
# wrapToPi as in the Mapping Toolbox, round(x, k) = round(x * 10^k) / 10^k, norm of a 2-vector = sqrt(x^2 + y^2).  em-core's functions
# (not vendored by the reference: UNPINNED) are the documented stand-ins: computeVerticalRate / computeHeadingRate = forward differences of
# the 1 s samples with the last value repeated (heading differences wrapped to (-pi, pi]), local_smooth = a centred moving average.
# ================================================================================================
import math


def _round_half_away(x):
    return math.floor(abs(x) + 0.5) * (1.0 if x >= 0 else -1.0)      # MATLAB round()


def sincosd(deg):
    n = _round_half_away(deg / 90.0)
    x = (math.pi / 180.0) * (deg - n * 90.0)
    m = math.fmod(n, 4.0)
    if m < 0:
        m += 4.0
    sx, cx = math.sin(x), math.cos(x)
    return {0.0: (sx, cx), 1.0: (cx, -sx), 2.0: (-sx, -cx), 3.0: (-cx, sx)}[m]


def wrapTo360(lon):
    positive = lon > 0
    lon = lon - math.floor(lon / 360.0) * 360.0
    return 360.0 if (lon == 0 and positive) else lon


def wrapTo180(lon):
    return wrapTo360(lon + 180.0) - 180.0 if (lon < -180.0 or 180.0 < lon) else lon


def wrapToPi(x):
    if x < -math.pi or math.pi < x:
        y = x + math.pi
        positive = y > 0
        y = y - math.floor(y / (2 * math.pi)) * (2 * math.pi)
        if y == 0 and positive:
            y = 2 * math.pi
        return y - math.pi
    return x


def atan2d(y, x):
    return math.atan2(y, x) * (180.0 / math.pi)


def discretize_bayes(x, thresholds):
    # discretize_bayes.m:14-22
    if x >= thresholds[-1]:
        return len(thresholds) + 1
    return int(np.nonzero(x < np.asarray(thresholds))[0][0]) + 1


def set_transition_priors(G_transition, r_transition, temporal_map, prior, N_transition):
    # setTransitionPriors.m:12-33: alpha{ii}(kk, n(kk-1)+1 : n kk) = prior, n = q / r_jj; zeros elsewhere and for nodes without parents
    alpha = [None if N is None else np.zeros(N.shape) for N in N_transition]
    for kk_ in range(temporal_map.shape[0]):
        jj, ii = int(temporal_map[kk_, 0]), int(temporal_map[kk_, 1])
        if not G_transition[:, ii - 1].any():
            continue
        r = int(r_transition[jj - 1])
        q = alpha[ii - 1].shape[1]
        n = q // r
        for kk in range(1, r + 1):
            alpha[ii - 1][kk - 1, n * (kk - 1): n * kk] = prior
    return alpha


def geom_sample(p, n_samples, seed, bounds_sample=None, lim1=(0.0, np.inf), lim2=(0.0, np.inf), start=None, prior=0.0, R=None):
    """@CorTerminalModel/sample.m:29-77: bn_sample on the 15-variable geometry network, dediscretize, box and speed rejection."""
    R = R or Rand(seed)
    ni = p["n_initial"]
    labs = [s.strip('"') for s in p["labels_initial"]]
    Ni = [p["N_initial"][v] for v in range(ni)]
    di = [np.full(N.shape, prior) for N in Ni]
    if start is None:
        start = [None] * ni
    out = np.zeros((n_samples, ni))
    for ii in range(n_samples):
        while True:
            initial = bn_sample(p["G_initial"], p["r_initial"], Ni, di, start, p["order_initial"], R)          # :34
            for kk in range(ni):                                                                                # :37-42
                if len(p["boundaries"][kk]):
                    zb = int(p["zero_bins"][kk])
                    initial[kk] = dediscretize(initial[kk], p["boundaries"][kk], zb, R)
            is_good = True
            if bounds_sample is not None:                                                                       # :45-53
                is_good = bool(np.all((initial >= bounds_sample[:, 0]) & (initial <= bounds_sample[:, 1])))
            if is_good:                                                                                         # :56-70
                sample = dict(zip(labs, initial))
                s1 = sample["own_speed"] <= lim1[1] and sample["own_speed"] >= lim1[0]
                s2 = sample["int_speed"] <= lim2[1] and sample["int_speed"] >= lim2[0]
                is_good = bool(s1 and s2)
            if is_good:
                break
        out[ii] = initial
    return out, R


def _traj_struct(p):
    """What createEncounter.m:133-142 casts a trajectory model to (struct(mdl)) + the derived properties it reads."""
    ni = p["n_initial"]
    q = dict(p)
    q["N_initial"] = [p["N_initial"][v] for v in range(ni)]
    q["N_transition"] = [p["N_transition"].get(v) for v in range(p["n_transition"])]
    q["cutpoints_initial"] = [np.asarray(b[1:-1], dtype=float) if len(b) else np.zeros(0) for b in p["boundaries"]]   # EncounterModel.m:325-339
    q["bounds_initial"] = [(b[0], b[-1]) if len(b) else (np.nan, np.nan) for b in p["boundaries"]]
    return q


def propagate_trajectory(p, is_ownship, dt_s, x0_nm, y0_nm, z0_ft, v0_ft_s, heading0_deg, intent, tmax_s, dynlims, R):
    """PropagateTrajectory, createEncounter.m:93-265.  dynlims: dict with minVel_ft_s maxVel_ft_s maxTurnRate_deg_s maxAltitude_ft
    maxVertRate_ft_s.  Returns dict of lists t_s x_nm y_nm z_ft heading_deg v_ft_s."""
    mdl = _traj_struct(p)
    x0_nm, y0_nm, z0_ft, v0_ft_s, heading0_deg = float(x0_nm), float(y0_nm), float(z0_ft), float(v0_ft_s), float(heading0_deg)
    labs = p["labels_initial"]
    assert labs[3] == '"heading"' and labs[4] == '"altitude"' and labs[5] == '"speed"'                          # :107-109
    idx = {k: labs.index('"%s"' % n_) for k, n_ in (("int", "intent"), ("dist", "distance"), ("bear", "bearing"), ("head", "heading"),
                                                     ("alt", "altitude"), ("spd", "speed"))}
    dp = p["boundaries"]
    le = np.nonzero(np.asarray(dp[idx["alt"]]) <= dynlims["maxAltitude_ft"])[0]                                  # :121
    valid_alt = list(range(1, int(le[-1]) + 2)) if len(le) else []
    ge = np.nonzero(~(np.asarray(dp[idx["spd"]]) >= dynlims["minVel_ft_s"]))[0]                                  # :124-126
    le = np.nonzero(np.asarray(dp[idx["spd"]]) <= dynlims["maxVel_ft_s"])[0]
    valid_v = list(range(int(ge[-1]) + 1, int(le[-1]) + 2)) if (len(ge) and len(le)) else []                     # s:1:e with s = [] is empty
    prior_initial = [np.zeros(N.shape) for N in mdl["N_initial"]]                                                # :128
    prior_transition = set_transition_priors(p["G_transition"], p["r_transition"], p["temporal_map"], 1.0, mdl["N_transition"])   # :129
    t_s, ii = 0.0, 1
    xy = [x0_nm, y0_nm]
    s0, c0 = sincosd(heading0_deg)
    v = [c0 * v0_ft_s - s0 * 0.0, s0 * v0_ft_s + c0 * 0.0]                                                       # :145
    z_ft, heading_deg = z0_ft, heading0_deg
    traj = {k: [] for k in ("t_s", "x_nm", "y_nm", "z_ft", "heading_deg", "v_ft_s")}
    is_resample = True
    while is_resample:                                                                                           # :160
        traj["t_s"].append(t_s); traj["x_nm"].append(xy[0]); traj["y_nm"].append(xy[1]); traj["z_ft"].append(z_ft)
        traj["heading_deg"].append(heading_deg); traj["v_ft_s"].append(math.sqrt(v[0] * v[0] + v[1] * v[1]))
        xy = [xy[0] + v[0] * dt_s / 6076.1154855643, xy[1] + v[1] * dt_s / 6076.1154855643]                      # :170-173
        curr_hdg = wrapTo360(atan2d(v[1], v[0]))                                                                 # :176-177
        traj["heading_deg"][ii - 1] = curr_hdg
        if ii > 1:                                                                                               # :180-184
            alt_diff = z_ft - traj["z_ft"][ii - 2]
            sgn = float(alt_diff > 0) - float(alt_diff < 0)
            traj["z_ft"][ii - 1] = traj["z_ft"][ii - 2] + sgn * min(dynlims["maxVertRate_ft_s"], abs(alt_diff))
        cp = mdl["cutpoints_initial"]                                                                            # :187, CreateStartDistribution :268-294
        heading_discrete = discretize_bayes(heading_deg, cp[idx["head"]])
        start = [intent, discretize_bayes(math.sqrt(xy[0] * xy[0] + xy[1] * xy[1]), cp[idx["dist"]]),
                 discretize_bayes(wrapTo360(atan2d(xy[1], xy[0])), cp[idx["bear"]]), heading_discrete,
                 discretize_bayes(z_ft, cp[idx["alt"]]), discretize_bayes(math.sqrt(v[0] * v[0] + v[1] * v[1]), cp[idx["spd"]])]
        is_resample = True
        while is_resample:                                                                                       # :192
            _, events = dbn_sample(mdl, prior_initial, prior_transition, 2, start, R)                            # :193
            is_resample = False
            for jj in range(events.shape[0]):                                                                    # :198
                var, val = int(events[jj, 1]), int(events[jj, 2])
                if var == 4:
                    if val != heading_discrete:
                        heading_deg = dediscretize(val, dp[idx["head"]], 0, R)
                    is_resample = False
                elif var == 5:
                    if val in valid_alt:
                        z_ft = dediscretize(val, dp[idx["alt"]], 0, R)
                        is_resample = False
                    else:
                        is_resample = True
                elif var == 6:
                    if val in valid_v:
                        v0 = dediscretize(val, dp[idx["spd"]], 0, R)
                        v0 = max(v0, dynlims["minVel_ft_s"]) if v0 < dynlims["minVel_ft_s"] else v0
                        v0 = dynlims["maxVel_ft_s"] if v0 > dynlims["maxVel_ft_s"] else v0
                        sh, ch = sincosd(heading_deg)
                        v = [ch * v0 - sh * 0.0, sh * v0 + ch * 0.0]
                        is_resample = False
                    else:
                        is_resample = True
                if is_resample:
                    break
        turn1 = heading_deg - curr_hdg                                                                           # :242
        turn1 = _round_half_away(turn1 * 100.0) / 100.0
        sgn = float(turn1 > 0) - float(turn1 < 0)
        delta = min(abs(turn1), dynlims["maxTurnRate_deg_s"]) * sgn
        sd, cd = sincosd(delta)
        v = [cd * v[0] - sd * v[1], sd * v[0] + cd * v[1]]                                                       # :256
        t_s = t_s + dt_s
        ii += 1
        d_nm = math.sqrt(xy[0] * xy[0] + xy[1] * xy[1])                                                          # CheckTrajectoryConditions :296-329
        violate = [abs(t_s) > tmax_s, d_nm > mdl["bounds_initial"][idx["dist"]][1],
                   d_nm <= 0.25 if intent in (1, 2) else False, bool(is_ownship) and xy[1] > 0.25]
        is_resample = not any(violate)
    return traj


def local_smooth(x, w):
    # the stand-in for em-core's local_smooth (UNPINNED): centred moving average, the window shrunk symmetrically at the ends
    x = np.asarray(x, dtype=float)
    h = (w - 1) // 2
    out = np.zeros(len(x))
    for i in range(len(x)):
        k = min(h, i, len(x) - 1 - i)
        acc = 0.0
        for q in range(i - k, i + k + 1):
            acc += x[q]
        out[i] = acc / (2 * k + 1)
    return out


def create_encounter_inputs(sample_geo):
    """createEncounter.m:13-49: what the four PropagateTrajectory calls of one encounter are handed.  sample_geo: dict by label (one geometry
    sample).  Returns (geo[12] = x0_nm y0_nm z0_ft v0_ft_s heading0_deg intent for aircraft 1, then 2 (:41-45, :60,:67),
    model_of[4] = [own fwd, own bck, int fwd, int bck] as positions in CorTerminalModel.m:84-100's list of ten (:13-38))."""
    geo = []
    for pre in ("own", "int"):
        s, c = sincosd(sample_geo[pre + "_bearing"])
        geo += [sample_geo[pre + "_distance"] * c, sample_geo[pre + "_distance"] * s, sample_geo[pre + "_alt"], sample_geo[pre + "_speed"],
                sample_geo[pre + "_heading"], float(int(sample_geo[pre + "_intent"]))]
    oi, ii = int(sample_geo["own_intent"]), int(sample_geo["int_intent"])
    if oi not in (1, 2):
        raise ValueError("Unknown own_intent = %d" % oi)                  # :21-22
    if ii not in (1, 2, 3):
        raise ValueError("Unknown int_intent = %d" % ii)                  # :36-37
    return geo, [2 * (oi - 1), 2 * (oi - 1) + 1, 4 + 2 * (ii - 1), 4 + 2 * (ii - 1) + 1]


def create_encounter(traj_models, model_of, sample_geo, tmax_s, dynlims_pair, R, smooth=False):
    """createEncounter.m:40-91.  traj_models: the 10 parsed trajectory models; model_of: the 4 indices [own fwd, own bck, int fwd, int bck]
    (createEncounter.m:21-38 picks them by intent); sample_geo: dict by label.  Returns [own, int] dicts of numpy arrays sorted in time."""
    geo = []
    for pre in ("own", "int"):
        s, c = sincosd(sample_geo[pre + "_bearing"])
        geo.append((sample_geo[pre + "_distance"] * c, sample_geo[pre + "_distance"] * s, sample_geo[pre + "_alt"], sample_geo[pre + "_speed"],
                    sample_geo[pre + "_heading"], int(sample_geo[pre + "_intent"])))
    out = []
    for a in range(2):
        x0, y0, z0, v0, h0, intent = geo[a]
        fwd = propagate_trajectory(traj_models[model_of[2 * a]], a == 0, 1.0, x0, y0, z0, v0, h0, intent, tmax_s, dynlims_pair[a], R)       # :71
        bck = propagate_trajectory(traj_models[model_of[2 * a + 1]], a == 0, -1.0, x0, y0, z0, v0, h0, intent, tmax_s, dynlims_pair[a], R)  # :72
        tr = {k: np.array(fwd[k] + bck[k][1:], dtype=float) for k in fwd}                                                                    # :75-78
        order = np.argsort(tr["t_s"], kind="stable")                                                                                       # :81-84
        tr = {k: v[order] for k, v in tr.items()}
        if smooth:                                                                                                                           # :88-89
            tr["v_ft_s"] = local_smooth(tr["v_ft_s"], 5)
            tr["z_ft"] = local_smooth(tr["z_ft"], 15)
        out.append(tr)
    return out


# ---- CorTerminalModel.m:117-316 and the filters of track.m:62-145
def get_generated_miss_distance(traj):
    common = np.intersect1d(traj[0]["t_s"], traj[1]["t_s"])                                                      # :122
    ia = np.array([int(np.nonzero(traj[0]["t_s"] == t)[0][0]) for t in common], dtype=int)
    ib = np.array([int(np.nonzero(traj[1]["t_s"] == t)[0][0]) for t in common], dtype=int)
    dx = traj[0]["x_nm"][ia] - traj[1]["x_nm"][ib]
    dy = traj[0]["y_nm"][ia] - traj[1]["y_nm"][ib]
    dxy_ft = np.sqrt(dx * dx + dy * dy) * 6076.1154855643
    dz_ft = traj[1]["z_ft"][ib] - traj[0]["z_ft"][ia]
    k = int(np.argmin(dxy_ft))
    return dxy_ft[k], dz_ft[k], traj[0]["t_s"][ia[k]], len(common)


def check_cum_turn(heading, limit):
    # CorTerminalModel.m:135-185
    heading = np.array([wrapTo180(h) for h in heading])
    hd = np.array([_round_half_away(d * 10.0) / 10.0 for d in np.diff(heading)])
    if len(hd) == 0:
        return False
    turn_start = [int(i) + 2 for i in np.nonzero((hd[:-1] == 0) & (hd[1:] != 0))[0]]                            # find(...) + 1, 1-based
    turn_end = [int(i) + 1 for i in np.nonzero((hd[:-1] != 0) & (hd[1:] == 0))[0]]
    if not turn_start:
        turn_start = [1]
    if not turn_end:
        turn_end = [len(hd)]
    if len(turn_start) > len(turn_end):
        turn_end.append(len(hd))
    for i in range(len(turn_start)):
        hs = np.array([wrapTo180(h) for h in hd[turn_start[i] - 1: turn_end[i]]])
        if len(hs) == 0:
            continue
        sg = np.sign(hs)
        start_idx = [int(k) + 1 for k in np.nonzero(np.concatenate([[0], np.diff(sg)]) != 0)[0]]                 # 1-based
        if not start_idx:
            if np.any(np.abs(np.cumsum(hs)) > limit):
                return True
        else:
            pts = sorted(set([1] + start_idx + [len(hs) + 1]))
            for j in range(len(pts) - 1):
                if np.any(np.abs(np.cumsum(hs[pts[j] - 1: pts[j + 1] - 1])) > limit):
                    return True
    return False


def _forward_rate(z):
    z = np.asarray(z, dtype=float)
    if len(z) == 1:
        return np.zeros(1)
    d = np.diff(z)
    return np.concatenate([d, d[-1:]])


def check_runway_proximity(tr, thres_dist_ft, thres_alt_low_ft):
    d_ft = np.hypot(tr["x_nm"], tr["y_nm"]) * 1.68781                                                            # :213-215 (sic: nm -> ft by 1.68781)
    close = d_ft <= thres_dist_ft
    low = tr["z_ft"][close] <= thres_alt_low_ft if close.any() else np.zeros(0, dtype=bool)
    return bool(close.any()), bool(low.any())


def check_intent_vertical(traj, thres_vertrate_ft_s):
    climb, descend = [], []
    for tr in traj:                                                                                              # :239-264
        n = len(tr["t_s"])
        dh = _forward_rate(tr["z_ft"])
        thr_time = (tr["z_ft"].max() - tr["z_ft"].min()) / thres_vertrate_ft_s
        pth = min(0.2, thr_time / n)
        climb.append(np.count_nonzero(dh >= thres_vertrate_ft_s) / len(dh) >= pth)
        descend.append(np.count_nonzero(dh <= -thres_vertrate_ft_s) / len(dh) >= pth)
    return climb, descend


def check_dynamic_limits(tr, dl, max_cum_turn_deg, pitch_deg):
    n = len(tr["t_s"])                                                                                           # :268-316
    if n <= 1:
        return False
    z, v = tr["z_ft"], tr["v_ft_s"]
    is_alt = (z > 0) & (z <= dl["maxAltitude_ft"])
    is_spd = (v >= dl["minVel_ft_s"]) & (v <= dl["maxVel_ft_s"])
    is_vr = np.abs(_forward_rate(z)) <= dl["maxVertRate_ft_s"]
    hr = tr["heading_deg"] * (math.pi / 180.0)
    rate = np.array([wrapToPi(hr[i + 1] - hr[i]) for i in range(n - 1)])
    rate = np.concatenate([rate, rate[-1:]])
    is_turn = np.abs(rate) <= dl["maxTurnRate_deg_s"]                                                            # (sic: rad/s against a deg/s limit, :296)
    ratio = np.abs(np.diff(z)) / v[1:]
    with np.errstate(invalid="ignore"):
        is_pitch = np.array([abs(math.asin(r) * (180.0 / math.pi)) <= pitch_deg if r <= 1 else pitch_deg == np.inf for r in ratio])
    is_pitch = np.concatenate([[True], is_pitch])
    is_cum = not check_cum_turn(tr["heading_deg"], max_cum_turn_deg)
    return bool(np.all(is_alt & is_spd & is_vr & is_turn & is_pitch)) and is_cum


def terminal_filters(traj, own_intent, int_intent, dl_pair, max_cum_turn_deg, pitch_deg, min_enc_time_s=30.0, thres_dist_ft=2.5 * 6076,
                     thres_alt_low_ft=750.0, thres_vertrate_ft_s=5.0):
    """track.m:79-145 (`isClimb`, undefined there, read as is_climb).  Returns (is_good, (tcpa_s, hmd_ft, vmd_ft, enc_time_s))."""
    if len(np.intersect1d(traj[0]["t_s"], traj[1]["t_s"])) == 0:
        return False, (0.0, 0.0, 0.0, 0.0)
    hmd, vmd, tcpa, enc_time = get_generated_miss_distance(traj)
    meta = (tcpa, hmd, vmd, float(enc_time))
    if not abs(tcpa) <= 10:                                                                                      # :84-87
        return False, meta
    is_long = enc_time >= min_enc_time_s                                                                         # :90-91
    c1, l1 = check_runway_proximity(traj[0], thres_dist_ft, thres_alt_low_ft)
    c2, l2 = check_runway_proximity(traj[1], thres_dist_ft, thres_alt_low_ft)
    prox1 = (c1 and l1) or not c1
    prox2 = (not (c2 and l2)) if int_intent == 3 else ((c2 and l2) or not c2)
    climb, descend = check_intent_vertical(traj, thres_vertrate_ft_s)
    int_ok = descend[1] if int_intent == 1 else (climb[1] if int_intent == 2 else True)
    own_ok = False
    if own_intent in (1, 2):
        c = 90.0 if own_intent == 1 else 270.0
        h = traj[0]["heading_deg"]
        pct = np.count_nonzero((h >= c - 30) & (h <= c + 30)) / len(h)
        own_ok = (descend[0] if own_intent == 1 else climb[0]) and pct >= .95
    d1 = check_dynamic_limits(traj[0], dl_pair[0], max_cum_turn_deg[0], pitch_deg[0])
    d2 = check_dynamic_limits(traj[1], dl_pair[1], max_cum_turn_deg[1], pitch_deg[1])
    return bool(is_long and prox1 and prox2 and own_ok and int_ok and d1 and d2), meta


# ---- sample2track.m:183-243
def sample2track(alt0, speed0, updates, ur_speed, ur_vertrate, ur_heading, min_speed, max_speed):
    """One id: alt0 (ft), speed0 (model units), updates [T, 3] = vertical rate, acceleration, turn rate (model units).
    Returns (x_ft, y_ft, z_ft arrays of T + 1, is_cfit, is_reject_speed)."""
    min_speed = min_speed * ur_speed; max_speed = max_speed * ur_speed                                           # :138-139
    z = [alt0]; sp = [speed0 * ur_speed]; hd = [0.0]; x = [0.0]; y = [0.0]                                       # :184-189, :126
    for pInd in range(updates.shape[0]):                                                                         # :198-216
        uv, ua, ut = updates[pInd, 0] * ur_vertrate, updates[pInd, 1] * ur_speed, updates[pInd, 2] * ur_heading  # :131-133
        z.append(z[pInd] + uv); sp.append(sp[pInd] + ua); hd.append(hd[pInd] + ut)
        s, c = sincosd(hd[pInd])
        x.append(x[pInd] + sp[pInd] * c); y.append(y[pInd] + sp[pInd] * s)
    z = np.array(z); sp = np.array(sp)
    return np.array(x), np.array(y), z, bool(np.any(z < 0)), bool(np.any((sp <= min_speed) | (sp >= max_speed)))
