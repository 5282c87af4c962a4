"""Deterministic synthetic stand-ins for model files that are absent from the reference mount
(the reference repository's .MISSING_LARGE_BLOBS): the 20 terminal trajectory models
terminal_v3_*_{ownship,intruder}_{landing,takeoff,transit}_model[_reverse].txt.

Structure from doc/model_terminal_traj_fwd.png / _bck.png and createEncounter.m:107-109,:293
(SURVEY.md section 8 a15): initial variables {intent, distance, bearing, heading, altitude, speed},
  heading(t+1)  <- distance, bearing, heading(t)
  altitude(t+1) <- distance, bearing, heading(t), altitude(t)
  speed(t+1)    <- distance, bearing, heading(t), speed(t)
with the bin boundaries of the shipped geometry model (terminal_v3_*_encounter_model.txt:29-33).
Counts are synthetic ("stay" mass plus neighbours plus sparse noise); they exercise the code path,
they are NOT the trained models.
"""
import numpy as np

from .em_io import Parms

_BND = {
    "distance": np.array([0, 0.5, 1, 2, 3, 4, 5, 8], dtype=np.float64),
    "bearing": np.arange(0, 361, 10, dtype=np.float64),
    "heading": np.arange(0, 361, 10, dtype=np.float64),
    "altitude": np.array([200, 500, 1000, 1500, 2000, 2500, 3000, 5000], dtype=np.float64),
    "speed": np.array([75, 150, 225, 300, 375, 450], dtype=np.float64),
}


def terminal_trajectory_model(seed, n_intent=3, reverse=False, alt_drift=0):
    """alt_drift: -1 / +1 adds mass to the next lower / higher altitude bin per step (a landing descends in forward time and climbs
    in backward time, a take-off the other way), so that the vertical-intent filters of CorTerminalModel.track see plausible tracks."""
    rs = np.random.RandomState(seed)
    labels_i = ['"intent"', '"distance"', '"bearing"', '"heading"', '"altitude"', '"speed"']
    tag = "(t-1)" if reverse else "(t+1)"
    labels_t = ['"intent"', '"distance"', '"bearing"', '"heading(t)"', '"altitude(t)"', '"speed(t)"',
                '"heading%s"' % tag, '"altitude%s"' % tag, '"speed%s"' % tag]
    r_i = np.array([n_intent, 7, 36, 36, 7, 5], dtype=np.int32)
    r_t = np.concatenate([r_i, [36, 7, 5]]).astype(np.int32)
    G_i = np.zeros((6, 6), dtype=bool)
    G_t = np.zeros((9, 9), dtype=bool)
    G_t[[1, 2, 3], 6] = True          # heading(t+1)
    G_t[[1, 2, 3, 4], 7] = True       # altitude(t+1)
    G_t[[1, 2, 3, 5], 8] = True       # speed(t+1)
    N_i = [np.ones((int(r), 1)) for r in r_i]

    def table(r_own, q_other, wrap, drift=0):
        # columns: own variable is the slowest-varying parent (setTransitionPriors.m:20-27 relies on it)
        q = q_other * r_own
        own = np.repeat(np.arange(r_own), q_other)
        N = np.zeros((r_own, q))
        cols = np.arange(q)
        N[own, cols] = rs.randint(60, 400, q)
        for d in (-1, 1):
            nb = own + d
            nb = np.mod(nb, r_own) if wrap else np.clip(nb, 0, r_own - 1)
            N[nb, cols] += rs.randint(0, 30, q) * (rs.rand(q) < 0.7)
        if drift:
            nb = np.clip(own + drift, 0, r_own - 1)
            N[nb, cols] += rs.randint(40, 160, q)
        far = rs.randint(0, r_own, q)
        N[far, cols] += rs.randint(0, 6, q) * (rs.rand(q) < 0.15)
        empty = rs.rand(q) < 0.02          # unobserved parent configurations: all-zero columns (bin 1 without the prior)
        N[:, empty] = 0
        return N
    N_t = [np.zeros((0, 0))] * 6 + [table(36, 7 * 36, True), table(7, 7 * 36 * 36, False, alt_drift), table(5, 7 * 36 * 36, False)]
    p = Parms(labels_initial=labels_i, n_initial=6, G_initial=G_i, r_initial=r_i, N_initial=N_i,
              labels_transition=labels_t, n_transition=9, G_transition=G_t, r_transition=r_t, N_transition=N_t,
              boundaries=[np.zeros(0), _BND["distance"], _BND["bearing"], _BND["heading"], _BND["altitude"], _BND["speed"]],
              resample_rates=np.zeros(6))
    return p


# file stems of CorTerminalModel.m:60 in the order (own landing, own takeoff, int landing, int takeoff, int transit) x (fwd, bck)
TERMINAL_FILE_STEMS = ["ownship_landing_model", "ownship_landing_model_reverse", "ownship_takeoff_model", "ownship_takeoff_model_reverse",
                       "intruder_landing_model", "intruder_landing_model_reverse", "intruder_takeoff_model", "intruder_takeoff_model_reverse",
                       "intruder_transit_model", "intruder_transit_model_reverse"]


def write_terminal_directory(out_dir, src="terminalradar", seed=0x5EED0005):
    """Materialise a correlated_terminal/<src> directory: the shipped geometry model plus ten synthetic
    trajectory models, as reference-format .txt files.  Returns the directory."""
    import os
    from . import em_io
    os.makedirs(out_dir, exist_ok=True)
    name = {"terminalradar": "terminal_v3_radar_encounter_model", "opensky": "terminal_v3_opensky_encounter_model"}[src]
    em_io.materialize_model(name, out_dir)
    prefix = name.replace("encounter_model", "")
    for k, stem in enumerate(TERMINAL_FILE_STEMS):
        path = os.path.join(out_dir, prefix + stem + ".txt")
        if not os.path.exists(path):
            n_intent = 2 if stem.startswith("ownship") else 3
            rev = stem.endswith("reverse")
            drift = (-1 if "landing" in stem else (1 if "takeoff" in stem else 0)) * (-1 if rev else 1)
            em_io.em_write(terminal_trajectory_model(seed + k, n_intent, rev, drift), path)
    return out_dir


def correlated_v2p1_like(seed=0x5EED0003):
    """A deterministic stand-in for model/cor_v2p1.txt (absent from the reference mount: .MISSING_LARGE_BLOBS:1) for
    BASELINE.json configs[2]: the correlated two-aircraft network of cor_v1 -- same 16 initial
    variables, same initial graph and bins, the 4 dynamic variables \\dot h_1, \\dot h_2, \\dot\\psi_1, \\dot\\psi_2 -- with the
    transition tables ALSO conditioned on the airspace class A (cor_v1: L only), i.e. 4x more columns, and counts redrawn from
    `seed`: a "stay" mass plus neighbours plus sparse far jumps, 3 % unobserved (all-zero) columns.  Dependent branch like cor_v1
    (\\dot\\psi_k(t+1) has \\dot h_k(t+1) as a parent).  Synthetic counts: it exercises the path, it is not a trained model."""
    import os
    from . import em_io
    base = em_io.load_npz(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "models", "cor_v1.npz"))
    rs = np.random.RandomState(seed & 0x7FFFFFFF)
    ni, nt = base["n_initial"], base["n_transition"]
    r_i = np.asarray(base["r_initial"], dtype=np.int32)
    r_t = np.asarray(base["r_transition"], dtype=np.int32)
    G_i = np.asarray(base["G_initial"], dtype=bool)
    G_t = np.asarray(base["G_transition"], dtype=bool).copy()
    G_t[0, ni:nt] = True                                   # + airspace class A (variable 1) as a parent of every (t+1) node

    def initial_counts(v):
        q = int(np.prod(r_i[G_i[:, v]])) if G_i[:, v].any() else 1
        N = rs.gamma(0.6, 400.0, (int(r_i[v]), q)).round()
        N *= rs.rand(int(r_i[v]), q) < 0.8
        N[rs.randint(int(r_i[v])), :] += 50                # no empty initial column: every configuration can be drawn
        return N
    N_i = [initial_counts(v) for v in range(ni)]

    tmap = {ni + k: 10 + k for k in range(nt - ni)}       # (t+1) node -> its (t) node, 0-based: cor_v1.txt temporal map [11 17; 12 18; 13 19; 14 20]

    def transition_counts(v):
        par = np.flatnonzero(G_t[:, v])
        q = int(np.prod(r_t[par]))
        rv = int(r_t[v])
        own = tmap[v]                                      # the variable's own value at t
        stride = int(np.prod(r_t[par[par < own]]))         # asub2ind: earlier parents vary faster
        cur = (np.arange(q) // stride) % int(r_t[own])
        N = np.zeros((rv, q))
        cols = np.arange(q)
        N[cur, cols] = rs.randint(2000, 60000, q)
        for d in (-1, 1):
            nb = np.clip(cur + d, 0, rv - 1)
            N[nb, cols] += rs.randint(0, 900, q) * (rs.rand(q) < 0.8)
        for d in (-2, 2):
            nb = np.clip(cur + d, 0, rv - 1)
            N[nb, cols] += rs.randint(0, 60, q) * (rs.rand(q) < 0.3)
        N[rs.randint(0, rv, q), cols] += rs.randint(0, 5, q) * (rs.rand(q) < 0.1)
        N[:, rs.rand(q) < 0.03] = 0
        return N
    N_t = [np.zeros((0, 0))] * ni + [transition_counts(v) for v in range(ni, nt)]
    return Parms(labels_initial=list(base["labels_initial"]), n_initial=ni, G_initial=G_i, r_initial=r_i, N_initial=N_i,
                 labels_transition=list(base["labels_transition"]), n_transition=nt, G_transition=G_t, r_transition=r_t, N_transition=N_t,
                 boundaries=[np.asarray(b, dtype=np.float64) for b in base["boundaries"]],
                 resample_rates=np.asarray(base["resample_rates"], dtype=np.float64))


def write_correlated_v2p1_like(out_dir, seed=0x5EED0003):
    """Materialise correlated_v2p1_like as <out_dir>/cor_v2p1_like.txt in the reference's file format; returns the path."""
    import os
    from . import em_io
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "cor_v2p1_like.txt")
    if not os.path.exists(path):
        em_io.em_write(correlated_v2p1_like(seed), path)
    return path
