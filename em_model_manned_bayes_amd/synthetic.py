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


def terminal_trajectory_model(seed, n_intent=3, reverse=False):
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

    def table(r_own, q_other, wrap):
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
        far = rs.randint(0, r_own, q)
        N[far, cols] += rs.randint(0, 6, q) * (rs.rand(q) < 0.15)
        empty = rs.rand(q) < 0.02          # unobserved parent configurations: all-zero columns (bin 1 without the prior)
        N[:, empty] = 0
        return N
    N_t = [np.zeros((0, 0))] * 6 + [table(36, 7 * 36, True), table(7, 7 * 36 * 36, False), table(5, 7 * 36 * 36, False)]
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
            em_io.em_write(terminal_trajectory_model(seed + k, n_intent, stem.endswith("reverse")), path)
    return out_dir
