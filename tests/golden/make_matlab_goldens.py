"""tests/golden/make_matlab_goldens.py -- the text fixtures tools/matlab/emgpu_check_parity.m reads.

The reference is MATLAB-only and cannot run in the build image, so parity against real MATLAB is
unpinned.  These files are what closes that gap on a machine that HAS
MATLAB + the reference checkout: one command there compares the reference's own output with the
answers of this repository's CPU oracle in MT19937 mode (rng(seed,'twister') == numpy RandomState).
Everything is written as plain CSV with %.17g so MATLAB's readmatrix / dlmread round-trips exactly.

  config1_inits.csv / config1_events.csv     mdl.sample(100, 120, 'seed', 1) on uncor_1200code_v2p1
                                             (BASELINE.json configs[0]; fast branch + rejection loop)
  hier_<model>_*.csv                         a plain loop of dbn_hierarchical_sample calls on one stream for
                                             uncor_1200code_v1 (dependent branch, non-identity order_transition)
                                             and cor_v1 through em_read (16 variables, non-identity order_initial)
  bn_sort_orders.csv                         topological orders of every non-upper-triangular graph shipped
  trig_table.csv                             sind / cosd / wrapTo360 / atan2d as restated for createEncounter.m
  geom_<srcData>_inits.csv                   CorTerminalModel('srcData', s).sample(500, 'seed', 1) (RUN_terminal.m:39 without a start) and the
  geom_<srcData>_start_inits.csv             same with mdl.start = {2, 1, 3, [], ...} (a row of InitStartTerminal): the 15-variable geometry
                                             network, dediscretize and the GENERIC speed rejection (@CorTerminalModel/sample.m:29-77)

The oracle is cross-checked draw-for-draw against the second restatement (oracle/pyref.py) before
anything is written.
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402
import pyref as P  # noqa: E402
from em_model_manned_bayes_amd import em_io  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "matlab")


def save(name, a, header):
    np.savetxt(os.path.join(OUT, name), np.atleast_2d(a), fmt="%.17g", delimiter=",", header=header, comments="% ")


def events_table(evs):
    """rows [sample(1-based) dt var value]"""
    return np.concatenate([np.column_stack([np.full(len(e), i + 1.0), e[:, :3]]) for i, e in enumerate(evs)])


def hier_loop_pyref(pp, n, T, seed):
    R = P.Rand(seed)
    ni = pp["n_initial"]
    Ni = [pp["N_initial"][v] for v in range(ni)]
    Nt = [pp["N_transition"].get(v) for v in range(pp["n_transition"])]
    di = [np.zeros(N.shape) for N in Ni]
    dt = [None if N is None else np.zeros(N.shape) for N in Nt]
    q = dict(pp); q["N_initial"] = Ni; q["N_transition"] = Nt
    zb = [int(z) for z in pp["zero_bins"]]
    out = [P.dbn_hierarchical_sample(q, di, dt, T, pp["boundaries"], zb, pp["resample_rates"], [None] * ni, R) for _ in range(n)]
    return out, R.count


def main():
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp()
    # ---- config 1
    pp = O.parse_model_txt(em_io.materialize_model("uncor_1200code_v2p1", tmp))
    r = O.uncor_sample(O.OracleModel(pp), 100, 120, 1, mode=O.RNG_MT19937)
    ref, nd = P.uncor_sample(pp, 100, 120, 1)
    assert nd == r["n_draws"] and all(np.array_equal(ref[i][0], r["init_val"][i]) and np.array_equal(ref[i][1], r["events"][i][:, :3]) for i in range(100))
    save("config1_inits.csv", r["init_val"], "out_inits of UncorEncounterModel.sample(100, 120, 'seed', 1) on uncor_1200code_v2p1.txt: 100 x 7")
    save("config1_events.csv", events_table(r["events"]), "out_events{i} stacked: [i dt var value]")
    # ---- plain dbn_hierarchical_sample loops
    for name, n, T, seed in (("uncor_1200code_v1", 40, 60, 7), ("cor_v1", 25, 50, 11)):
        pp = O.parse_model_txt(em_io.materialize_model(name, tmp))
        r = O.uncor_sample(O.OracleModel(pp), n, T, seed, mode=O.RNG_MT19937, reject=False)
        ref, nd = hier_loop_pyref(pp, n, T, seed)
        assert nd == r["n_draws"], (nd, r["n_draws"])
        for i in range(n):
            assert np.array_equal(ref[i][0], r["init_val"][i]) and np.array_equal(ref[i][1], r["events"][i][:, :3]), (name, i)
        save("hier_%s_inits.csv" % name, r["init_val"], "initial of %d successive dbn_hierarchical_sample(parms, a_i, a_t, %d, ...) calls after rng(%d,'twister'): %d x %d"
             % (n, T, seed, n, pp["n_initial"]))
        save("hier_%s_events.csv" % name, events_table(r["events"]), "events stacked: [call dt var value]")
    # ---- the terminal geometry network (@CorTerminalModel/sample.m:29-77): acType GENERIC / GENERIC, bounds_sample = +-inf (CorTerminalModel.m:81,108)
    for src, stem in (("terminalradar", "terminal_v3_radar_encounter_model"), ("opensky", "terminal_v3_opensky_encounter_model")):
        pp = O.parse_model_txt(em_io.materialize_model(stem, tmp))
        labs = [x.strip('"') for x in pp["labels_initial"]]
        io, ii = labs.index("own_speed") + 1, labs.index("int_speed") + 1
        for tag, start in (("", [0] * 15), ("_start", [2, 1, 3] + [0] * 12)):
            _, ov, _ = O.geom_sample(O.OracleModel(pp, start=start), 500, 1, mode=O.RNG_MT19937, idx_own_speed=io, idx_int_speed=ii, lim1=(50, 506), lim2=(50, 506))
            ref, _ = P.geom_sample(pp, 500, 1, lim1=(50, 506), lim2=(50, 506), start=[v or None for v in start])
            assert np.array_equal(ov, ref), (src, tag)
            save("geom_%s%s_inits.csv" % (src, tag), ov, "outInits of CorTerminalModel('srcData','%s')%s.sample(500, 'seed', 1): 500 x 15 (acType GENERIC / GENERIC)"
                 % (src, "" if not tag else " with start = {2, 1, 3, [], ...}"))
    # ---- bn_sort: every shipped graph whose order is not the identity
    rows = []
    names = sorted(os.path.splitext(f)[0] for f in os.listdir(os.path.join(ROOT, "models")) if f.endswith(".npz"))
    for k, name in enumerate(names):
        pp = O.parse_model_txt(em_io.materialize_model(name, tmp))
        for which, key in ((1, "G_initial"), (2, "G_transition")):
            G = pp.get(key)
            if G is None or np.size(G) == 0:
                continue
            order = O.bn_sort(np.asarray(G))
            if not np.array_equal(order, np.arange(1, len(order) + 1)):
                rows.append((name, which, order))
    width = max(len(o) for _, _, o in rows)
    with open(os.path.join(OUT, "bn_sort_orders.csv"), "w") as f:
        f.write("% model file stem, 1 = G_initial / 2 = G_transition, expected bn_sort order (index-lexicographic Kahn == assumed toposort 'stable'), 0-padded\n")
        for name, which, o in rows:
            f.write("%s,%d,%s\n" % (name, which, ",".join(str(int(x)) for x in list(o) + [0] * (width - len(o)))))
    # ---- trig restatements used by createEncounter.m (oracle/em_oracle.c em_sincosd, em_wrapTo360, em_atan2d)
    L = O.lib()
    import ctypes as C
    rs = np.random.RandomState(3)
    ang = np.concatenate([np.arange(-720, 721, 10.0), rs.uniform(-400, 800, 200), [1e-9, 359.99999, 44.999999, 45.0, 135.0]])
    s, c = np.zeros_like(ang), np.zeros_like(ang)
    if hasattr(L, "em_sincosd_table"):
        L.em_sincosd_table(ang.ctypes.data_as(C.c_void_p), C.c_int(len(ang)), s.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p))
        save("trig_table.csv", np.column_stack([ang, s, c]), "x_deg, sind(x), cosd(x) as restated in oracle/em_oracle.c em_sincosd")
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
