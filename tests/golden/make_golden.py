"""tests/golden/make_golden.py -- regenerates the committed golden vectors.

The reference (MATLAB) ships no golden vectors and cannot run here, so these fixtures are
produced by the CPU oracle (oracle/em_oracle.c) after it has been cross-checked draw-for-draw
against the independent numpy restatement (oracle/pyref.py).  They pin (a) the MT19937-stream
answers a MATLAB user could later compare with `mdl.sample(100, 120, 'seed', 1)` (BASELINE.json
configs[0]) and (b) the Philox slot map of DESIGN.md section 3 across rounds.
"parity unpinned" against real MATLAB output.
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

HERE = os.path.dirname(os.path.abspath(__file__))


def pack_events(evs):
    cnt = np.array([len(e) for e in evs], dtype=np.int32)
    flat = np.concatenate(evs, axis=0)
    return cnt, flat


def terminal_golden(tmp):
    """PropagateTrajectory in Philox mode (createEncounter.m:93-265 on the synthetic trajectory tables of synthetic.py): pins the TERM_TRANS /
    TERM_DEDISC slots of DESIGN.md section 3 -- round 5 moved an attempt's first dediscretize draw to word 3 of its TERM_TRANS block."""
    import glob
    from em_model_manned_bayes_amd import synthetic
    d = synthetic.write_terminal_directory(os.path.join(tmp, "terminal"))
    files = [glob.glob(os.path.join(d, "*_" + stem + ".txt"))[0] for stem in synthetic.TERMINAL_FILE_STEMS]
    oms = []
    for f in files:
        pp = O.parse_model_txt(f)
        oms.append(O.OracleModel(pp, alpha_transition=O.stay_prior_alpha(pp, 1.0)))
    n, seed, first = 32, 0x5EED0005, 7
    rs = np.random.RandomState(20261004)
    geo = np.zeros((n, 12))
    mo = np.zeros((n, 4), dtype=np.int32)
    for a in range(2):
        dist, bear = rs.uniform(0.6, 6.0, n), rs.uniform(0, 360, n)
        geo[:, 6 * a + 0] = dist * np.cos(np.deg2rad(bear)); geo[:, 6 * a + 1] = dist * np.sin(np.deg2rad(bear))
        geo[:, 6 * a + 2] = rs.uniform(300, 3000, n); geo[:, 6 * a + 3] = rs.uniform(100, 400, n)
        geo[:, 6 * a + 4] = rs.uniform(0, 360, n)
    geo[:, 5] = rs.randint(1, 3, n); geo[:, 11] = rs.randint(1, 4, n)
    mo[:, 0] = 2 * (geo[:, 5].astype(int) - 1); mo[:, 1] = mo[:, 0] + 1
    mo[:, 2] = 4 + 2 * (geo[:, 11].astype(int) - 1); mo[:, 3] = mo[:, 2] + 1
    dl = np.array([[50, 506, 12, 5000, 100.0], [68, 338, 3, 5000, 25.0]])          # GENERIC ownship, RTCA228_A2 intruder (getDynamicLimits.m:15-62)
    out, rows = O.propagate(oms, mo, geo, seed, dl, first_index=first, tmax_s=120.0)
    cap = out.shape[1]
    keep = np.arange(cap)[None, :] < rows[:, None]
    np.savez_compressed(os.path.join(HERE, "terminal_propagate_phx_seed5eed0005_first7_32.npz"), geo=geo, model_of=mo.reshape(-1), dyn_limits=dl,
                        rows=rows, tracks=np.where(keep[:, :, None], out, 0.0), meta=np.array([n, seed, first, cap], dtype=np.int64))
    print("wrote terminal_propagate_phx_seed5eed0005_first7_32 (%d track rows, %d tracks with an event-driven turn)" % (rows.sum(), (np.abs(np.diff(out[:, :, 4], axis=1)) > 0).any(axis=1).sum()))


def main():
    tmp = tempfile.mkdtemp()
    terminal_golden(tmp)
    # ---- config 1: uncor_1200code_v2p1, sample(100, 120, 'seed', 1), MT19937 stream
    path = em_io.materialize_model("uncor_1200code_v2p1", tmp)
    pp = O.parse_model_txt(path)
    om = O.OracleModel(pp)
    r = O.uncor_sample(om, 100, 120, 1, mode=O.RNG_MT19937)
    ref, ndraw = P.uncor_sample(pp, 100, 120, 1)
    for i in range(100):  # the two restatements must agree before anything is written
        assert np.array_equal(ref[i][0], r["init_val"][i]) and np.array_equal(ref[i][1], r["events"][i][:, :3])
    assert ndraw == r["n_draws"]
    cnt, flat = pack_events(r["events"])
    np.savez_compressed(os.path.join(HERE, "config1_uncor_v2p1_mt19937_seed1_100x120.npz"),
                        init_bin=r["init_bin"], init_val=r["init_val"], ev_count=cnt, ev_flat=flat,
                        dense_bin=r["dense_bin"], n_draws=np.array([r["n_draws"]]), attempts=r["attempts"])
    # ---- Philox slot-map goldens (small): fast-branch, dependent-branch, per-step
    for name, n, T, seed, first, per_step in [("uncor_1200code_v2p1", 96, 240, 0x5EED0002, 0, False),
                                              ("uncor_1200only_fwse_v1p2", 64, 61, 7, 2**40 + 3, False),
                                              ("uncor_1200code_v1", 64, 60, 11, 5, False),
                                              ("glider_v1", 64, 50, 13, 0, False),
                                              ("uncor_1200code_v2p1", 64, 33, 99, 1, True)]:
        path = em_io.materialize_model(name, tmp)
        om = O.OracleModel(O.parse_model_txt(path))
        r = O.uncor_sample(om, n, T, seed, mode=O.RNG_PHILOX, first_index=first, per_step=per_step)
        cnt, flat = pack_events(r["events"])
        tag = "%s_philox_seed%x_first%d_%dx%d%s" % (name, seed, first, n, T, "_perstep" if per_step else "")
        np.savez_compressed(os.path.join(HERE, tag + ".npz"), init_bin=r["init_bin"], init_val=r["init_val"],
                            ev_count=cnt, ev_flat=flat, dense_bin=r["dense_bin"], dense_val=r["dense_val"],
                            attempts=r["attempts"], meta=np.array([n, T, seed, first, int(per_step)], dtype=np.int64))
        print("wrote", tag)
    # ---- sample2track.m:183-243 (hand-checkable cases first, then random columns); cor_v1 Philox golden
    rs = np.random.RandomState(20261003)
    n, T = 48, 40
    upd = np.stack([rs.normal(0, 700, (n, T)), rs.normal(0, 1.2, (n, T)), rs.normal(0, 3.0, (n, T))], axis=2)
    upd[0] = 0.0
    upd[1, :, 2] = 22.5
    upd[2, :, 0] = -3000.0
    alt0 = rs.uniform(100, 9000, n); v0 = rs.uniform(40, 250, n)
    ur = ((1852.0 / 0.3048) / 3600.0, 1.0 / 60.0, 1.0)
    xyz, flags, vmm = O.sample2track(alt0, v0, upd, *ur, 30.0, 300.0)
    np.savez_compressed(os.path.join(HERE, "sample2track_48x40.npz"), alt0=alt0, speed0=v0, updates=upd, ur=np.array(ur),
                        min_speed=np.array([30.0]), max_speed=np.array([300.0]), xyz=xyz, flags=flags, speed_minmax=vmm)
    print("wrote sample2track_48x40")
    path = em_io.materialize_model("cor_v1", tmp)
    om = O.OracleModel(O.parse_model_txt(path))
    r = O.uncor_sample(om, 48, 40, 0xC0, mode=O.RNG_PHILOX, first_index=9)
    cnt, flat = pack_events(r["events"])
    np.savez_compressed(os.path.join(HERE, "cor_v1_philox_seedc0_first9_48x40.npz"), init_bin=r["init_bin"], init_val=r["init_val"],
                        ev_count=cnt, ev_flat=flat, dense_bin=r["dense_bin"], dense_val=r["dense_val"],
                        attempts=r["attempts"], meta=np.array([48, 40, 0xC0, 9, 0], dtype=np.int64))
    print("wrote cor_v1")


if __name__ == "__main__":
    main()
