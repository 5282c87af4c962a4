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


def main():
    tmp = tempfile.mkdtemp()
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


if __name__ == "__main__":
    main()
