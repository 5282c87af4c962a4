"""tests/soak/terminal_soak.py [n_encounters] -- CorTerminalModel.sample + PropagateTrajectory of the HIP path against the CPU oracle at
scale (checker-side: it runs the oracle; not collected by pytest).  For each aircraft-type pair: n encounters (4 tracks each, <= 122 rows)
on the synthetic trajectory tables -- row counts BIT-EXACT, every recorded row within 1e-6 (relative and absolute) of the oracle's f64.
Needs a GPU."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for q in ("", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, q))
import numpy as np
import oracle as O
import em_model_manned_bayes_amd as E
from em_model_manned_bayes_amd import native, synthetic


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    chunk = 50000
    ctx = native.Context(0)
    tdir = synthetic.write_terminal_directory(tempfile.mkdtemp())
    total_rows = total_tracks = 0
    for pair in (("GENERIC", "GENERIC"), ("RTCA228_A1", "RTCA228_A2"), ("RTCA228_A3", "TEST")):
        t = E.CorTerminalModel(srcData="terminalradar", parameters_directory=tdir)
        t.acType1, t.acType2 = pair
        dl = t._dyn_rows()
        oms = []
        for m in t._traj:
            pp = O.parse_model_txt(m.parameters_filename)
            oms.append(O.OracleModel(pp, alpha_transition=O.stay_prior_alpha(pp, 1.0)))
        pair_rows = pair_tracks = 0
        worst, tg, to = 0.0, 0.0, 0.0
        for c0 in range(0, n, chunk):
            nc = min(chunk, n - c0)
            seed = 0x7E50AC + 977 * c0 + len(pair[0]) * 3 + len(pair[1])
            t0 = time.time()
            _, samples = t.sample(nc, seed=seed, ctx=ctx)
            geo, mo = t._geo_rows(samples)
            got, rows = native.propagate_terminal_host(ctx, [m.native for m in t._traj], geo, mo, seed, tmax_s=120.0, dyn_limits=dl)
            t1 = time.time()
            ref, ref_rows = O.propagate(oms, mo, geo, seed, dl, tmax_s=120.0)
            t2 = time.time()
            tg += t1 - t0; to += t2 - t1
            if not np.array_equal(rows, ref_rows):
                bad = np.flatnonzero(rows != ref_rows)
                print("MISMATCH %s: %d of %d tracks differ in length (first: track %d, %d vs %d rows)" % (pair, len(bad), len(rows), bad[0], rows[bad[0]], ref_rows[bad[0]]))
                return 1
            live = np.arange(got.shape[1])[None, :] < rows[:, None]          # [tracks, cap]
            g, r = got[live], ref[live]
            err = np.abs(g - r) / np.maximum(1.0, np.abs(r))
            if not (err <= 1e-6).all():
                print("MISMATCH %s: a row differs by %.3g (relative, floor 1)" % (pair, err.max()))
                return 1
            worst = max(worst, float(err.max()))
            pair_rows += int(rows.sum()); pair_tracks += len(rows)
        total_rows += pair_rows; total_tracks += pair_tracks
        print("%-24s %7d encounters, %8d tracks, %10d recorded rows: lengths bit-exact, rows within %.1e (max)   GPU side %.1f s, oracle %.1f s (%s)"
              % ("/".join(pair), n, pair_tracks, pair_rows, worst, tg, to, ctx.last_kernel()), flush=True)
    print("TOTAL %d tracks, %d rows" % (total_tracks, total_rows))
    return 0


if __name__ == "__main__":
    sys.exit(main())
