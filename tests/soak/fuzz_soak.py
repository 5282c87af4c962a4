"""tests/soak/fuzz_soak.py [n_models [first_seed]] -- the model fuzzer of tests/test_gpu_parity.py (tests/util.random_model) over many
seeds and both widths (3-7 and 8-14 initial variables): event lists (with the dense trace, alone, plain dbn_sample.m) + dense
REFERENCE_AUTO + dense PER_STEP of the HIP path against the CPU oracle, bit-exact.  Checker-side (runs the oracle; not collected by pytest).  Needs a GPU."""
import os, sys, tempfile, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for q in ("", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, q))
import numpy as np
import oracle as O
from em_model_manned_bayes_amd import em_io, native, _lib as L
from util import assert_uncor_parity, random_model


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    ctx = native.Context(0)
    tmp = tempfile.mkdtemp()
    kernels = collections.Counter()
    for seed in range(first, first + count):
        rs = np.random.RandomState(seed)
        wide = bool(seed & 1)
        parms = random_model(rs, nd=None if not wide else int(rs.randint(1, 5)), dependent=None, ni=int(rs.randint(8, 15)) if wide else None)
        path = os.path.join(tmp, "m%d.txt" % seed)
        em_io.em_write(parms, path)
        try:
            nm = native.NativeModel.load_txt(path)
        except L.EmgpuError:   # the generator now and then writes a file the loader rejects (a boundary list shorter than r + 1)
            kernels["rejected by the loader"] += 1
            continue
        pp = O.parse_model_txt(path)
        om = O.OracleModel(pp)
        n, T = int(rs.randint(300, 900)), int(rs.choice([1, 7, 8, 9, 33, 64, 97, 120]))
        try:
            ref = O.uncor_sample(om, n, T, seed)
            got = native.sample_dbn_host(ctx, nm, n, T, seed, want_dense=True, want_events=True)
            kernels[got["kernel"].split("<")[0] + ("+events" if "events" in got["kernel"] or "_ev" in got["kernel"] else "")] += 1
            assert_uncor_parity(got, ref, T)
            got = native.sample_dbn_host(ctx, nm, n, T, seed, want_dense=True, want_events=False)
            kernels[got["kernel"].split("<")[0]] += 1
            assert_uncor_parity(got, ref, T, check_events=False)
            # the list asked for alone (round 4: its rows built by the wave), and plain dbn_sample.m
            got = native.sample_dbn_host(ctx, nm, n, T, seed, want_dense=False, want_events=True)
            kernels[got["kernel"].split("<")[0] + ("+rows-by-wave" if "rows-by-wave" in got["kernel"] else "") + " (list alone)"] += 1
            assert np.array_equal(got["ev_count"], np.array([len(e) for e in ref["events"]])), "list alone: counts"
            for i in range(n):
                g, r = got["events"][i], ref["events"][i]
                assert np.array_equal(g["dt"], r[:, 0]) and np.array_equal(g["var"], r[:, 1]) and np.array_equal(g["bin"], r[:, 3]) \
                    and np.array_equal(g["value"], r[:, 2].astype(np.float32)), "list alone: trajectory %d" % i
            flags = L.FLAG_NO_RESAMPLE | L.FLAG_NO_DEDISC | L.FLAG_NO_TERMINATOR
            got = native.sample_dbn_host(ctx, nm, n, T, seed, want_dense=False, want_events=True, flags=flags, event_cap=nm.n_initial * T + 1, max_attempts=1)
            kernels[got["kernel"].split("<")[0] + ("+rows-by-wave" if "rows-by-wave" in got["kernel"] else "") + " (plain dbn_sample)"] += 1
            rb0, rev = O.dbn_sample(om, n, T, seed)
            assert np.array_equal(got["init_bin"], rb0), "plain dbn_sample: initial bins"
            for i in range(n):
                g, r = got["events"][i], rev[i]
                assert len(g["dt"]) == len(r) and np.array_equal(g["dt"], r[:, 0]) and np.array_equal(g["var"], r[:, 1]) and np.array_equal(g["bin"], r[:, 2]), \
                    "plain dbn_sample: trajectory %d" % i
            refp = O.uncor_sample(om, n, T, seed, per_step=True, want_events=False)
            got = native.sample_dbn_host(ctx, nm, n, T, seed, want_dense=True, want_events=False, transition_mode=L.TRANSITION_PER_STEP)
            assert_uncor_parity(got, refp, T, check_events=False)
        except L.EmgpuError as e:   # the generator now and then writes a boundary list shorter than r + 1: refused when the plan is compiled
            if "boundaries shorter" not in str(e):
                raise
            kernels["refused: boundaries shorter than r + 1"] += 1
        except AssertionError as e:
            print("MISMATCH seed %d (ni %d, nd %d, T %d, kernel %s): %s" % (seed, parms["n_initial"], parms["n_transition"] - parms["n_initial"], T, got["kernel"], str(e)[:200]))
            return 1
    print("%d generated models (seeds %d..%d), event lists (with the trace / alone / plain dbn_sample) + dense + PER_STEP bit-exact; kernels: %s" % (count, first, first + count - 1, dict(kernels)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
