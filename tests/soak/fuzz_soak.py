"""tests/soak/fuzz_soak.py [n_models [first_seed]] -- the model fuzzer of tests/test_gpu_parity.py (tests/util.random_model) over many
seeds and both widths (3-7 and 8-14 initial variables): event lists + dense REFERENCE_AUTO + dense PER_STEP of the HIP path against
the CPU oracle, bit-exact.  Checker-side (runs the oracle; not collected by pytest).  Needs a GPU."""
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
    print("%d generated models (seeds %d..%d), event lists + dense + PER_STEP bit-exact; kernels: %s" % (count, first, first + count - 1, dict(kernels)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
