"""tests/soak/parity_soak.py [n_per_model [model ...]] -- the dense trace of the HIP path against the CPU oracle at scale (checker-side:
it runs the oracle, so it lives under tests/; not collected by pytest).  For every packed model with dynamic variables: n trajectories
x 240 s in chunks, GPU (emgpu_sample_dbn_host) vs oracle.uncor_sample_mt on every allowed core, compared BIT-EXACT (bins as u8, values
as the oracle's f64 rounded to f32).  Also the first chunk under EMGPU_TRANSITION_PER_STEP, and 20 000 trajectories with everything a call
returns (initial state, rejection attempts, dense trace AND event lists: tests/util.assert_uncor_parity, the single-thread oracle), the
same lists asked for alone, and plain dbn_sample.m (no resample rows, values = bins).  Prints one line per model and a total;
exit code 1 on the first mismatch.  Needs a GPU."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for q in ("", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, q))
import numpy as np
import oracle as O
from em_model_manned_bayes_amd import native, _lib as L
from util import assert_uncor_parity, load_pair, uncor_indices


def cores():
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        pass
    return len(os.sched_getaffinity(0))


def main():
    n_per = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    names = sys.argv[2:] or sorted(f[:-4] for f in os.listdir(os.path.join(ROOT, "models")) if f.endswith(".npz") and not f.startswith("terminal")) + ["cor_v2p1_like"]
    T, chunk, thr = 240, 125_000, cores()
    ctx = native.Context(0)
    tmp = tempfile.mkdtemp()
    total, t_start = 0, time.time()
    for name in names:
        nm, pp, _ = load_pair(name, tmp)
        if nm.n_dyn == 0:
            continue
        om = O.OracleModel(pp)
        idx = uncor_indices(pp)
        done, t0, kernels = 0, time.time(), set()
        plan = [(False, lo) for lo in range(0, n_per, chunk)] + [(True, 0)]
        for per_step, lo in plan:
            n = min(chunk, n_per - lo) if not per_step else min(chunk, n_per)
            seed, first = 0x50AC0000 + len(name), 1_000_003 * (1 + len(name)) + lo
            kw = dict(idx)
            if per_step:
                kw["transition_mode"] = L.TRANSITION_PER_STEP
            # (round 6: the host path is a pipeline -- every other chunk goes into pageable arrays, i.e. through the staging buffers and the host
            # threads in two pieces; the others into the context's pinned pool, i.e. pitched copies by the copy engine)
            got = native.sample_dbn_host(ctx, nm, n, T, seed, first_index=first, want_dense=True, want_events=False, pinned=(lo // chunk) % 2 == 0, **kw)
            kernels.add(got["kernel"])
            rb, rv = O.uncor_sample_mt(om, n, T, seed, thr, first_index=first, per_step=per_step)
            if not np.array_equal(got["dyn_bin"], rb):
                bad = np.argwhere(got["dyn_bin"] != rb)[0]
                print("MISMATCH %s per_step=%s: bin of trajectory %d second %d variable %d: %d vs %d" % (name, per_step, first + bad[0], bad[1], bad[2], got["dyn_bin"][tuple(bad)], rb[tuple(bad)]))
                return 1
            if not np.array_equal(got["dyn_val"], rv.astype(np.float32)):
                bad = np.argwhere(got["dyn_val"] != rv.astype(np.float32))[0]
                print("MISMATCH %s per_step=%s: value of trajectory %d second %d variable %d: %r vs %r" % (name, per_step, first + bad[0], bad[1], bad[2], got["dyn_val"][tuple(bad)], rv[tuple(bad)]))
                return 1
            done += n
        ne = 20_000   # the full output set, event lists included
        got = native.sample_dbn_host(ctx, nm, ne, T, 0x50AC1111, first_index=424_242, want_dense=True, want_events=True, **idx)
        ref = O.uncor_sample(om, ne, T, 0x50AC1111, first_index=424_242)
        try:
            assert_uncor_parity(got, ref, T)
        except AssertionError as e:
            print("MISMATCH %s (full outputs): %s" % (name, str(e)[:300]))
            return 1
        kernels.add(got["kernel"])
        # the list asked for alone (round 4: the rows of a block built by the wave on the fast-branch models), against the same oracle run
        alone = native.sample_dbn_host(ctx, nm, ne, T, 0x50AC1111, first_index=424_242, want_dense=False, want_events=True, event_cap=2048, **idx)
        kernels.add(alone["kernel"])
        if not np.array_equal(alone["ev_count"], np.array([len(e) for e in ref["events"]])):
            print("MISMATCH %s (list alone): row counts" % name)
            return 1
        for i in range(ne):
            g, r = alone["events"][i], ref["events"][i]
            if not (np.array_equal(g["dt"], r[:, 0]) and np.array_equal(g["var"], r[:, 1]) and np.array_equal(g["bin"], r[:, 3])
                    and np.array_equal(g["value"], r[:, 2].astype(np.float32))):
                print("MISMATCH %s (list alone): trajectory %d" % (name, 424_242 + i))
                return 1
        # plain dbn_sample.m (no resample rows, values = bins, no terminator)
        flags = L.FLAG_NO_RESAMPLE | L.FLAG_NO_DEDISC | L.FLAG_NO_TERMINATOR
        plain = native.sample_dbn_host(ctx, nm, ne, T, 0x50AC2222, first_index=77, want_dense=False, want_events=True, flags=flags,
                                       event_cap=nm.n_initial * T + 1, max_attempts=1)
        kernels.add(plain["kernel"] + "(plain)")
        rb0, rev = O.dbn_sample(om, ne, T, 0x50AC2222, first_index=77)
        if not np.array_equal(plain["init_bin"], rb0):
            print("MISMATCH %s (plain dbn_sample): initial bins" % name)
            return 1
        for i in range(ne):
            g, r = plain["events"][i], rev[i]
            if not (len(g["dt"]) == len(r) and np.array_equal(g["dt"], r[:, 0]) and np.array_equal(g["var"], r[:, 1]) and np.array_equal(g["bin"], r[:, 2])):
                print("MISMATCH %s (plain dbn_sample): trajectory %d" % (name, 77 + i))
                return 1
        total += done
        total_ev = ne
        print("%-34s %8d trajectories x %d s dense bit-exact + %d with event lists (with the trace, alone, plain dbn_sample)  %5.1f s  %s" % (name, done, T, ne, time.time() - t0, ", ".join(sorted(kernels))), flush=True)
    print("TOTAL %d trajectories x %d s bit-exact on %d oracle threads in %.0f s" % (total, T, thr, time.time() - t_start))
    return 0


if __name__ == "__main__":
    sys.exit(main())
