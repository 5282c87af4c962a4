"""Shared helpers for the parity tests: run the HIP path and the oracle on the same inputs."""
import numpy as np

import oracle as O
from em_model_manned_bayes_amd import em_io, native, _lib as L

_cache = {}


def load_pair(name, model_dir, **read_kw):
    """(native model, oracle parms dict) for a packed model, both parsed from the SAME .txt file:
    the native one by the C++ loader, the oracle one by oracle.parse_model_txt."""
    key = (name, tuple(sorted(read_kw.items())))
    if key not in _cache:
        if name == "cor_v2p1_like":           # the generator-made stand-in for the absent cor_v2p1.txt
            from em_model_manned_bayes_amd import synthetic
            path = synthetic.write_correlated_v2p1_like(model_dir)
        else:
            path = em_io.materialize_model(name, model_dir)
        nm = native.NativeModel.load_txt(path, read_kw.get("idx_zero_boundaries", (1, 2, 3)), read_kw.get("is_overwrite_zero_boundaries", False))
        pp = O.parse_model_txt(path, read_kw.get("idx_zero_boundaries", (1, 2, 3)), read_kw.get("is_overwrite_zero_boundaries", False))
        _cache[key] = (nm, pp, path)
    return _cache[key]


def label_index(labels, name):
    q = '"%s"' % name
    return labels.index(q) + 1 if q in labels else 0


def uncor_indices(pp):
    labs = pp["labels_initial"]
    return dict(idx_L=label_index(labs, "L"), idx_v=label_index(labs, "v"), idx_dh=label_index(labs, "\\dot h"))


def assert_uncor_parity(got, ref, T, check_events=True, tol_rel=1e-6):
    """got: native.sample_dbn_host dict; ref: oracle.uncor_sample dict.
    Discrete state/event sequence bit-exact; dediscretised floats within tol_rel relative
    (north_star: 1e-6) -- and, because both sides round the same f64 expression to f32, we also
    require exact equality of the f32 values."""
    assert np.array_equal(got["init_bin"].astype(np.int32), ref["init_bin"]), "initial bins differ"
    assert np.array_equal(got["attempts"], ref["attempts"]), "rejection attempts differ"
    rv = ref["init_val"].astype(np.float32)
    np.testing.assert_allclose(got["init_val"], ref["init_val"], rtol=tol_rel, atol=0)
    assert np.array_equal(got["init_val"], rv), "initial values not bit-equal after f32 rounding"
    if "dyn_bin" in got:
        assert np.array_equal(got["dyn_bin"], ref["dense_bin"]), "dense bins differ"
        np.testing.assert_allclose(got["dyn_val"], ref["dense_val"], rtol=tol_rel, atol=0)
        assert np.array_equal(got["dyn_val"], ref["dense_val"].astype(np.float32)), "dense values not bit-equal"
    if check_events and "events" in got:
        for i, (e, r) in enumerate(zip(got["events"], ref["events"])):
            assert len(e) == r.shape[0], "trajectory %d: %d vs %d event rows" % (i, len(e), r.shape[0])
            assert np.array_equal(e["dt"].astype(np.float64), r[:, 0]), "trajectory %d: dt differs" % i
            assert np.array_equal(e["var"].astype(np.float64), r[:, 1]), "trajectory %d: var differs" % i
            assert np.array_equal(e["bin"].astype(np.float64), r[:, 3]), "trajectory %d: bin differs" % i
            np.testing.assert_allclose(e["value"], r[:, 2], rtol=tol_rel, atol=0)
            assert np.array_equal(e["value"], r[:, 2].astype(np.float32)), "trajectory %d: values not bit-equal" % i


def random_model(rs, dependent=None, nd=None, ni=None):
    """A random small model in the em_read dict layout (for em_io.em_write): random DAGs, sparse count tables
    (zero entries, all-zero columns), categorical and continuous variables, zero-crossing boundaries,
    zero and non-zero resample rates.  rs: numpy RandomState."""
    ni = int(ni or rs.randint(3, 8))
    r = rs.randint(2, 9, ni)
    if rs.rand() < 0.3:
        r[rs.randint(ni)] = rs.randint(9, 13)
    nd = int(nd or rs.randint(1, min(ni, 4) + 1))
    dyn = sorted(rs.choice(ni, nd, replace=False).tolist())
    Gi = np.zeros((ni, ni), dtype=np.uint8)
    for v in range(1, ni):
        cand = list(range(v))
        rs.shuffle(cand)
        q = 1
        for p in cand[: rs.randint(0, 4)]:
            if q * r[p] <= 400:
                Gi[p, v] = 1
                q *= r[p]
    nt = ni + nd
    rt = np.concatenate([r, r[dyn]])
    Gt = np.zeros((nt, nt), dtype=np.uint8)
    dependent = (rs.rand() < 0.5) if dependent is None else dependent
    for k, d in enumerate(dyn):
        v = ni + k
        Gt[d, v] = 1
        q = r[d]
        cand = [p for p in range(ni) if p != d]
        rs.shuffle(cand)
        for p in cand[: rs.randint(0, 3)]:
            if q * r[p] <= 600:
                Gt[p, v] = 1
                q *= r[p]
        if dependent and k > 0:
            for kk in range(k):
                if rs.rand() < 0.6 and q * rt[ni + kk] <= 900:
                    Gt[ni + kk, v] = 1
                    q *= rt[ni + kk]

    def counts(rv, parents_r):
        q = int(np.prod(parents_r)) if len(parents_r) else 1
        N = rs.randint(1, 2000, (rv, q)).astype(np.float64)
        N *= rs.rand(rv, q) < rs.choice([0.35, 0.6, 0.9])
        N[:, rs.rand(q) < 0.05] = 0                      # all-zero columns: select_random gives bin 1
        if rs.rand() < 0.3:
            N[:, rs.randint(q)] = 0
            N[rs.randint(rv), rs.randint(q)] = 1.56e9    # one huge count (dueregard-size)
        return N

    N_initial = [counts(r[v], r[Gi[:, v] > 0]) for v in range(ni)]
    N_transition = [np.zeros((0, 0))] * ni + [counts(rt[v], rt[Gt[:, v] > 0]) for v in range(ni, nt)]
    boundaries = []
    for v in range(ni):
        if v not in dyn and rs.rand() < 0.35:
            boundaries.append(np.zeros(0))               # categorical
            continue
        if rs.rand() < 0.6:                              # a bin that straddles 0 => zero bin (em_read.m:143-156)
            lo, hi = -rs.uniform(1, 50), rs.uniform(1, 50)
        else:
            lo = rs.uniform(0, 100); hi = lo + rs.uniform(1, 500)
        e = np.sort(np.round(rs.uniform(lo, hi, r[v] - 1), 3))
        boundaries.append(np.unique(np.concatenate([[np.round(lo, 3)], e, [np.round(hi, 3)]])) if len(np.unique(e)) == r[v] - 1 else np.round(np.linspace(lo, hi, r[v] + 1), 3))
    rates = np.where(rs.rand(ni) < 0.4, 0.0, np.round(rs.uniform(0.001, 0.25, ni), 6))
    labels_initial = ['"v%d"' % (v + 1) for v in range(ni)]
    labels_transition = ['"v%d(t)"' % (v + 1) if v in dyn else '"v%d"' % (v + 1) for v in range(ni)] + ['"v%d(t+1)"' % (d + 1) for d in dyn]
    return {"n_initial": ni, "n_transition": nt, "labels_initial": labels_initial, "labels_transition": labels_transition,
            "G_initial": Gi, "G_transition": Gt, "r_initial": r, "r_transition": rt, "N_initial": N_initial,
            "N_transition": N_transition, "boundaries": boundaries, "resample_rates": rates}


def assert_parting_only_on_a_threshold(got_attempts, ref_attempts, margins, tol, what):
    """.track parity, exact about every disagreement: GPU and oracle must accept the SAME attempt of every unit, except where the
    oracle's own decision margin of the attempt at which the two part (one side accepted it, the other rejected it) is below `tol`
    -- some value of that attempt sat on a threshold to within the last bits in which device and host arithmetic differ.
    margins[i, j]: smallest |value - threshold| / scale over every discrete decision of attempt j + 1 of unit i (oracle em_note).
    Returns the boolean mask of units that agree."""
    got_attempts, ref_attempts = np.asarray(got_attempts), np.asarray(ref_attempts)
    same = got_attempts == ref_attempts
    for i in np.flatnonzero(~same):
        g, r = int(got_attempts[i]), int(ref_attempts[i])
        j = min(x for x in (g, r) if x > 0)   # the earlier acceptance: the other side rejected this attempt (or never accepted)
        assert j - 1 < margins.shape[1], "%s %d: parted at attempt %d, beyond the recorded margins" % (what, i, j)
        m = margins[i, j - 1]
        assert m < tol, ("%s %d: GPU accepted attempt %d, oracle attempt %d, but no decision of attempt %d was closer than %.3g "
                         "to its threshold (tolerance %.3g): a real accept/reject difference" % (what, i, g, r, j, m, tol))
    return same


def f32_ulp_distance(a, b):
    """|a - b| counted in f32 representation steps (a, b: f32 arrays of one shape; +0 and -0 are 0 apart)."""
    ia = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    ib = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


def assert_f32_of_f64(got, ref, what="", abs_floor=1e-9):
    """`got` (f32, what the device stored) against `ref` (the oracle's f64): equal after rounding the oracle's value to f32, or ONE f32 step
    apart -- the device's and the host's f64 transcendental functions differ in their last bits (1e-15 relative), which moves a value
    across an f32 rounding boundary now and then but never further.  abs_floor: a value that is a difference of larger quantities (x_nm
    near the runway, a heading just past 0) carries their absolute error; below abs_floor the absolute difference is compared instead.
    Replaces rtol = atol = 1e-6 on the terminal tracks (the soak's worst observation was 6e-8 relative = half an f32 step)."""
    got = np.asarray(got, dtype=np.float32)
    ref = np.asarray(ref, dtype=np.float64)
    d = f32_ulp_distance(got, ref.astype(np.float32))
    bad = (d > 1) & (np.abs(got.astype(np.float64) - ref) > abs_floor)
    if bad.any():
        k = np.argwhere(bad)[0]
        raise AssertionError("%s: %d of %d values more than one f32 step from the oracle, first at %s: %r vs %r"
                             % (what, int(bad.sum()), bad.size, tuple(k), got[tuple(k)], ref[tuple(k)]))
    return int(d.max()) if d.size else 0
