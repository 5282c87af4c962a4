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
